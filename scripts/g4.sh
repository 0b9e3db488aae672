cd $GRAFT_REPO_ROOT
for t in 2 4 2 4 2 4; do ASTTS_BENCH_TRIALS=$t ASTTS_BENCH_VERBOSE=1 timeout 600 python bench.py --no-cpu-baseline --no-24khz --no-cobatch 2> gpurun_out/r03_b.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('trials $t:', round(d['value'],1), round(d['ms_per_step'],2), d['pipelining'][-60:])"; grep autotune gpurun_out/r03_b.err | tr '\n' ';' | cut -c1-400; echo; done
