cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lm_step_gpu.py -q -m gpu -x --tb=short -k "graph_replay" 2>&1 | tail -5
for gon in 1 0 1; do ASTTS_LM_GRAPH=$gon ASTTS_LM_DEBUG_TIMING=1 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2> gpurun_out/r03_dbg.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph=$gon', round(d['value'],1), round(d['ms_per_step'],2), d['stages_ms'])"
grep "host enqueue" gpurun_out/r03_dbg.err | tail -12 | awk '{print $(NF-1)}' | tr '\n' ' '; echo; done
