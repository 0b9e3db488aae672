cd $GRAFT_REPO_ROOT
timeout 1100 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "^$" > gpurun_out/r03_full_3.log; tail -2 gpurun_out/r03_full_3.log
FLOW_N=3 python3 scripts/flow_only.py 2>&1 | tail -1
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03_bench3.json 2> gpurun_out/r03_bench3.err; tail -c 400 gpurun_out/r03_bench3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench3.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','stages_ms','sequential_ms_per_step')}); print(d['cobatched_lm_side_measurement']); print(d['value_24khz']); print(d['roofline']['frac'], d['roofline']['avg_us'], d['roofline']['sequential'])
PY
