cd $GRAFT_REPO_ROOT
ASTTS_LM_ENGINE=v1 PROBE_B=64 PROBE_TS=1500 PROBE_ITERS=2 timeout 600 python scripts/fullsize_probe.py 2>&1 | grep iter
PROBE_B=64 PROBE_TS=1500 PROBE_ITERS=2 timeout 600 python scripts/fullsize_probe.py 2>&1 | grep iter
timeout 1500 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -30
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03_bench1.json 2> gpurun_out/r03_bench1.err; tail -c 600 gpurun_out/r03_bench1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','stages_ms','sequential_ms_per_step')}); print(d['cobatched_lm_side_measurement']); print(d['value_24khz']); print(d['roofline'])
PY
