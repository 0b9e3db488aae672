cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_lm_step_gpu.py tests/test_configs_gpu.py tests/test_synth_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err; tail -c 600 gpurun_out/r03_bench2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','stages_ms','sequential_ms_per_step')}); print(d['cobatched_lm_side_measurement']); print(d['value_24khz']); 
PY
