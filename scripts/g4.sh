cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lm_step_gpu.py tests/test_synth_gpu.py -q -m gpu -x --tb=short 2>&1 | tail -2
for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03_bench4.json 2> gpurun_out/r03_bench4.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench4.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','stages_ms','sequential_ms_per_step')}); print(d['cobatched_lm_side_measurement']['value'], d['cobatched_lm_side_measurement']['decode_chains'], d['cobatched_lm_side_measurement']['batches_per_chain']); print(d['value_24khz']['value']); print(d['roofline']['frac'], d['roofline']['avg_us'])
PY
done
