#!/bin/bash
# round 4, session 4: full GPU suite, config-3 driver test with its prints, default bench (calibrated cpu_baseline, kNN traffic)
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -4
python -m pytest tests/test_configs_gpu.py::test_config3_longform_through_its_driver -x -q -m gpu -s 2>&1 | grep -E "config 3|passed|failed"
echo "=== default bench"; timeout 900 python bench.py 2> gpurun_out/r04_bench_b.err > gpurun_out/r04_bench_b.json; tail -2 gpurun_out/r04_bench_b.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_bench_b.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'sequential_ms_per_step', 'stages_ms', 'knn_qps')})
print('cpu_baseline', d.get('cpu_baseline'))
print('knn_roofline', d['knn_roofline'])
print('cobatch', d['cobatched_lm_side_measurement']['value'], '24k', d['value_24khz']['value'])
r = d['roofline']; print('roofline', {k: r[k] for k in ('kernel', 'achieved', 'frac', 'avg_us', 'traffic')}, r['pipelined'])
PY
