#!/bin/bash
mkdir -p gpurun_out
bash scripts/g_profiles.sh r04 2>&1 | tail -12
echo "=== default bench"; timeout 900 python bench.py 2> gpurun_out/r04_bench_c.err > gpurun_out/r04_bench_c.json; tail -2 gpurun_out/r04_bench_c.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_bench_c.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'sequential_ms_per_step', 'stages_ms', 'knn_qps')}, d['pipelining'][:40])
print('cpu_baseline', d.get('cpu_baseline'))
c = d['cobatched_lm_side_measurement']; print('cobatch', c['value'], c['decode_chains'], c['batches_per_chain'], '24k', d['value_24khz']['value'])
r = d['roofline']; print('roofline', {k: r[k] for k in ('kernel', 'achieved', 'frac', 'avg_us', 'traffic')}, r['pipelined']['avg_us'], r['pipelined']['frac'])
PY
