#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | head -5
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
echo "=== default bench"; timeout 900 python bench.py 2> gpurun_out/r04_bench_d.err > gpurun_out/r04_bench_d.json; tail -1 gpurun_out/r04_bench_d.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_bench_d.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'sequential_ms_per_step')}, d['pipelining'][:14], 'cobatch', d['cobatched_lm_side_measurement']['value'], '24k', d['value_24khz']['value'])
PY
echo "=== force-dist bench"; timeout 900 python bench.py --force-dist --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['rccl_backend'], d['gathered_ids_match_oracle'])"
