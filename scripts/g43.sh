#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for w in config3 config5 config4; do timeout 1200 python bench.py --workload $w --steps 2 --warmup 1 2>/dev/null > gpurun_out/r04_side_${w}_end.json; python3 -c "import json,sys; d=json.loads(open('gpurun_out/r04_side_${w}_end.json').read().strip().splitlines()[-1]); print('$w', round(d['value'],1), round(d['ms_per_step'],1))"; done
