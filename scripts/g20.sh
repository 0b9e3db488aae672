#!/bin/bash
# full default bench at 4 vs 8 hardware queues, and deeper pipelines at 4
run() { v=$(env "$@" timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cobatched_lm_side_measurement']; print(round(d['value'],1), round(d['ms_per_step'],2), d['pipelining'][:14], 'cobatch', round(c['value'],1), c['decode_chains'], c['batches_per_chain'], '24k', round(d['value_24khz']['value'],1))"); echo "$*: $v"; }
run GPU_MAX_HW_QUEUES=4
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=4 ASTTS_BENCH_DEPTHS=3,4
run GPU_MAX_HW_QUEUES=4 ASTTS_BENCH_DEPTHS=4,5
run GPU_MAX_HW_QUEUES=4
