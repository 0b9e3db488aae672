#!/bin/bash
# A/B in one box: fewer, fatter decode workgroups inside the PIPELINED benchmark
run() { v=$(env "$@" timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d['stages_ms']['lm_ms'])"); echo "$*: $v"; }
for rep in 1 2; do
  run A=0
  run ASTTS_LM_HALF8_MAX_BLOCKS=0
  run ASTTS_LM_KSPLIT=1
  run ASTTS_LM_HALF8_MAX_BLOCKS=0 ASTTS_LM_KSPLIT=1
  run GPU_MAX_HW_QUEUES=4
done
