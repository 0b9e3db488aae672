#!/bin/bash
run() { v=$(env "$@" timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d['pipelining'][:14])"); echo "$*: $v"; }
run ASTTS_PIPE_LM_PRIORITY=0
run ASTTS_PIPE_LM_PRIORITY=-1
run ASTTS_PIPE_LM_PRIORITY=0
run ASTTS_PIPE_LM_PRIORITY=-1
run ASTTS_PIPE_LM_PRIORITY=-1 GPU_MAX_HW_QUEUES=8
