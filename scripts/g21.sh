#!/bin/bash
# 4 hardware queues as the default: full GPU suite, then A/B of the front-stream prefill and the ragged scheduler
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | head -5
run() { v=$(env "$@" timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d['pipelining'][:14])"); echo "$*: $v"; }
run A=0
run ASTTS_PIPE_FRONT_PREFILL=0
run A=0
run ASTTS_PIPE_FRONT_PREFILL=0
for w in config3 config5 config4; do timeout 900 python bench.py --workload $w --steps 2 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['value'],1), round(d['ms_per_step'],1))"; done
