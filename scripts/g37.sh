#!/bin/bash
# upper bound of launch fusion in engine v2: the pipelined step with the out-projection (and FFN-out) launches simply dropped (garbage results)
cd "$GRAFT_REPO_ROOT"
run() { v=$(env "$@" timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print(round(d['value'],1), round(d['ms_per_step'],2), 'lm', s['lm_ms'], 'flow', s['flow_ms'], d['pipelining'][:14])"); echo "$*: $v"; }
run A=0
run ASTTS_LM_SKIP=1
run ASTTS_LM_SKIP=3
run A=0
run ASTTS_LM_SKIP=1
