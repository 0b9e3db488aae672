#!/bin/bash
# sensitivity of the three-chain pipelined step to the render stage (L2 weight prefetch off: flow +1.7 ms) and to the decode chain
# (wide forms + unsplit attention: LM +10 ms per 250 steps; engine v1: +40 ms)
run() { v=$(env "$@" timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print(round(d['value'],1), round(d['ms_per_step'],2), 'lm', s['lm_ms'], 'flow', s['flow_ms'], d['pipelining'][:14])"); echo "$*: $v"; }
run A=0
run ASTTS_TFM_PREFETCH=0
run ASTTS_LM_HALF8_MAX_BLOCKS=0 ASTTS_LM_KSPLIT=1 ASTTS_LM_WIDE=1
run ASTTS_LM_ENGINE=v1
run A=0
