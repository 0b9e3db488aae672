#!/bin/bash
# lm_attn with one round trip for a chunk's keys (unconditional clamped loads) + sampler logits in one round trip: parity + timing
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_lm_step_gpu.py tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -4
echo "=== alone"
LM_TIME_ENGINES=v2 timeout 300 python scripts/lm_engine_time.py 2>&1 | grep "^b="
echo "=== pipelined"
run() { v=$(env "$@" timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print(round(d['value'],1), round(d['ms_per_step'],2), 'lm', s['lm_ms'], 'flow', s['flow_ms'], d['pipelining'][:14])"); echo "$*: $v"; }
run A=0
run A=0
