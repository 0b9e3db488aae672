#!/bin/bash
# lm_gemv: out-projection's partial loads unconditional (one round trip with the weights) on top of the row prefetch; base = previous commit
cd "$GRAFT_REPO_ROOT"
L=autostyle-tts_amd/astts
timeout 1500 python -m pytest tests/test_lm_step_gpu.py -m gpu -x -q 2>&1 | tail -3
cp $L/libastts.so /tmp/new.so
LM_TIME_ENGINES=v2,v2 timeout 300 python scripts/lm_engine_time.py 2>&1 | grep "^b=" | sed "s/^/new /"
run() { python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print('$1', round(d['value'],1), round(d['ms_per_step'],2), 'seq', d['sequential_ms_per_step'], s)"; }
for i in 1 2; do
  cp $L/libastts_base.so $L/libastts.so; run base
  cp /tmp/new.so $L/libastts.so; run new
done
