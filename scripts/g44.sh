#!/bin/bash
# lm_gemv: all rows of a wave prefetched in one round trip (16 / 32-row launches) -- parity, chain timing, side workloads; base = previous commit
cd "$GRAFT_REPO_ROOT"
L=autostyle-tts_amd/astts
timeout 1500 python -m pytest tests/test_lm_step_gpu.py tests/test_synth_gpu.py -m gpu -x -q 2>&1 | tail -3
cp $L/libastts.so /tmp/new.so
for which in base new base new; do
  if [ $which = base ]; then cp $L/libastts_base.so $L/libastts.so; else cp /tmp/new.so $L/libastts.so; fi
  LM_TIME_ENGINES=v2 timeout 300 python scripts/lm_engine_time.py 2>&1 | grep "^b=" | sed "s/^/$which /"
done
for which in base new base new; do
  if [ $which = base ]; then cp $L/libastts_base.so $L/libastts.so; else cp /tmp/new.so $L/libastts.so; fi
  for w in config3 config5; do timeout 1200 python bench.py --workload $w --steps 2 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which $w', round(d['value'],1), round(d['ms_per_step'],1))"; done
done
cp /tmp/new.so $L/libastts.so
