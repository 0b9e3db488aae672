#!/bin/bash
# scratch: decode-step kernels after the VALU lane-exchange reductions
cd scripts/micro
for b in 8 16 32; do
  echo "=== B=$b ops"; DC_KSPLIT=2 DC_OPS=1 timeout 120 ./decode_chain $b 50 2>&1 | tail -7
  echo "=== B=$b chain"; DC_KSPLIT=2 timeout 120 ./decode_chain $b 200 2>&1 | grep -E "1 chain|2 concurrent|eager" | head -3
done
for b in 8 32; do echo "== stamps B=$b"; DC_KSPLIT=2 timeout 60 ./decode_chain_stamps $b 10 | sed -n 3,9p; done
