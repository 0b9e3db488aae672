#!/bin/bash
# round 4, session 1: CU census, nt weight loads A/B, decode chains on masked streams, complementary-mask pipeline
mkdir -p gpurun_out
cd scripts/micro
echo "=== census"; timeout 120 ./cu_census
echo "=== nt A/B (B=8)"
for rep in 1 2; do
  for bin in decode_chain decode_chain_nt; do
    echo "--- $bin"; DC_KSPLIT=2 timeout 200 ./$bin 8 200 2>&1 | grep -E "eager:|eager, TWO|eager, 2 concurrent|eager, 3 concurrent" | head -4
  done
done
echo "=== chains on masked streams (both chains share the high N mask bits)"
for n in 128 64 32; do
  echo "--- N=$n"; DC_CUMASK=3 DC_CUMASK_N=$n DC_KSPLIT=2 timeout 200 ./decode_chain 8 200 2>&1 | grep -E "eager, 2 concurrent" | head -2
done
cd ../..
echo "=== complementary masks in the pipeline"
timeout 900 python3 scripts/cu_mask_probe2.py 2>&1 | grep -v Warning
