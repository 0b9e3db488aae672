#!/bin/bash
# A/B in one box: the L2 weight prefetch of the fused transformer-block kernels inside the PIPELINED benchmark
for rep in 1 2; do
  for pf in 1 0; do
    v=$(ASTTS_TFM_PREFETCH=$pf timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d['stages_ms'])")
    echo "prefetch=$pf: $v"
  done
done
