#!/bin/bash
# A/B in one box: staggered start of the decode chains (steps 16 = bench default; 20 = what the driver passed in round 3)
run() { v=$(env "$@" timeout 600 python bench.py --steps ${STEPS:-16} --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d['pipelining'][:14])"); echo "$* steps=${STEPS:-16}: $v"; }
for st in 0 30 60 90 120 0 60; do run ASTTS_PIPE_STAGGER_MS=$st; done
STEPS=20; for st in 0 60 90; do run ASTTS_PIPE_STAGGER_MS=$st; done
STEPS=8; for st in 0 60 90; do run ASTTS_PIPE_STAGGER_MS=$st; done
