#!/bin/bash
# round 4, session 3: full GPU suite, chains-only masks, round profiles
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -8
echo "=== masks: chains only"
PROBE2_ONLY_CHAINS=1 timeout 900 python3 scripts/cu_mask_probe2.py 2>&1 | grep -E "unmasked:|chains high|pipe classes" | grep -v "render low"
echo "=== profiles"
bash scripts/g_profiles.sh r04 2>&1 | tail -30
