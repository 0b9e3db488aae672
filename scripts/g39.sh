#!/bin/bash
# marginal cost of each launch of a layer in the chain: 250 steps alone with one kind of launch dropped (garbage results, timing only)
cd "$GRAFT_REPO_ROOT"
for s in 0 4 8 1 16 2 0; do ASTTS_LM_SKIP=$s LM_TIME_ROWS=8,32 LM_TIME_ENGINES=v2 timeout 300 python scripts/lm_engine_time.py 2>&1 | grep "^b=" | sed "s/^/skip=$s /"; done
