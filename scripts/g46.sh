#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_lm_step_gpu.py tests/test_lm_fused_gpu.py -m gpu -q 2>&1 | tail -25
LM_TIME_ENGINES=v2,v2 timeout 300 python scripts/lm_engine_time.py 2>&1 | grep "^b="
