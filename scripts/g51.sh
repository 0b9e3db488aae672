#!/bin/bash
# the documented fallback switches still work: FFN-out as one K slice; engine v1
cd "$GRAFT_REPO_ROOT"
ASTTS_LM_FFN_SPLIT=0 timeout 1200 python -m pytest tests/test_lm_step_gpu.py -m gpu -x -q 2>&1 | tail -2
