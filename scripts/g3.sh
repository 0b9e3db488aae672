cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_synth_gpu.py -x -q -m gpu -k "pipe_classes" -s 2>&1 | grep -E "pipes|passed|failed|Error|error|assert" | tail -8
