cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_synth_gpu.py -x -q -m gpu -k "hift or engine or stream" -s 2>&1 | grep -E "parity|passed|failed|Error|error|assert" | tail -12
timeout 300 python scripts/hift_only.py 2>&1 | tail -1
