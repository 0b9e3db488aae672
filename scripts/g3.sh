cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_synth_gpu.py -x -q -m gpu -k "end_to_end" -s 2>&1 | grep -E "parity|passed|failed|Error|error" | tail -8
