cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_synth_gpu.py -x -q -m gpu -k "stream" -s 2>&1 | grep -E "parity|passed|failed|Error|error" | tail -20
timeout 900 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "contract" 2>&1 | tail -15
