cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "tfm" 2>&1 | tail -2
timeout 300 python scripts/flow_only.py
