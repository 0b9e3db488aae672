cd $GRAFT_REPO_ROOT
timeout 300 python scripts/gn_probe.py
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "groupnorm or norm" 2>&1 | tail -2
timeout 300 python scripts/flow_only.py
