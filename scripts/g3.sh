cd $GRAFT_REPO_ROOT
timeout 300 python scripts/tfm_probe.py
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k tfm 2>&1 | tail -2
