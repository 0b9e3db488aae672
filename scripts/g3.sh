cd $GRAFT_REPO_ROOT
timeout 600 python scripts/conv_probe.py 2>&1 | tail -6
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
