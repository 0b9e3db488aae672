cd $GRAFT_REPO_ROOT
timeout 300 python scripts/knn_small.py 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_knn_gpu.py -x -q -m gpu 2>&1 | tail -3
