cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py -x -q -m gpu -s 2>&1 | grep -E "config 3|passed|failed" | tail
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py 2>&1 | tail -2
