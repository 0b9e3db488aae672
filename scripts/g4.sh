cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_knn_gpu.py -q -m gpu -x --tb=short 2>&1 | tail -3
python scripts/knn_bench.py 2>&1 | tail -6
ASTTS_GEMM_RING=1 python scripts/knn_bench.py 2>&1 | tail -4 | grep "Q= 256"
