cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "resnet_conv" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_synth_gpu.py tests/test_bench_shapes_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do ASTTS_RCONV_MT=1 timeout 300 python scripts/flow_only.py; timeout 300 python scripts/flow_only.py; done
