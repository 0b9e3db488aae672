cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_synth_gpu.py -x -q -m gpu -k "resnet or flow" 2>&1 | tail -2
for i in 1 2; do ASTTS_TFM_PREFETCH=0 timeout 300 python scripts/flow_only.py; timeout 300 python scripts/flow_only.py; done
