cd $GRAFT_REPO_ROOT
timeout 300 python scripts/sampler_probe.py
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "ras" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_synth_gpu.py tests/test_lm_step_gpu.py -x -q -m gpu 2>&1 | tail -2
