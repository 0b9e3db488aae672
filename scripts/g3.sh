cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_synth_gpu.py -x -q -m gpu -k "norm or flow or engine" 2>&1 | tail -2
timeout 600 bash scripts/g2.sh 2>&1 | grep -E "launches|groupnorm|flow solve"
timeout 300 python scripts/flow_only.py
