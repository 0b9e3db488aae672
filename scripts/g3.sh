cd $GRAFT_REPO_ROOT
PROBE_B=64 PROBE_TS=300 PROBE_ITERS=2 timeout 900 python scripts/fullsize_probe.py 2>&1 | tail -2
timeout 900 python scripts/ragged_probe.py 2>&1 | tail -3
