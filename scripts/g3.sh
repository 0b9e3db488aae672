cd $GRAFT_REPO_ROOT
timeout 600 python scripts/phase_probe.py 2>&1 | tail -8
