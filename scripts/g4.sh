cd $GRAFT_REPO_ROOT
timeout 300 python scripts/queue_probe.py
