cd $GRAFT_REPO_ROOT
for i in 1 2; do
ASTTS_TFM_BALANCE=0 timeout 300 python scripts/tfm_probe.py
ASTTS_TFM_BALANCE=1 timeout 300 python scripts/tfm_probe.py
done
for i in 1 2; do
ASTTS_TFM_BALANCE=0 timeout 300 python scripts/flow_only.py
ASTTS_TFM_BALANCE=1 timeout 300 python scripts/flow_only.py
done
