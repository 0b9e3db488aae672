#!/bin/bash
# engine v3: per-kernel trace (v2 and v3 in one process)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf /tmp/pv3
LM_TIME_ROWS=8 LM_TIME_ENGINES=v2,v3 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pv3 -o v3 --output-format csv -- python3 scripts/lm_engine_time.py > /dev/null 2>&1
f=$(find /tmp/pv3 -name "*kernel_stats.csv" | head -1)
head -14 "$f" | cut -c1-220
