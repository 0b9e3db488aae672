#!/bin/bash
# engine v3 (lm_fused.hip): parity, then timing alone, then the pipelined bench A/B
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_lm_fused_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40
echo "=== alone"
timeout 300 python scripts/lm_engine_time.py 2>&1 | tail -14
echo "=== pipelined A/B"
for e in v2 v3 v2 v3; do
  ASTTS_LM_ENGINE=$e timeout 400 python bench.py --no-cpu-baseline --no-24khz --no-cobatch 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['value'], d['ms_per_step'], d.get('sequential_ms_per_step'))"
done
