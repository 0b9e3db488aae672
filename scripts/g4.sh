cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x --tb=short -k "tfm_attn" 2>&1 | tail -1
ASTTS_TFM_ATTN_SPLIT=2 timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x --tb=short -k "tfm_attn" 2>&1 | tail -1
for sp in 1 2 1 2; do ASTTS_TFM_ATTN_SPLIT=$sp timeout 900 python bench.py --no-cpu-baseline --no-24khz --no-cobatch > gpurun_out/r03_bench5.json 2> gpurun_out/r03_bench5.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r03_bench5.json').read().strip().splitlines()[-1])
print("split $sp:", {k:d[k] for k in ('value','ms_per_step','stages_ms')})
PY
done
