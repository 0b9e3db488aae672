cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_final.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_final.json').read())
print({k: d[k] for k in ('value', 'ms_per_step', 'knn_qps', 'stages_ms', 'sequential_ms_per_step')})
print(d['roofline_by_stage']['flow']); print(d['value_24khz']['value'], d['cobatched_lm_side_measurement']['value'])
PY
