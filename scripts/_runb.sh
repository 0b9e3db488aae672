cd /root/repo
timeout 900 python -m pytest tests/test_cli_gpu.py -k "wires_its_frontend or checkpoint_directory" tests/test_frontend_nets_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^E |Error|Warning: astts" | tail -12
timeout 600 python bench.py --steps 8 --warmup 2 --no-side --no-cobatch --no-24khz --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err; python - <<'PY'
import json
r=json.load(open('/tmp/b.json'))
print('value',r['value'],'host_io',{k:r['host_io'][k] for k in ('value','ms_per_step','frontend_ms_per_prompt','frontend_ms_per_step_batched')})
PY
grep -i "graph" /tmp/b.err | head -3
