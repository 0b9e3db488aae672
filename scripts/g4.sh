cd /root/repo; export TMPDIR=/tmp
rm -rf /tmp/trl && PROBE_ITERS=2 rocprofv3 --kernel-trace -d /tmp/trl -o tr --output-format csv -- python3 scripts/fullsize_probe.py > gpurun_out/lm_trace.log 2>&1
tail -2 gpurun_out/lm_trace.log
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/trl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# second iteration only: find last 250*73 decode kernels by taking lm_attn launches
att = [r for r in rows if 'lm_attn' in r['Kernel_Name']]
n = len(att) // 2
att = att[n:]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in att]
print('lm_attn launches', len(d), 'avg %.2f us' % (sum(d) / len(d)))
per_step = [sum(d[i * 14:(i + 1) * 14]) / 14 for i in range(len(d) // 14)]
print('by step: first 10 avg %.2f, steps 100-110 avg %.2f, last 10 avg %.2f' % (sum(per_step[:10]) / 10, sum(per_step[100:110]) / 10, sum(per_step[-10:]) / 10))
# gaps: kernel period in the decode chain
dec = [r for r in rows if any(k in r['Kernel_Name'] for k in ('lm_gemv', 'lm_attn', 'ras_sample'))]
dec = dec[len(dec) // 2:]
per = collections.defaultdict(list)
for a, b in zip(dec[:-1], dec[1:]):
    name = a['Kernel_Name'].split('(')[0][-40:]
    per[name].append(((int(a['End_Timestamp']) - int(a['Start_Timestamp'])) / 1e3, (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3))
for k, v in per.items():
    print(f'{k:42s} x{len(v):6d} dur {sum(x[0] for x in v) / len(v):6.2f} us, gap to next {sum(x[1] for x in v) / len(v):5.2f} us')
PY
