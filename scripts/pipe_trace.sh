cd /root/repo; export TMPDIR=/tmp
rm -rf /tmp/trp && rocprofv3 --kernel-trace -d /tmp/trp -o tr --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-24khz > gpurun_out/pipe_trace.json 2>/dev/null
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/trp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print('columns', list(rows[0].keys()))
# take the last 40% of the trace (timed region + side measurements vary): find the window where lm_gemv kernels from 2 queues overlap with tfm kernels
rows.sort(key=lambda r: int(r['Start_Timestamp']))
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
T0, T1 = int(rows[0]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
# window: middle 10% of launches
n = len(rows)
win = rows[int(n * 0.45):int(n * 0.55)]
w0, w1 = int(win[0]['Start_Timestamp']), int(win[-1]['End_Timestamp'])
print('window ms', (w1 - w0) / 1e6, 'launches', len(win))
byq = collections.defaultdict(list)
for r in win:
    byq[r[qkey]].append(r)
for q, rs in byq.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e6
    gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(rs[:-1], rs[1:])]
    names = collections.Counter(r['Kernel_Name'].split('(')[0][-30:] for r in rs).most_common(2)
    gs = sorted(gaps)
    print(f'queue {q}: {len(rs)} launches, busy {busy:.2f} ms, gaps avg {sum(gaps) / max(len(gaps), 1):.2f} us median {gs[len(gs) // 2] if gs else 0:.2f} us p90 {gs[int(len(gs) * 0.9)] if gs else 0:.2f}; top {names}')
PY
