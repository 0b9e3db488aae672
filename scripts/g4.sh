cd /root/repo; export TMPDIR=/tmp
rm -rf /tmp/trg && rocprofv3 --kernel-trace -d /tmp/trg -o tr --output-format csv -- python3 scripts/gn_probe.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/trg/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'groupnorm_fused' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
for k, name in enumerate(('full kernel', 'return after the tile load', 'return after the statistics')):
    seg = d[k * 50 + 10:(k + 1) * 50]
    print(f'{name}: {sum(seg) / len(seg):.2f} us')
PY
