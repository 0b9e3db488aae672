"""Summarise one LM decode step from a rocprofv3 kernel trace: per kernel (name, grid) avg duration and
the wall span between two consecutive ras_sample launches."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'ras_sample' in r['Kernel_Name']]
i0, i1 = idx[len(idx)//2], idx[len(idx)//2 + 1]
span = (int(rows[i1]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3
agg = collections.OrderedDict()
busy = 0.0
for r in rows[i0:i1]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    busy += d
    k = (r['Kernel_Name'].split('(')[0][-40:], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
print(f'step span {span:.1f} us, kernel busy {busy:.1f} us, launches {i1 - i0}')
for (n, g), (c, t) in agg.items():
    print(f'  {n:42s} blocks {g:6d}  x{c:3d}  avg {t / c:7.2f} us  total {t:8.1f}')
