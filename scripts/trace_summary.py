"""Aggregate a rocprofv3 kernel trace by (kernel, grid, block): count, avg us, total ms, share."""
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict(); tot = 0.0
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = re.sub(r'^void |astts::', '', r['Kernel_Name']).split('(')[0][:70]
    k = (name, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']) // max(int(r['Workgroup_Size_Y']), 1), int(r['Workgroup_Size_X']))
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d; tot += d
span = (max(int(r['End_Timestamp']) for r in rows) - min(int(r['Start_Timestamp']) for r in rows)) / 1e6
print(f'{len(rows)} launches, busy {tot / 1e3:.1f} ms, span {span:.1f} ms')
for (n, gx, gy, bs), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f'  {n:70s} grid {gx:5d}x{gy:<4d} blk {bs:4d}  x{c:5d}  avg {t / c:8.2f} us  total {t / 1e3:7.2f} ms  {100 * t / tot:5.1f}%')
