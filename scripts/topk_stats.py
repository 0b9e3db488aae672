import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
iters = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
tot = 0.0
rows = list(csv.DictReader(open(f)))
for r in rows:
    tot += int(r['TotalDurationNs'])
print(f'total kernel time per iter {tot / iters / 1e6:.1f} ms')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{r['Name'][:70]:70s} calls/iter {int(r['Calls']) / iters:8.0f} ms/iter {int(r['TotalDurationNs']) / iters / 1e6:7.2f} avg {float(r['AverageNs']) / 1e3:7.1f} us")
