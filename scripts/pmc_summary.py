"""Summarise a rocprofv3 --pmc run: per kernel name, launches and mean counter value."""
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-48:]
        a = agg[k][r['Counter_Name']]
        a[0] += 1; a[1] += float(r['Counter_Value'])
for k, cs in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:14]:
    print(k, {c: (n, round(s / n, 2)) for c, (n, s) in cs.items()})
