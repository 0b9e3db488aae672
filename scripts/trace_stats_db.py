"""Per-kernel totals of a rocprofv3 --kernel-trace results database (sqlite): python scripts/trace_stats_db.py <results.db> [out.csv] [top_n].
Written for traces too large to bring back from the GPU box (a config-4 pass is millions of launches): aggregate there, keep the CSV."""
import csv
import re
import sqlite3
import sys

db, out, top = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None), int(sys.argv[3]) if len(sys.argv) > 3 else 40
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = con.execute(f"select s.kernel_name, count(*), sum(k.end - k.start), avg(k.end - k.start), min(k.end - k.start), max(k.end - k.start) "
                   f"from {kd} k join {ks} s on k.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
if out:
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, c, t, a, mn, mx in rows:
            w.writerow([n, c, t, f"{a:.1f}", f"{100.0 * t / tot:.3f}", mn, mx])
print(f"{len(rows)} kernels, {sum(r[1] for r in rows)} launches, {tot / 1e6:.1f} ms of kernel time")
for n, c, t, a, mn, mx in rows[:top]:
    print(f"  {re.sub(r'[(].*', '', n)[:86]:86s} {c:8d} {a / 1e3:9.1f} us {t / 1e6:10.1f} ms {100.0 * t / tot:5.1f}%")
