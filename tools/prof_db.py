"""rocprofv3 (rocpd sqlite output) -> per-kernel summary CSV.  usage: python tools/prof_db.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.3f" % (100.0 * r[2] / tot), r[4], r[5]])
print("%d kernels, %.3f ms total" % (len(rows), tot / 1e6))
