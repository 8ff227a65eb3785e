"""Print per-kernel stats from a rocprofv3 results .db (rocpd sqlite): python tools/kstats.py <db> [csv_out]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(lds_size) "
                       "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
if len(sys.argv) > 2 and sys.argv[2]:
    with open(sys.argv[2], "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 3), r[4], r[5]])
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r[0][:70]:70s} {r[1]:7d} {r[2]/1e6:9.2f}ms {r[3]/1e3:8.1f}us {100*r[2]/tot:5.1f}%  vgpr {r[6]} lds {r[7]}")
