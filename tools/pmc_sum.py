"""Per-kernel average of rocprofv3 --pmc counters: python tools/pmc_sum.py <counter_collection.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"][:56]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
names = sorted({r["Counter_Name"] for r in rows})
for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
    print(f"{k:58s} " + "  ".join(f"{c} {agg[k][c] / max(cnt[k][c], 1):12.1f}" for c in names) + f"  n={max(cnt[k].values())}")
