"""SQ / instruction-cache counters of the rollout's kernels: python tools/pmc_rollout.py <tag> CTR1,CTR2,.. [c3|c2]   (on the GPU box)
One rocprofv3 --pmc pass (with --kernel-trace only) over a few forced-length collects; per-kernel sums / averages -> gpurun_out/<tag>_pmc_rollout.md."""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if os.environ.get("CIRS_PMC_WORKER"):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
    import torch
    import bench
    wl = bench.WORKLOADS[os.environ.get("CIRS_PMC_WL", "c3")]
    eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"), dropout=0.1)
    eng.rollout.force_length = wl["T"]
    for _ in range(3):
        eng.collect()
    torch.cuda.synchronize()
    sys.exit(0)

tag, ctrs = sys.argv[1], sys.argv[2].split(",")
wl = sys.argv[3] if len(sys.argv) > 3 else "c3"
d = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
cmd = ["timeout", "600", "rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__)]
subprocess.run(cmd, check=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", CIRS_PMC_WORKER="1", CIRS_PMC_WL=wl))
found = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
assert found, "no counter_collection.csv"
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(found[0])):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == ctrs[0]:
        n[k] += 1
with open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_rollout.md"), "w") as f:
    f.write(f"per-launch averages, workload {wl}, counters {ctrs}\n\n| kernel | launches | " + " | ".join(ctrs) + " |\n|---|---|" + "---|" * len(ctrs) + "\n")
    for k in sorted(acc, key=lambda k: -n[k]):
        if n[k] >= 10:
            f.write(f"| {k} | {n[k]} | " + " | ".join(f"{acc[k][c] / n[k]:.0f}" for c in ctrs) + " |\n")
print(open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_rollout.md")).read())
