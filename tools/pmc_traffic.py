"""HBM traffic per kernel launch from rocprofv3 PMC passes -> profiles/pmc_traffic.json (read by bench.py) + a markdown table.

    python tools/pmc_traffic.py collect [tag]     on the GPU box: two SEPARATE passes (FETCH_SIZE, then WRITE_SIZE; never together,
                                                  never with other trace domains) of tools/pmc_workload.py, parsed into
                                                  gpurun_out/pmc_traffic.json and gpurun_out/<tag>_pmc_traffic.md
    python tools/pmc_traffic.py parse <fetch.csv> <write.csv> [tag]     re-parse existing counter_collection CSVs

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): FETCH_SIZE is doubled per the gfx950 wide-read correction of
MI355X_MICROARCH.md (the counter tallies 128-B requests at 64 B); WRITE_SIZE as reported.  Write-back caching smears WRITE_SIZE over
the kernels that follow a producer, so per-step sums are more meaningful than single rows.  The JSON records the fingerprint of the
kernel sources it was taken on; bench.py refuses to quote it for any other build (no stale constants).  Copy the two files into
profiles/ and commit them."""
import collections
import csv
import datetime
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    return name.replace("cirs::", "")


def per_kernel(csv_path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(csv_path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def parse(fetch_csv, write_csv, tag):
    import bench
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        if k.startswith("at::") or "at::native" in k or k.startswith("rccl") or "elementwise" in k:
            continue
        fk, wk = f.get(k, (0.0, 0))[0], w.get(k, (0.0, 0))[0]
        kernels[k] = {"fetch_kb": round(fk, 1), "write_kb": round(wk, 1), "launches": max(f.get(k, (0, 0))[1], w.get(k, (0, 0))[1]),
                      "bytes_per_launch": int((2 * fk + wk) * 1024)}
    out = {"source_hash": bench.kernel_source_hash(), "workload": os.environ.get("CIRS_PMC_WORKLOAD", "c3"), "tag": tag,
           "taken": datetime.date.today().isoformat(), "formula": "bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024",
           "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python tools/pmc_workload.py", "kernels": kernels}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_traffic.md"), "w") as fh:
        fh.write(f"# PMC passes {tag}: HBM traffic per launch (kernel sources {out['source_hash']}, workload {out['workload']})\n\n"
                 f"`{out['command']}` — two separate runs.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE.\n\n"
                 "| kernel | launches | FETCH_SIZE KB | WRITE_SIZE KB | HBM MB / launch |\n|---|---|---|---|---|\n")
        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"]):
            fh.write(f"| `{k}` | {v['launches']} | {v['fetch_kb']} | {v['write_kb']} | {v['bytes_per_launch'] / 1e6:.2f} |\n")
        mb = [kernels[k]["bytes_per_launch"] for k in bench.MINIBATCH_KERNELS if k in kernels]
        if len(mb) == len(bench.MINIBATCH_KERNELS):
            fh.write(f"\nPPO minibatch step ({len(mb)} kernels): **{sum(mb) / 1e6:.1f} MB**\n")
    print(open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_traffic.md")).read())


def collect(tag):
    outs = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{ctr}")
        cmd = ["timeout", "600", "rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py")]
        env = dict(os.environ, TMPDIR="/tmp")
        subprocess.run(cmd, check=True, cwd="/tmp", env=env)
        found = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        assert found, f"no counter_collection.csv under {d}"
        outs[ctr] = found[0]
    parse(outs["FETCH_SIZE"], outs["WRITE_SIZE"], tag)
    # the raw CSVs are large: keep only the parsed summary
    for ctr in outs:
        for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{ctr}", "**", "*.csv"), recursive=True):
            os.remove(f)


if __name__ == "__main__":
    if sys.argv[1] == "collect":
        collect(sys.argv[2] if len(sys.argv) > 2 else "r02")
    else:
        parse(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "r02")
