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


def per_dispatch(csv_path, counter, prefix):
    """Counter values of the kernels whose short name starts with `prefix`, in dispatch order."""
    rows = [(int(r.get("Dispatch_Id") or r.get("Correlation_Id") or i), short(r["Kernel_Name"]), float(r["Counter_Value"]))
            for i, r in enumerate(csv.DictReader(open(csv_path))) if r["Counter_Name"] == counter and short(r["Kernel_Name"]).startswith(prefix)]
    return [(k, v) for _, k, v in sorted(rows)]


CAL_ROWS = 1 << 21   # tools/pmc_workload.py


def calibration(fetch_csv, write_csv):
    """Known-bytes launches of gather_rows_kernel (the LAST three dispatches of that kernel: streaming 256-B rows, random 256-B rows,
    random 128-B rows over a 512 MiB table) -> bytes per FETCH_SIZE KB / WRITE_SIZE KB for each access pattern."""
    f, w = per_dispatch(fetch_csv, "FETCH_SIZE", "gather_rows_kernel"), per_dispatch(write_csv, "WRITE_SIZE", "gather_rows_kernel")
    if len(f) < 3 or len(w) < 3:
        return None
    out = {}
    for (name, row_bytes), (_, fk), (_, wk) in zip((("stream_256B", 256), ("random_256B", 256), ("random_128B", 128)), f[-3:], w[-3:]):
        rd, wr = CAL_ROWS * (row_bytes + 8), CAL_ROWS * row_bytes
        out[name] = {"known_read_bytes": rd, "known_write_bytes": wr, "fetch_kb": round(fk, 1), "write_kb": round(wk, 1),
                     "fetch_factor": rd / (fk * 1024) if fk else None, "write_factor": wr / (wk * 1024) if wk else None}
    return out


def gather_fm_cases(fetch_csv, write_csv, cal):
    """One row per CASE of bench.gather_fm_probe (dispatch order: c3_E32, c3_E16, mid_E64, c5_E64, 1 + reps launches each), not per template
    instantiation: the two E = 64 cases (tables inside / past the Infinity Cache) get their own bytes.  Physical bytes with the calibrated
    factors: the X / out streams are wide coalesced traffic (stream factor); what a table row costs beyond them uses the random-row factor
    of its width."""
    f, w = per_dispatch(fetch_csv, "FETCH_SIZE", "gather_fm_kernel"), per_dispatch(write_csv, "WRITE_SIZE", "gather_fm_kernel")
    names = ("c3_E32", "c3_E16", "mid_E64", "c5_E64")
    if len(f) % 4 or len(f) != len(w) or not f:
        return None
    per = len(f) // 4
    out = {}
    for i, nm in enumerate(names):
        fk = sum(v for _, v in f[i * per + 1:(i + 1) * per]) / max(per - 1, 1)       # (the first launch of a case is its warm-up)
        wk = sum(v for _, v in w[i * per + 1:(i + 1) * per]) / max(per - 1, 1)
        rec = {"kernel": f[i * per][0], "fetch_kb": round(fk, 1), "write_kb": round(wk, 1), "bytes_doubled_fetch": int((2 * fk + wk) * 1024)}
        if cal:
            E = int(nm.split("E")[1])
            key = "random_256B" if 4 * E + 4 > 192 else "random_128B"
            ff, wf = cal[key]["fetch_factor"], cal["stream_256B"]["write_factor"]
            if ff and wf:
                rec["bytes_calibrated"] = int(fk * 1024 * ff + wk * 1024 * wf)
                rec["calibration"] = f"FETCH_SIZE x {ff:.2f} ({key}), WRITE_SIZE x {wf:.2f} (stream_256B)"
        out[nm] = rec
    return out


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
    out["calibration"] = calibration(fetch_csv, write_csv)
    out["gather_fm_cases"] = gather_fm_cases(fetch_csv, write_csv, out["calibration"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_traffic.md"), "w") as fh:
        fh.write(f"# PMC passes {tag}: HBM traffic per launch (kernel sources {out['source_hash']}, workload {out['workload']})\n\n"
                 f"`{out['command']}` — two separate runs.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE.\n\n"
                 "| kernel | launches | FETCH_SIZE KB | WRITE_SIZE KB | HBM MB / launch |\n|---|---|---|---|---|\n")
        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"]):
            fh.write(f"| `{k}` | {v['launches']} | {v['fetch_kb']} | {v['write_kb']} | {v['bytes_per_launch'] / 1e6:.2f} |\n")
        for label, names in (("fused head", bench.MINIBATCH_KERNELS), ("split head", bench.MINIBATCH_KERNELS_SPLIT)):
            mb = [next((v["bytes_per_launch"] for kk, v in kernels.items() if kk.split("<")[0] == k), None) for k in names]
            if all(x is not None for x in mb):
                fh.write(f"\nPPO minibatch step, {label} ({len(mb)} kernels; the shared ones are averaged over both paths' launches): **{sum(mb) / 1e6:.1f} MB**\n")
        if out["calibration"]:
            fh.write("\n## Calibration (known bytes: gather_rows_kernel over a 512 MiB table, 2^21 rows per launch)\n\n| pattern | known read MB | FETCH_SIZE KB | bytes per counted KB x 1024 | known write MB | WRITE_SIZE KB | factor |\n|---|---|---|---|---|---|---|\n")
            for k, c in out["calibration"].items():
                fh.write(f"| {k} | {c['known_read_bytes'] / 1e6:.1f} | {c['fetch_kb']} | {c['fetch_factor']:.3f} | {c['known_write_bytes'] / 1e6:.1f} | {c['write_kb']} | {c['write_factor']:.3f} |\n")
        if out["gather_fm_cases"]:
            fh.write("\n## K1-K2 per case (bench.gather_fm_probe order)\n\n| case | kernel | FETCH_SIZE KB | WRITE_SIZE KB | 2 x FETCH + WRITE MB | calibrated MB |\n|---|---|---|---|---|---|\n")
            for k, c in out["gather_fm_cases"].items():
                fh.write(f"| {k} | `{c['kernel']}` | {c['fetch_kb']} | {c['write_kb']} | {c['bytes_doubled_fetch'] / 1e6:.1f} | {c.get('bytes_calibrated', 0) / 1e6:.1f} |\n")
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
