"""Workload of the PMC passes (tools/pmc_traffic.py): a few C3 bench steps (rollout + PPO update), the K1-K2 gather+FM kernel at the micro-benchmark's shapes (one dispatch group per case, in the order of bench.gather_fm_probe), the
DeepFM catalogue sweep / sweep-mode step -- every kernel whose HBM traffic bench.py quotes -- and three KNOWN-BYTES calibration launches
of cirs_gather_rows over a 512 MiB table (far past L2 and the 256 MiB Infinity Cache): a streaming read of 256-byte rows, random 256-byte
rows and random 128-byte rows.  Their counter values fix, per access pattern, the factor between FETCH_SIZE and bytes on this box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from cirs_hip.sharded import hip_gather_rows  # noqa: E402

CAL_ROWS = 1 << 21      # launches per calibration case: 1; rows gathered per launch

wl = bench.WORKLOADS[os.environ.get("CIRS_PMC_WORKLOAD", "c3")]
dev = torch.device("cuda:0")
eng, _ = bench.build_engine(wl, 0, 1, dev)
for _ in range(3):
    eng.collect()
    eng.update(1024, 2)
torch.cuda.synchronize()
bench.gather_fm_probe(dev, reps=3)
bench.deepfm_sweep_probe(wl, dev, reps=2)
bench.sweep_mode_probe(wl, eng, dev, reps=2)
torch.cuda.synchronize()
# ---- calibration: gather_rows_kernel, dispatch order = (seq 256 B, random 256 B, random 128 B); bytes read = rows x row bytes (+ 8 B index),
# bytes written = rows x row bytes
g = torch.Generator(device=dev).manual_seed(7)
for row_floats, order in ((64, "seq"), (64, "rand"), (32, "rand")):
    n_table = (512 << 20) // (4 * row_floats)
    table = torch.empty((n_table, row_floats), dtype=torch.float32, device=dev).normal_(generator=g)
    idx = (torch.arange(CAL_ROWS, device=dev, dtype=torch.int64) if order == "seq"
           else torch.randint(0, n_table, (CAL_ROWS,), device=dev, generator=g, dtype=torch.int64))
    torch.cuda.synchronize()
    out = hip_gather_rows(table, idx)
    torch.cuda.synchronize()
    del table, idx, out
    torch.cuda.empty_cache()
print("pmc workload done", flush=True)
