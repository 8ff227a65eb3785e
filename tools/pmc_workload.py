"""Workload of the PMC passes (tools/pmc_traffic.py): a few C3 bench steps (rollout + PPO update), the K1-K2 gather+FM kernel at
the micro-benchmark's shapes and the DeepFM catalogue sweep / sweep-mode step -- every kernel whose HBM traffic bench.py quotes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

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
print("pmc workload done", flush=True)
