"""A/B timing of the PPO minibatch kernels for two builds of the library on the SAME box:  python tools/ab_headbwd.py <lib.so> [reps]
Prints the library's HIP-event averages of head_bwd_fused_kernel / head_stats_kernel and the whole cirs_ppo_minibatch call (C3, 1024 rows)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import torch
from cirs_hip import abi

abi.LIB_PATH = os.path.abspath(sys.argv[1])
import bench

reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 3     # updates before timing: a trained policy has clamped probabilities (p < eps)
wl = bench.WORKLOADS["c3"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"))
for _ in range(pre):
    eng.collect(); eng.update(1024, 2)
eng.collect(); eng.learner.prepare(eng.rollout.traj, eng.lengths.cpu().numpy(), lens_dev=eng.lengths)
out = []
for k in range(3):
    t_mb, mb, t_k = bench.hip_event_kernel_time(eng, wl, reps=reps)
    out.append((t_k["head_bwd_fused_kernel"] * 1e6, t_k["head_stats_kernel"] * 1e6, t_mb * 1e6))
print(os.path.basename(sys.argv[1]), " | ".join("bwd %.2f stats %.2f step %.2f" % o for o in out), flush=True)
