"""Learning curves under the three dropout procedures of the state tracker, same synthetic task as
tests/test_gpu_plugin_surface.py::test_policy_learns_longer_and_more_rewarding_trajectories (300 users x 800 items, 128 envs, max_turn 30,
exit rule punishes repeated categories), CirsEngine on one MI355X:

  off      dropout 0 (the eval-mode tracker of the parity fixtures and of the headline bench line)
  sticky   dropout 0.1, production mode: a position keeps its masks for the rest of the episode (K/V-cached decode, csrc/rng.h)
  redraw   dropout 0.1, the reference's procedure: fresh masks over the whole prefix at every build_state call (cirs_hip/redraw.py)

    python tools/compare_dropout_modes.py [epochs] [seeds]      -> markdown table on stdout (mean trajectory length / reward per epoch)
    CMP_SHAPE=c2 ...                                            -> the same at BASELINE configs[1]'s shape (1411 x 3327, 64 envs)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np
import torch

from cirs_hip.engine import CirsEngine
from cirs_hip.env import DeviceEnvTables
from cirs_hip.synthetic import make_tables

EPOCHS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
SEEDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
U, I, B, T, STEPS_PER_EPOCH = 300, 800, 128, 30, 6000
if os.environ.get("CMP_SHAPE") == "c2":          # BASELINE configs[1]'s shape: 1411 users x 3327 items, 64 envs
    U, I, B = 1411, 3327, 64
tab = make_tables(U, I, seed=0, build_dist=True)
a_env = tab.alpha_u[tab.raw_uid, 0].astype(np.float64); b_env = tab.beta_i[tab.raw_pid, 0].astype(np.float64)
dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, dist=tab.dist, alpha_env=a_env, beta_env=b_env)


def run(mode, seed):
    eng = CirsEngine(dt, B, max_turn=T, num_leave_compute=3, leave_threshold=1, tau=10.0, gamma_exposure=10.0, seed=seed,
                     dropout=0.0 if mode == "off" else 0.1, dropout_redraw=(mode == "redraw"), batch_size_hint=1024)
    curve = []
    t0 = time.time()
    for ep in range(EPOCHS):
        collected = 0
        while collected < STEPS_PER_EPOCH:
            eng.collect()
            st = eng.collect_stats()
            collected += st["n/st"]
            eng.update(batch_size=1024, repeat=2)
        curve.append((st["len"], st["rew"]))
    torch.cuda.synchronize()
    return np.array(curve), time.time() - t0


res = {}
for mode in ("off", "sticky", "redraw"):
    runs = [run(mode, 3 + 17 * s) for s in range(SEEDS)]
    res[mode] = (np.mean([r[0] for r in runs], axis=0), np.std([r[0] for r in runs], axis=0), np.mean([r[1] for r in runs]))
print(f"# Tracker dropout procedures: learning curves ({SEEDS} seeds, {EPOCHS} epochs of >= {STEPS_PER_EPOCH} env-steps, {B} envs, {U}x{I}, max_turn {T})\n")
print("mean trajectory length / mean trajectory reward of the last training collect of each epoch (mean over seeds, +- std over seeds)\n")
print("| epoch | off: len | off: rew | sticky: len | sticky: rew | redraw: len | redraw: rew |\n|---|---|---|---|---|---|---|")
for ep in range(EPOCHS):
    row = [f"{ep + 1}"]
    for mode in ("off", "sticky", "redraw"):
        m, s, _ = res[mode]
        row += [f"{m[ep, 0]:.2f} +- {s[ep, 0]:.2f}", f"{m[ep, 1]:.2f} +- {s[ep, 1]:.2f}"]
    print("| " + " | ".join(row) + " |")
print("\nwall time per run (s): " + ", ".join(f"{mode} {res[mode][2]:.1f}" for mode in res))
