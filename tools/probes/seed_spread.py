"""The driver protocol (5 warm-up + 20 timed steps, C3) under several engine seeds: how much of the 20-step value is trajectory luck.
    python tools/probes/seed_spread.py"""
import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np, torch, bench
from cirs_hip.engine import CirsEngine
from cirs_hip.env import DeviceEnvTables
from cirs_hip.synthetic import make_tables
wl = bench.WORKLOADS["c3"]; dev = torch.device("cuda:0")
tab = make_tables(wl["U"], wl["I"], seed=0, build_dist=False)
a_env = tab.alpha_u[tab.raw_uid, 0].astype(np.float64); b_env = tab.beta_i[tab.raw_pid, 0].astype(np.float64)
dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, alpha_env=a_env, beta_env=b_env, device=dev, build_dist_on_device=True)
for seed in (2023, 1, 7, 99, 777):
    eng = CirsEngine(dt, wl["B"], max_turn=wl["T"], num_leave_compute=wl["N"], leave_threshold=wl["thr"], tau=wl["tau"], gamma_exposure=wl["gamma_exposure"], seed=seed)
    for _ in range(5):
        eng.collect(); eng.update(1024, 2)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n_tot = 0
    for _ in range(20):
        eng.collect(); l, n = eng.update(1024, 2); n_tot += n
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"seed {seed}: env-steps {n_tot}  ms/step {1e3*el/20:.3f}  value {n_tot/el/1e6:.3f} M  us/env-step {1e6*el/n_tot:.4f}", flush=True)
    del eng
