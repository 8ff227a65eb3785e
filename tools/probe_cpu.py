"""Scratch probe: host-side enqueue time of collect()/update() vs the synchronised step time (is the CPU the bottleneck?)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np, torch
from cirs_hip.synthetic import make_tables
from cirs_hip.env import DeviceEnvTables
from cirs_hip.engine import CirsEngine

tab = make_tables(7176, 10728, seed=0, build_dist=False)
a_env = tab.alpha_u[tab.raw_uid, 0].astype(np.float64); b_env = tab.beta_i[tab.raw_pid, 0].astype(np.float64)
dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, alpha_env=a_env, beta_env=b_env, build_dist_on_device=True)
eng = CirsEngine(dt, 1024, max_turn=30, num_leave_compute=10, leave_threshold=4, tau=10.0, gamma_exposure=10.0)
for w in range(3):
    eng.collect(); eng.update(1024, 2)
torch.cuda.synchronize()
tc = tu = 0.0
t0 = time.perf_counter()
N = 20
for k in range(N):
    a = time.perf_counter(); eng.collect(); b = time.perf_counter(); eng.update(1024, 2); c = time.perf_counter()
    tc += b - a; tu += c - b
torch.cuda.synchronize()
t1 = time.perf_counter()
print(json.dumps(dict(ms_step=1e3 * (t1 - t0) / N, cpu_collect_ms=1e3 * tc / N, cpu_update_ms=1e3 * tu / N)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(5):
    eng.collect(); eng.update(1024, 2)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
