"""Scratch probe: full step (collect + update) timing split at C2/C3 shapes."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np, torch
from cirs_hip.synthetic import make_tables
from cirs_hip.env import DeviceEnvTables
from cirs_hip.engine import CirsEngine

def main(U, I, B, T, iters=5):
    tab = make_tables(U, I, seed=0, build_dist=False)
    a_env = tab.alpha_u[tab.raw_uid, 0].astype(np.float64); b_env = tab.beta_i[tab.raw_pid, 0].astype(np.float64)
    dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, alpha_env=a_env, beta_env=b_env, build_dist_on_device=True)
    eng = CirsEngine(dt, B, max_turn=T, num_leave_compute=10, leave_threshold=4, tau=10.0, gamma_exposure=10.0)
    for w in range(2):
        eng.collect(); eng.update(1024, 2)
    torch.cuda.synchronize()
    tc = tu = 0.0; steps = 0; nmb = 0
    for k in range(iters):
        t0 = time.perf_counter(); lens = eng.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
        losses, n = eng.update(1024, 2); torch.cuda.synchronize(); t2 = time.perf_counter()
        tc += t1 - t0; tu += t2 - t1; steps += n; nmb += losses.shape[0]
    print(json.dumps(dict(U=U, I=I, B=B, ms_collect=1e3 * tc / iters, ms_update=1e3 * tu / iters, env_steps=steps, minibatches=nmb,
                          env_steps_per_s=steps / (tc + tu), us_per_minibatch=1e6 * tu / nmb, last_loss=losses[-1].tolist())), flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    if which == "c2": main(1411, 3327, 64, 30)
    elif which == "c4": main(7176, 10728, 8192, 30, iters=2)
    else: main(7176, 10728, 1024, 30)
