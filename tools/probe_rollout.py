"""Scratch probe (not the contract bench): rollout-only timing at C2/C3 shapes."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rolloutcase
from cirs_hip.synthetic import make_tables

def main(U, I, B, T, iters=5, sync_every=None):
    t0 = time.time()
    tab = make_tables(U, I, seed=0, build_dist=False)
    ro, tp, arrs, envp = rolloutcase.build_device_stack(tab, B, T)
    users = torch.as_tensor(np.random.RandomState(1).randint(0, U, B)).cuda()
    print(f"setup {time.time()-t0:.1f}s", flush=True)
    for w in range(2):
        lengths = ro.collect(users, seed=1, rng_base=w * T, sync_every=sync_every)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); steps = 0
    for k in range(iters):
        lengths = ro.collect(users, seed=1, rng_base=(k + 2) * T, sync_every=sync_every)
        steps += int(lengths.sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(U=U, I=I, B=B, T=T, sync_every=sync_every, env_steps=steps, sec=dt, env_steps_per_s=steps / dt,
                          ms_per_collect=1e3 * dt / iters, mean_len=float(lengths.float().mean()))), flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    se = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "none" else None
    if which == "c2": main(1411, 3327, 64, 30, sync_every=se)
    else: main(7176, 10728, 1024, 30, sync_every=se)
