"""Diagnostic: cirs_ppo_learn (prefetch / no prefetch, fold / no fold) against one cirs_ppo_minibatch call per step; prints the first step whose
losses differ and the largest parameter difference.  python tools/probes/learn_loop_diag.py [I B T bs rep]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import numpy as np, torch
import test_gpu_learn as tl
from cirs_hip.rollout import Trajectory

I, B, T, bs, rep = [int(x) for x in (sys.argv[1:6] or [10728, 160, 30, 1024, 2])]
pp, lens, acts, rews, dones, obs, n, rng = tl._random_case(I, B, T, seed=I + bs)
value, logp = tl.rollout_time_value_logp(pp, obs, acts, lens)
perms = [rng.permutation(n) for _ in range(rep)]
hyper = [0.95, 0.95, 0.2, 0.25, 0.0, 0.5, 1e-3, bs, rep]

def run(step_calls, env):
    for k in ("CIRS_PPO_NO_FOLD", "CIRS_PPO_LEARN_PREFETCH", "CIRS_PPO_ROWS_KERNEL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    traj = Trajectory(B, T, 20, "cuda")
    tl.upload_traj(traj, acts, rews, dones, lens, obs, value, logp)
    ln, _ = tl.make_learner(pp, I, B, T, hyper)
    ln.prepare(traj, lens)
    losses = ln.learn(bs, rep, perms=perms, step_calls=step_calls)
    torch.cuda.synchronize()
    return losses.cpu().numpy(), ln.params.cpu().numpy(), ln.dobs.cpu().numpy()

def cmp(name, a, b):
    bad = np.where((a[0] != b[0]).any(1))[0]
    print(f"{name}: steps {a[0].shape[0]}, first differing step {bad[0] if len(bad) else None}, n differing {len(bad)}, max |dp| {np.abs(a[1] - b[1]).max():.3e}, "
          f"max |ddobs| {np.abs(a[2] - b[2]).max():.3e}", flush=True)

OLD = {"CIRS_PPO_ROWS_KERNEL": "0"}        # dh2_sum_kernel + trunk_bwd_kernel + sumsq_partial_kernel
s_rows = run(True, {})
cmp("steps/rows twice", s_rows, run(True, {}))
s_old = run(True, OLD)
cmp("old sequence: steps vs loop(prefetch)", s_old, run(False, OLD))
cmp("rows: steps vs loop(no prefetch)", s_rows, run(False, {"CIRS_PPO_LEARN_PREFETCH": "0"}))
cmp("rows: steps vs loop(prefetch)", s_rows, run(False, {}))
cmp("rows vs old sequence (steps)", s_rows, s_old)
