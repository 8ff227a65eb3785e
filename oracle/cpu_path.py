"""CPU port of one full hot-path step assembled from the oracle pieces.  TEST INFRASTRUCTURE / bench.py cpu_baseline leg.

One step = Collector.collect(n_episode = B)  +  policy.update(0, buffer, batch_size, repeat), structured like the
reference executes it (core/collector.py:219-317): per vector step
    policy forward + sample  (C oracle, OpenMP over rows)                -- ppo.py:111-163
    env.step for every live env (C oracle, one env after another)        -- simulated_env.py:111-145
    state tracker rebuilt over the whole prefix (torch fp32, O(L^2))     -- state_tracker.py:225-248
then the torch-fp32 PPO update restatement with autograd through the tracker (ppo.py:166-246).
It is a *port* used as the reported CPU baseline; it is never part of the product path.
"""
import time

import numpy as np
import torch

import envcase
import nn_oracle
import policycase


def run_cpu_step(tab, tp, arrs, B, T, *, N=10, thr=4, tau=10.0, gamma_exposure=10.0, batch_size=1024, repeat=2, seed=0,
                 dist_mode=1, do_update=True):
    """-> dict(env_steps, t_collect, t_update, minibatches).  tp: tracker params (torch, reference names); arrs: policy
    arrays keyed w1..bc (numpy)."""
    U, I = tab.n_users, tab.n_items
    a_env, b_env = envcase.ab_env_tables(tab.raw_uid, tab.raw_pid, tab.alpha_u, tab.beta_i, U, I)
    cfg = envcase.env_cfg(U, I, dist_mode=dist_mode, num_leave_compute=N, leave_threshold=thr, max_turn=T, tau=tau,
                          gamma_exposure=gamma_exposure, version=1, r_decay=1.0, has_ab=True)
    env = envcase.HostEnv(cfg, tab.mat, tab.normed_mat, tab.dist if dist_mode == 0 else None, tab.item_cats, a_env, b_env, B)
    rng = np.random.RandomState(seed)
    users = rng.randint(0, U, B)
    t0 = time.perf_counter()
    env.reset(users)
    acts = np.full((B, T), -1, np.int64); rews = np.zeros((B, T)); dones = np.zeros((B, T), bool)
    lens = np.zeros(B, np.int64)
    with torch.no_grad():
        x0 = nn_oracle.tracker_inputs(tp, users, np.zeros((B, 0), np.int64), np.zeros((B, 0)))
        state = nn_oracle.tracker_forward_all(tp, x0)[:, -1].numpy()
    ready = np.arange(B)
    for t in range(T):
        if len(ready) == 0:
            break
        a, _, _, _ = policycase.oracle_sample(arrs, state, seed=seed, rng_step=t, env_ids=ready.astype(np.int32))
        o, r, d, c, _ = env.step(a, ready)
        acts[ready, t] = a; rews[ready, t] = r; dones[ready, t] = d; lens[ready] = t + 1
        with torch.no_grad():  # whole-prefix recompute for the live envs, like the reference's build_state
            xs = nn_oracle.tracker_inputs(tp, users[ready], acts[ready, :t + 1], rews[ready, :t + 1])
            s_all = nn_oracle.tracker_forward_all(tp, xs)[:, -1].numpy()
        keep = ~d
        ready = ready[keep]
        state = s_all[keep]
    t1 = time.perf_counter()
    out = dict(env_steps=int(lens.sum()), t_collect=t1 - t0, t_update=0.0, minibatches=0)
    if do_update:
        n = int(lens.sum())
        perms = [np.random.RandomState(seed + 1 + k).permutation(n) for k in range(repeat)]
        tpc = {k: v.clone() for k, v in tp.items()}
        pp = {k: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)) for k, v in arrs.items()}
        res = nn_oracle.ppo_update(tpc, pp, users, acts, rews, dones, lens, perms, gamma=0.95, lam=0.95, eps_clip=0.2,
                                   vf_coef=0.25, ent_coef=0.0, max_grad_norm=0.5, lr=1e-3, batch_size=batch_size, repeat=repeat)
        out["t_update"] = time.perf_counter() - t1
        out["minibatches"] = len(res["loss"])
    return out
