"""Generate golden input/output vectors by running the REFERENCE implementation (dev container only).

TEST INFRASTRUCTURE.  Run as  `python oracle/gen_golden.py [family ...]`  from the repo root; writes small
`.npz` fixtures (inputs + expected outputs, no reference source) under `tests/golden/`.  The reference tree
(`/root/reference`) cannot travel to the GPU box, the fixtures can.

Families
  env      SimulatedEnv(KuaishouEnv).step / KuaishouEnv.step, teacher-forced actions
           (reference core/env/simulatedEnv/simulated_env.py:111-168, environments/KuaishouRec/env/kuaishouEnv.py:161-218)
  gae      BasePolicy.compute_episodic_return / _gae_return (tianshou/policy/base.py:271-313,380-396)
  deepfm   UserModel_Pairwise.forward with the shipped weights + compute_normed_reward block
  tracker  StateTrackerTransformer.build_state over whole episodes (eval mode)
  policy   Actor/Critic forward + shared-noise sampling (ppo.py:111-163)
  learn    one full PPOPolicy.update on a recorded rollout (ppo.py:96-246)
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))

import ref_harness  # noqa: E402

ref_harness.install()

import pandas as pd  # noqa: E402
import torch  # noqa: E402
from sklearn.preprocessing import LabelEncoder  # noqa: E402

from cirs_hip.synthetic import make_tables  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLDEN, exist_ok=True)


# --------------------------------------------------------------------------------------------------
# helpers to build reference objects from synthetic tables
# --------------------------------------------------------------------------------------------------
def build_reference_env_kwargs(tab, num_leave_compute, leave_threshold, max_turn):
    lbe_user = LabelEncoder().fit(tab.raw_uid)
    lbe_photo = LabelEncoder().fit(tab.raw_pid)
    df_dist_small = pd.DataFrame(tab.dist, index=tab.raw_pid, columns=tab.raw_pid)
    df_photo_env = pd.DataFrame(
        np.where(tab.item_cats < 0, 0, tab.item_cats + 1), index=tab.raw_pid,
        columns=["feat0", "feat1", "feat2", "feat3"])
    df_photo_env.index.name = "photo_id"
    df_photo_env["photo_duration"] = tab.duration
    return dict(mat=tab.mat, lbe_user=lbe_user, lbe_photo=lbe_photo, list_feat=tab.list_feat,
                df_photo_env=df_photo_env, df_dist_small=df_dist_small,
                num_leave_compute=num_leave_compute, leave_threshold=leave_threshold, max_turn=max_turn)


def register_envs(tab, *, num_leave_compute, leave_threshold, max_turn, tau, gamma_exposure, version, r_decay,
                  with_ab, user_model=None):
    import gym
    kw = build_reference_env_kwargs(tab, num_leave_compute, leave_threshold, max_turn)
    gym.register(id="KuaishouEnv-v0", entry_point="environments.KuaishouRec.env.kuaishouEnv:KuaishouEnv", kwargs=kw)
    if user_model is None:
        user_model = torch.nn.Identity()
    gym.register(id="SimulatedEnv-v0", entry_point="core.env.simulatedEnv.simulated_env:SimulatedEnv",
                 kwargs=dict(user_model=user_model, task_name="KuaishouEnv-v0", version=version, tau=tau,
                             alpha_u=tab.alpha_u if with_ab else None, beta_i=tab.beta_i if with_ab else None,
                             normed_mat=tab.normed_mat, gamma_exposure=gamma_exposure, r_decay=r_decay))
    return kw


def adversarial_actions(rng, tab, n_env, max_turn):
    """Action sequences with many repeats / category collisions so exit rule and r_decay paths fire."""
    acts = np.zeros((n_env, max_turn), dtype=np.int64)
    for b in range(n_env):
        mode = b % 4
        if mode == 0:      # uniform over catalogue
            acts[b] = rng.randint(0, tab.n_items, size=max_turn)
        elif mode == 1:    # tiny pool -> repeats and same-category streaks
            pool = rng.randint(0, tab.n_items, size=3)
            acts[b] = pool[rng.randint(0, 3, size=max_turn)]
        elif mode == 2:    # items sharing the most popular category
            pop = [i for i in range(tab.n_items) if 0 in tab.item_cats[i]]
            pool = np.array(pop if len(pop) >= 2 else list(range(tab.n_items)))
            acts[b] = pool[rng.randint(0, len(pool), size=max_turn)]
        else:              # same item forever
            acts[b] = rng.randint(0, tab.n_items)
    return acts


# --------------------------------------------------------------------------------------------------
# env family
# --------------------------------------------------------------------------------------------------
def gen_env():
    import gym
    tab = make_tables(48, 96, seed=0, with_ab=True, build_dist=True)
    n_env = 16
    cases = []
    cfgs = []
    for N in (1, 2, 3, 5, 10):
        for thr in (0, 1, 4):
            cfgs.append(dict(num_leave_compute=N, leave_threshold=thr, max_turn=30, tau=10.0, gamma_exposure=10.0,
                             version="v1", r_decay=1.0, with_ab=True))
    cfgs += [
        dict(num_leave_compute=3, leave_threshold=4, max_turn=5, tau=0.0, gamma_exposure=10.0, version="v1", r_decay=1.0, with_ab=True),
        dict(num_leave_compute=10, leave_threshold=30, max_turn=100, tau=100.0, gamma_exposure=1.0, version="v1", r_decay=0.9, with_ab=True),
        dict(num_leave_compute=4, leave_threshold=30, max_turn=30, tau=0.1, gamma_exposure=10.0, version="v2", r_decay=1.0, with_ab=True),
        dict(num_leave_compute=4, leave_threshold=30, max_turn=30, tau=10.0, gamma_exposure=10.0, version="v2", r_decay=0.9, with_ab=False),
        dict(num_leave_compute=2, leave_threshold=30, max_turn=30, tau=10.0, gamma_exposure=1.0, version="v1", r_decay=1.0, with_ab=False),
    ]
    out = dict(mat=tab.mat, normed_mat=tab.normed_mat, dist=tab.dist, item_cats=tab.item_cats,
               raw_uid=tab.raw_uid, raw_pid=tab.raw_pid, alpha_u=tab.alpha_u, beta_i=tab.beta_i,
               n_cases=np.int64(len(cfgs)))
    for ci, cfg in enumerate(cfgs):
        rng = np.random.RandomState(1000 + ci)
        T = cfg["max_turn"]
        register_envs(tab, **cfg)
        acts = adversarial_actions(rng, tab, n_env, T)
        users = np.zeros(n_env, dtype=np.int64)
        obs = np.full((n_env, T), -1, dtype=np.int64)
        rew = np.full((n_env, T), np.nan)
        real_rew = np.full((n_env, T), np.nan)
        done = np.zeros((n_env, T), dtype=bool)
        ctr = np.full((n_env, T), np.nan)
        expo = np.full((n_env, T), np.nan)
        length = np.zeros(n_env, dtype=np.int64)
        # simulated env (training) and bare KuaishouEnv (test) driven with the same users/actions
        for b in range(n_env):
            random.seed(7000 + 100 * ci + b)
            env = gym.make("SimulatedEnv-v0")
            o0 = env.reset()
            users[b] = int(o0[0])
            random.seed(7000 + 100 * ci + b)
            real_env = gym.make("KuaishouEnv-v0")
            r0 = real_env.reset()
            assert int(r0[0]) == users[b]
            for t in range(T):
                a = np.int64(acts[b, t])
                o, r, d, info = env.step(a)
                ro, rr, rd, rinfo = real_env.step(a)
                assert bool(rd) == bool(d) and int(ro[0]) == int(o[0])
                obs[b, t] = int(o[0]); rew[b, t] = float(r); done[b, t] = bool(d)
                ctr[b, t] = float(info["CTR"]); expo[b, t] = float(env.history_exposure[t])
                real_rew[b, t] = float(rr)
                length[b] = t + 1
                if d:
                    break
        pre = f"c{ci}_"
        out[pre + "cfg"] = np.array([cfg["num_leave_compute"], cfg["leave_threshold"], cfg["max_turn"],
                                     cfg["tau"], cfg["gamma_exposure"], 1 if cfg["version"] == "v1" else 2,
                                     cfg["r_decay"], 1 if cfg["with_ab"] else 0], dtype=np.float64)
        for k, v in dict(users=users, acts=acts, obs=obs, rew=rew, real_rew=real_rew, done=done, ctr=ctr,
                         expo=expo, length=length).items():
            out[pre + k] = v
        cases.append(length.mean())
    np.savez_compressed(os.path.join(GOLDEN, "env_step.npz"), **out)
    print("env_step.npz: cases", len(cfgs), "mean episode lengths", np.round(cases, 1))


FAMILIES = {"env": gen_env}

if __name__ == "__main__":
    names = sys.argv[1:] or list(FAMILIES)
    for n in names:
        FAMILIES[n]()
