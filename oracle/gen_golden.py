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
  collectorset   test_episode + CollectorSet.collect (FB / NX_0 / NX_k test collectors) under teacher-forced sampler noise
                 (tianshou/trainer/utils.py:10-31, core/collector_set.py:13-77, core/policy/utils.py:7-58)
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
# NOTE: cirs-codes_amd/ must NOT be on sys.path here: its mirror packages (`environments`, `core`, ...) carry the
# reference's module names and `environments` (a namespace package in the reference) would shadow the reference's.

import ref_harness  # noqa: E402

ref_harness.install()

import pandas as pd  # noqa: E402
import torch  # noqa: E402
from sklearn.preprocessing import LabelEncoder  # noqa: E402

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("cirs_synthetic", os.path.join(ROOT, "cirs-codes_amd", "cirs_hip", "synthetic.py"))
_syn = importlib.util.module_from_spec(_spec)
sys.modules["cirs_synthetic"] = _syn
_spec.loader.exec_module(_syn)
make_tables = _syn.make_tables

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


# --------------------------------------------------------------------------------------------------
# tracker family: StateTrackerTransformer.build_state, eval mode (dropout off)
# --------------------------------------------------------------------------------------------------
def make_reference_tracker(n_users, n_items, max_turn, seed, dim_model=32, dim_state=20, nhead=4, randomize=True):
    from core.inputs import SparseFeatP
    from core.state_tracker import StateTrackerTransformer
    from deepctr_torch.inputs import DenseFeat
    user_columns = [SparseFeatP("feat_user", n_users, embedding_dim=dim_model)]
    action_columns = [SparseFeatP("feat_item", n_items, embedding_dim=dim_model)]
    feedback_columns = [DenseFeat("feat_feedback", 1)]
    st = StateTrackerTransformer(user_columns, action_columns, feedback_columns, dim_model=dim_model,
                                 dim_state=dim_state, dim_max_batch=64, dataset="KuaishouEnv-v0",
                                 has_user_embedding=False, has_action_embedding=False, has_feedback_embedding=True,
                                 nhead=nhead, d_hid=128, nlayers=2, dropout=0.1, device="cpu", seed=seed,
                                 MAX_TURN=max_turn)
    if randomize:  # the stock init (N(0,1e-4) embeddings) makes the output almost input-independent; stress it
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for name, p in st.named_parameters():
                if "norm" in name and name.endswith("weight"):
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                elif "embedding_dict" in name:
                    p.copy_(0.5 * torch.randn(p.shape, generator=g))
                elif p.dim() >= 2:
                    p.copy_(torch.randn(p.shape, generator=g) * (1.5 / np.sqrt(p.shape[1])))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return st


def gen_tracker():
    U, I, B, T = 48, 96, 8, 12
    st = make_reference_tracker(U, I, T, seed=11)
    st.eval()
    rng = np.random.RandomState(5)
    users = rng.randint(0, U, size=B)
    acts = rng.randint(0, I, size=(B, T))
    rews = rng.uniform(0, 1, size=(B, T))
    # live-set schedule: envs 5..7 stop after turn 4, envs 2..4 after turn 8 (collector drops finished envs)
    last_turn = np.array([T, T, 8, 8, 8, 4, 4, 4])
    states = np.full((B, T + 1, 20), np.nan, dtype=np.float32)
    with torch.no_grad():
        st.build_state(dim_batch=B, reset=True)
        s0 = st.build_state(obs=users.reshape(-1, 1), env_id=np.arange(B))["obs"]
        states[:, 0] = s0.numpy()
        for t in range(T):
            live = np.where(last_turn > t)[0]
            out = st.build_state(obs_next=acts[live, t].reshape(-1, 1), rew=rews[live, t], done=None, info=None,
                                 policy=None, env_id=live)["obs_next"]
            states[live, t + 1] = out.numpy()
    sd = {"sd_" + k: v.detach().numpy() for k, v in st.state_dict().items()}
    np.savez_compressed(os.path.join(GOLDEN, "tracker.npz"), users=users, acts=acts, rews=rews, last_turn=last_turn,
                        states=states, dims=np.array([U, I, B, T, 32, 20, 4, 128, 2]), **sd)
    print("tracker.npz: states", states.shape, "finite", np.isfinite(states).mean())


# --------------------------------------------------------------------------------------------------
# policy family: Net/Actor/Critic forward + Categorical sampling under harness-supplied noise
# --------------------------------------------------------------------------------------------------
def make_reference_policy(n_items, seed, dim_state=20, hidden=(64, 64), lr=1e-3, **ppo_kw):
    from core.policy.ppo import PPOPolicy
    from tianshou.utils.net.common import Net
    from tianshou.utils.net.discrete import Actor, Critic
    import gym
    torch.manual_seed(seed)
    net = Net(dim_state, hidden_sizes=list(hidden), device="cpu")
    actor = Actor(net, n_items, device="cpu")
    critic = Critic(net, device="cpu")
    for m in list(actor.modules()) + list(critic.modules()):  # CIRS-RL-kuaishou.py:250-254
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    return net, actor, critic


def gen_policy():
    I, B = 96, 24
    net, actor, critic = make_reference_policy(I, seed=3)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():  # non-zero biases + a sharper head so probabilities are far from uniform
        for p in list(actor.parameters()) + list(critic.parameters()):
            if p.dim() == 1:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
        actor.last.model[0].weight.mul_(4.0)
    s = torch.randn(B, 20, generator=g)
    q = torch.empty(B, I).exponential_(1.0, generator=g)  # torch.multinomial's race noise
    visited = np.stack([np.random.RandomState(100 + b).choice(I, size=7, replace=False) for b in range(B)])
    with torch.no_grad():
        probs, _ = actor(s)
        value = critic(s).flatten()
        act = torch.argmax(probs / q, dim=-1)  # == multinomial(probs, 1) with exponential noise q
        dist = torch.distributions.Categorical(probs)
        logp = dist.log_prob(act)
        ent = dist.entropy()
        # masked path (core/policy/utils.py:30-58 + ppo.py:141-159): drop visited ids, renormalise, sample, map back
        from core.policy.utils import removed_recommended_id_from_embedding
        pm, idxm = removed_recommended_id_from_embedding(probs, visited)
        qm = q.masked_select(torch.ones_like(probs, dtype=torch.bool).scatter(1, torch.as_tensor(visited), 0)).reshape(B, -1)
        act_m_local = torch.argmax(pm / qm, dim=-1)
        act_m = idxm.gather(1, act_m_local.unsqueeze(-1)).squeeze(1)
    sd = {}
    for k, v in actor.state_dict().items():
        sd["actor_" + k] = v.numpy()
    for k, v in critic.state_dict().items():
        sd["critic_" + k] = v.numpy()
    top2 = torch.topk(torch.log(probs) - torch.log(q), 2, dim=-1).values
    np.savez_compressed(os.path.join(GOLDEN, "policy.npz"), s=s.numpy(), q=q.numpy(), visited=visited,
                        probs=probs.numpy(), value=value.numpy(), act=act.numpy(), logp=logp.numpy(),
                        ent=ent.numpy(), act_masked=act_m.numpy(), margin=(top2[:, 0] - top2[:, 1]).numpy(), **sd)
    print("policy.npz: min top-2 margin", float((top2[:, 0] - top2[:, 1]).min()), "keys", list(sd))


# --------------------------------------------------------------------------------------------------
# learn family: reference Collector.collect + PPOPolicy.update (process_fn + learn) on a tiny problem
# --------------------------------------------------------------------------------------------------
def gen_learn(name="learn", dual_clip=None, recompute=0, rounds=2):
    """name="learn": the reference's configuration; "learn_opts": dual-clip PPO + recompute_advantage, the two PPOPolicy options the
    CIRS scripts leave off (core/policy/ppo.py:73-99,176-177,190-193), one round."""
    import gym
    from core.collector import Collector
    from core.policy.ppo import PPOPolicy
    from tianshou.data import VectorReplayBuffer
    from tianshou.env import DummyVectorEnv

    U, I, B, T = 48, 96, 12, 10
    tab = make_tables(U, I, seed=0, with_ab=True, build_dist=True)
    envp = dict(num_leave_compute=3, leave_threshold=1, max_turn=T, tau=10.0, gamma_exposure=10.0, version="v1",
                r_decay=1.0, with_ab=True)
    register_envs(tab, **envp)
    st = make_reference_tracker(U, I, T, seed=21)
    st.eval()  # parity fixtures are recorded with dropout disabled
    net, actor, critic = make_reference_policy(I, seed=4)
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for p_ in list(actor.parameters()) + list(critic.parameters()):
            if p_.dim() == 1:
                p_.copy_(0.2 * torch.randn(p_.shape, generator=g))
        actor.last.model[0].weight.mul_(3.0)
    import warnings
    warnings.simplefilter("ignore")
    optim_RL = torch.optim.Adam(list(actor.parameters()) + list(critic.parameters()), lr=1e-3)
    optim_state = torch.optim.Adam(st.parameters(), lr=1e-3)
    sim_env = gym.make("SimulatedEnv-v0")
    policy = PPOPolicy(actor, critic, [optim_RL, optim_state], torch.distributions.Categorical, discount_factor=0.95,
                       max_grad_norm=0.5, eps_clip=0.2, vf_coef=0.25, ent_coef=0.0, reward_normalization=1,
                       advantage_normalization=1, recompute_advantage=recompute, value_clip=1, gae_lambda=0.95, dual_clip=dual_clip,
                       action_space=sim_env.action_space, action_bound_method="", action_scaling=False)
    random.seed(123); np.random.seed(123); torch.manual_seed(123)
    train_envs = DummyVectorEnv([lambda: gym.make("SimulatedEnv-v0") for _ in range(B)])
    collector = Collector(policy, train_envs, VectorReplayBuffer(B * T, B), preprocess_fn=st.build_state)
    policy.train()
    perms_all = []
    orig_perm = np.random.permutation

    def rec_perm(n):
        r = orig_perm(n); perms_all.append(np.array(r)); return r

    stash = {}
    orig_learn = policy.learn

    def learn_wrap(batch, **kw):
        if stash.get("locked"):
            return orig_learn(batch, **kw)
        stash.update(returns=batch.returns.detach().numpy().copy(), adv=batch.adv.detach().numpy().copy(),
                     v_s=batch.v_s.detach().numpy().copy(), logp_old=batch.logp_old.detach().numpy().copy(),
                     act=batch.act.detach().numpy().copy())
        return orig_learn(batch, **kw)
    policy.learn = learn_wrap

    def one_round(user_seed, perm_seed):
        """One Collector.collect(n_episode=B) + policy.update(...) of the reference -> recorded inputs / outputs."""
        random.seed(user_seed)
        res = collector.collect(n_episode=B)
        buf = collector.buffer
        users = np.array([int(w.env.cur_user[0]) for w in train_envs.workers])
        lens = np.array([len(b_) for b_ in buf.buffers])
        acts = np.full((B, T), -1, np.int64); rews = np.zeros((B, T)); dones = np.zeros((B, T), bool)
        obs = np.zeros((B, T + 1, 20), np.float32)
        for b in range(B):
            sl = slice(buf._offset[b], buf._offset[b] + lens[b])
            acts[b, :lens[b]] = buf.act[sl]; rews[b, :lens[b]] = buf.rew[sl]; dones[b, :lens[b]] = buf.done[sl]
            obs[b, :lens[b]] = buf.obs[sl].detach().numpy()
            obs[b, lens[b]] = buf.obs_next[sl][-1].detach().numpy()
        pre = {"pol_" + k: v.detach().clone().numpy() for k, v in policy.state_dict().items()}
        pre.update({"trk_" + k: v.detach().clone().numpy() for k, v in st.state_dict().items()})
        perms_all.clear()
        np.random.permutation = rec_perm
        np.random.seed(perm_seed)
        losses = policy.update(0, buf, batch_size=16, repeat=2)
        np.random.permutation = orig_perm
        perms = [p_.copy() for p_ in perms_all]
        post = {"post_pol_" + k: v.detach().numpy().copy() for k, v in policy.state_dict().items()}
        post.update({"post_trk_" + k: v.detach().numpy().copy() for k, v in st.state_dict().items()})
        # Collector.collect's result dict (core/collector.py:343-362): arrays in episode-completion order
        res_out = dict(res_rews=np.asarray(res["rews"], np.float64), res_lens=np.asarray(res["lens"], np.int64),
                       res_idxs=np.asarray(res["idxs"], np.int64),
                       res_scalars=np.array([res["n/ep"], res["n/st"], res["rew"], res["len"], res["rew_std"], res["len_std"]], np.float64))
        out = dict(users=users, lens=lens, acts=acts, rews=rews, dones=dones, obs=obs,
                   n_perm=np.int64(len(perms)), ret_rms=np.array([policy.ret_rms.mean, policy.ret_rms.var, policy.ret_rms.count], dtype=np.float64),
                   loss=np.array(losses["loss"]), loss_clip=np.array(losses["loss/clip"]), loss_vf=np.array(losses["loss/vf"]),
                   loss_ent=np.array(losses["loss/ent"]),
                   **{f"perm{i}": p_ for i, p_ in enumerate(perms)}, **{"b_" + k: v.copy() for k, v in stash.items()}, **pre, **post, **res_out)
        return out, lens, losses, perms

    out, lens, losses, perms = one_round(321, 77)
    out.update(hyper=np.array([0.95, 0.95, 0.2, 0.25, 0.0, 0.5, 1e-3, 16, 2]), dims=np.array([U, I, B, T]),
               opts=np.array([dual_clip or 0.0, float(recompute)]))
    if rounds < 2:
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **out)
        print(name + ".npz: N =", int(lens.sum()), "minibatches", len(losses["loss"]), "losses", np.round(losses["loss"], 4),
              "clip", np.round(losses["loss/clip"], 4))
        return
    # SECOND consecutive collect + update on the same policy / optimisers / ret_rms: the tracker's second Adam step pins
    # gradient MAGNITUDES (the first one is +-lr whatever the magnitude) and ret_rms / Adam-moment carry-over
    out2, lens2, losses2, perms2 = one_round(654, 78)
    out.update({"r2_" + k: v for k, v in out2.items() if not (k.startswith("pol_") or k.startswith("trk_"))})
    print("learn.npz round 2: N =", int(lens2.sum()), "lens", lens2.tolist(), "minibatches", len(losses2["loss"]))
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **out)
    print(name + ".npz: N =", int(lens.sum()), "lens", lens.tolist(), "minibatches", len(losses["loss"]), "losses", np.round(losses["loss"], 4),
          "perms", [len(p_) for p_ in perms])


# --------------------------------------------------------------------------------------------------
# collectorset family: the reference's test_episode over CollectorSet(FB, NX_0, NX_k) with the sampler noise
# supplied by the harness, so that the device rollout can be fed the very same noise and must return the same result dict.
# --------------------------------------------------------------------------------------------------
def gen_collectorset():
    import gym
    import core.policy.ppo as ref_ppo
    from core.collector_set import CollectorSet
    from core.policy.ppo import PPOPolicy
    from tianshou.env import DummyVectorEnv
    from tianshou.trainer.utils import test_episode
    import warnings
    warnings.simplefilter("ignore")

    U, I, B, T, K = 40, 120, 10, 12, 5
    tab = make_tables(U, I, seed=3, with_ab=True, build_dist=True)
    envp = dict(num_leave_compute=3, leave_threshold=1, max_turn=T, tau=10.0, gamma_exposure=10.0, version="v1", r_decay=1.0, with_ab=True)
    register_envs(tab, **envp)
    st = make_reference_tracker(U, I, T, seed=31)
    st.eval()
    net, actor, critic = make_reference_policy(I, seed=6)
    g = torch.Generator().manual_seed(23)
    with torch.no_grad():
        for p_ in list(actor.parameters()) + list(critic.parameters()):
            if p_.dim() == 1:
                p_.copy_(0.2 * torch.randn(p_.shape, generator=g))
        actor.last.model[0].weight.mul_(3.0)
    optim_RL = torch.optim.Adam(list(actor.parameters()) + list(critic.parameters()), lr=1e-3)
    optim_state = torch.optim.Adam(st.parameters(), lr=1e-3)
    env0 = gym.make("KuaishouEnv-v0")
    policy = PPOPolicy(actor, critic, [optim_RL, optim_state], torch.distributions.Categorical, discount_factor=0.95, max_grad_norm=0.5,
                       eps_clip=0.2, vf_coef=0.25, ent_coef=0.0, reward_normalization=1, advantage_normalization=1, recompute_advantage=0,
                       value_clip=1, gae_lambda=0.95, action_space=env0.action_space, action_bound_method="", action_scaling=False)
    names = ["FB", "NX_0", f"NX_{K}"]
    envs = {n: DummyVectorEnv([lambda: gym.make("KuaishouEnv-v0") for _ in range(B)]) for n in names}     # CIRS-RL-kuaishou.py:213-221
    cs = CollectorSet(policy, envs, B * T, B, preprocess_fn=st.build_state, force_length=K)                  # :297-299

    # ---- teacher-forced noise: q[name][t, env, item] ~ Exp(1); Categorical.sample := argmax(probs / q) (= torch.multinomial's race)
    gq = torch.Generator().manual_seed(99)
    noise = {n: torch.empty(T, B, I).exponential_(1.0, generator=gq) for n in names}
    ctx = dict(name=None, t=0, env_ids=None, idx=None)
    coll_of_policy_call = {id(c): n for n, c in cs.collector_dict.items()}

    orig_forward = policy.forward

    def forward_wrap(batch, buffer=None, remove_recommended_ids=False, state=None, **kw):
        n_rows = len(batch.obs)
        if len(buffer) == 0:
            env_ids = np.arange(n_rows)
        else:   # rows of self.data = live envs in ascending order (core/policy/utils.py:11 uses the same selection)
            env_ids = np.where(~buffer.done[buffer.last_index])[0]
        assert len(env_ids) == n_rows
        ctx["env_ids"], ctx["idx"] = env_ids, None
        out = orig_forward(batch, buffer=buffer, remove_recommended_ids=remove_recommended_ids, state=state, **kw)
        ctx["t"] += 1
        return out
    policy.forward = forward_wrap

    orig_remove = ref_ppo.removed_recommended_id_from_embedding

    def remove_wrap(logits, recommended_ids):
        lm, im = orig_remove(logits, recommended_ids)
        ctx["idx"] = im
        return lm, im
    ref_ppo.removed_recommended_id_from_embedding = remove_wrap

    def sample_patch(self, sample_shape=torch.Size()):
        q = noise[ctx["name"]][ctx["t"]][torch.as_tensor(ctx["env_ids"])]
        if ctx["idx"] is not None:
            q = q.gather(1, ctx["idx"])
        return torch.argmax(self.probs / q, dim=-1)
    orig_sample = torch.distributions.Categorical.sample
    torch.distributions.Categorical.sample = sample_patch

    for n, c in cs.collector_dict.items():     # tell the patch which collector is running
        oc = c.collect

        def wrapped(*a, _oc=oc, _n=n, **k):
            ctx["name"], ctx["t"] = _n, 0
            return _oc(*a, **k)
        c.collect = wrapped
    random.seed(2468)
    res = test_episode(policy, cs, None, 1, B, None, None)
    torch.distributions.Categorical.sample = orig_sample
    ref_ppo.removed_recommended_id_from_embedding = orig_remove

    out = dict(dims=np.array([U, I, B, T, K]), seed_tables=np.int64(3),
               env_params=np.array([envp["num_leave_compute"], envp["leave_threshold"], T]))
    for n, c in cs.collector_dict.items():
        buf = c.buffer
        lens = np.array([len(b_) for b_ in buf.buffers])
        acts = np.full((B, T), -1, np.int64); rews = np.zeros((B, T)); dones = np.zeros((B, T), bool)
        for b in range(B):
            sl = slice(buf._offset[b], buf._offset[b] + lens[b])
            acts[b, :lens[b]] = buf.act[sl]; rews[b, :lens[b]] = buf.rew[sl]; dones[b, :lens[b]] = buf.done[sl]
        out[f"{n}_users"] = np.array([int(np.asarray(w.env.cur_user).reshape(-1)[0]) for w in envs[n].workers])
        out[f"{n}_acts"], out[f"{n}_rews"], out[f"{n}_dones"], out[f"{n}_buf_lens"] = acts, rews, dones, lens
        out[f"{n}_gumbel"] = (-torch.log(noise[n])).numpy().astype(np.float32)
    for k, v in res.items():
        out["res_" + k.replace("/", "__")] = np.asarray(v)
    out.update({"pol_" + k: v.detach().numpy() for k, v in policy.state_dict().items()})
    out.update({"trk_" + k: v.detach().numpy() for k, v in st.state_dict().items()})
    np.savez_compressed(os.path.join(GOLDEN, "collectorset.npz"), **out)
    print("collectorset.npz: result keys", sorted(res), "FB lens", res["lens"].tolist(), "NX_0 lens", res["NX_0_lens"].tolist(),
          f"NX_{K} lens", res[f"NX_{K}_lens"].tolist())


# --------------------------------------------------------------------------------------------------
# virtualtb family (BASELINE configs[0], CPU plumbing): the reference's VirtualTB env with its shipped simulator weights, and
# SimulatedEnv(VirtualTB) around a seeded UserModel_MMOE, driven through torch's seeded CPU generator.
# (environments/VirtualTaobao/virtualTB/envs/virtualTB.py:74-133, core/env/simulatedEnv/simulated_env.py:49-168,
#  core/user_model_mmoe.py:144-262, CIRS-UserModel-taobao.py:100-148)
# --------------------------------------------------------------------------------------------------
def _pad_states(states, width=91):
    """states of VirtualTB are 91-d after a reset (user one-hot) and 30-d after a step (action): pad + keep the lengths"""
    out = np.full((len(states), width), np.nan)
    for k, s_ in enumerate(states):
        out[k, :len(s_)] = s_
    return out, np.array([len(s_) for s_ in states])


def gen_virtualtb():
    import collections
    import gym
    from core.user_model_mmoe import UserModel_MMOE
    from deepctr_torch.inputs import DenseFeat
    N, thr, T = 4, 2.4, 9
    gym.register(id="VirtualTB-v0", entry_point="virtualTB.envs.virtualTB:VirtualTB",
                 kwargs=dict(num_leave_compute=N, leave_threshold=thr, max_turn=T))
    rng = np.random.RandomState(5)
    n_steps = 40
    actions = rng.uniform(-1, 1, (n_steps, 27)).astype(np.float32)
    for k in range(3, n_steps, 5):            # near-repeats so that the exit rule fires (distance <= threshold)
        actions[k] = actions[k - 1] + rng.uniform(-0.2, 0.2, 27).astype(np.float32)
    out = dict(actions=actions, env_params=np.array([N, thr, T]))

    torch.manual_seed(11)
    env = gym.make("VirtualTB-v0")
    s = env.reset()
    states, rews, dones, ctrs = [np.asarray(s, np.float64)], [], [], []
    for a in actions:
        s, r, d, info = env.step(a)
        states.append(np.asarray(s, np.float64)); rews.append(r); dones.append(d); ctrs.append(info["CTR"])
        if d:
            s = env.reset()
            states.append(np.asarray(s, np.float64))
    es, el = _pad_states(states)
    out.update(env_states=es, env_state_len=el, env_rews=np.array(rews, np.float64), env_dones=np.array(dones), env_ctr=np.array(ctrs))

    x_columns = [DenseFeat("user_feat", 91), DenseFeat("feat_item", 27)]
    y_columns = [DenseFeat("y", 1)]
    tasks = collections.OrderedDict({f.name: "regression" for f in y_columns})
    task_logit_dim = {f.name: f.dimension for f in y_columns}
    model = UserModel_MMOE(x_columns, y_columns, len(tasks), tasks, task_logit_dim, dnn_hidden_units=(128, 128), seed=2022, device="cpu")
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():                    # the stock init (std 1e-4) makes the output almost constant: stress it
        for name, p_ in model.named_parameters():
            if name.startswith("dnn.") and name.endswith("weight"):
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.15)
            elif name.endswith("weight") and "linear_model" in name:
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.3)
            elif name.endswith("bias"):
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.1)
        model.tower_network[0].weight.mul_(0.05)
    model.eval()
    X = torch.randn(64, 118, generator=g)
    with torch.no_grad():
        out["mmoe_x"], out["mmoe_y"] = X.numpy(), model.forward(X).numpy()
    out.update({"mmoe_" + k: v.detach().numpy() for k, v in model.state_dict().items()})
    for ver, tau, gam in (("v1", 10.0, 3.0), ("v2", 1.0, 0.5)):
        gym.register(id="SimulatedEnv-v0", entry_point="core.env.simulatedEnv.simulated_env:SimulatedEnv",
                     kwargs=dict(user_model=model, task_name="VirtualTB-v0", version=ver, tau=tau, gamma_exposure=gam))
        torch.manual_seed(23)
        sim = gym.make("SimulatedEnv-v0")
        s = sim.reset()
        states, rews, dones, ctrs = [np.asarray(s, np.float64)], [], [], []
        for a in actions[:24]:
            s, r, d, info = sim.step(a)
            states.append(np.asarray(s, np.float64)); rews.append(float(r)); dones.append(d); ctrs.append(float(info["CTR"]))
            if d:
                s = sim.reset()
                states.append(np.asarray(s, np.float64))
        es, el = _pad_states(states)
        out.update({f"sim_{ver}_states": es, f"sim_{ver}_state_len": el, f"sim_{ver}_rews": np.array(rews), f"sim_{ver}_dones": np.array(dones),
                    f"sim_{ver}_ctr": np.array(ctrs), f"sim_{ver}_cfg": np.array([tau, gam])})
    np.savez_compressed(os.path.join(GOLDEN, "virtualtb.npz"), **out)
    print("virtualtb.npz: env dones", int(np.sum(out["env_dones"])), "rewards", out["env_rews"][:12], "sim v1 rews", np.round(out["sim_v1_rews"][:6], 4),
          "sim dones", int(out["sim_v1_dones"].sum()), int(out["sim_v2_dones"].sum()), "mmoe y", out["mmoe_y"][:4, 0])


# --------------------------------------------------------------------------------------------------
# deepfm family: UserModel_Pairwise.forward with the SHIPPED trained weights + compute_normed_reward
# --------------------------------------------------------------------------------------------------
def gen_deepfm():
    import pickle
    from core.user_model_pairwise import UserModel_Pairwise
    from environments.KuaishouRec.env.kuaishouEnv import KuaishouEnv
    base = os.path.join(ref_harness.REF_ROOT, "reproduce_results_of_our_paper", "results_alpha_beta")
    with open(os.path.join(base, "DeepFM_params_Pair11.pickle"), "rb") as f:
        params = pickle.load(f)
    params["device"] = "cpu"
    model = UserModel_Pairwise(**params)
    model.load_state_dict(torch.load(os.path.join(base, "DeepFM_Pair11.pt"), map_location="cpu"))
    model.eval()
    sd = model.state_dict()
    rng = np.random.RandomState(42)
    n_u, n_i = 48, 200
    raw_u = np.sort(rng.choice(7176, n_u, replace=False))
    raw_i = np.sort(rng.choice(10729, n_i, replace=False))
    n_cat = rng.randint(1, 5, size=n_i)
    feats = np.zeros((n_i, 4), np.int64)  # 0 = padding (kuaishouEnv.py:94-96: categories shifted by +1)
    for i in range(n_i):
        feats[i, :n_cat[i]] = rng.choice(31, n_cat[i], replace=False) + 1
    dur = rng.uniform(2, 60, n_i)
    # pair scoring: random (user, item) pairs
    n = 512
    pu = rng.randint(0, n_u, n); pi = rng.randint(0, n_i, n)
    X = np.concatenate([raw_u[pu, None], raw_i[pi, None], feats[pi], dur[pi, None]], axis=1)
    with torch.no_grad():
        y = model.forward(torch.tensor(X, dtype=torch.float)).squeeze(1).numpy()
    # full sweep + normalisation through the reference's own routine
    lbe_user = LabelEncoder().fit(raw_u); lbe_photo = LabelEncoder().fit(raw_i)
    df_photo_env = pd.DataFrame(feats, index=raw_i, columns=["feat0", "feat1", "feat2", "feat3"])
    df_photo_env.index.name = "photo_id"
    df_photo_env["photo_duration"] = dur
    normed = KuaishouEnv.compute_normed_reward(model, lbe_user, lbe_photo, df_photo_env)
    with torch.no_grad():
        pred = np.stack([model.forward(torch.tensor(np.concatenate([np.ones((n_i, 1)) * u, raw_i[:, None], feats, dur[:, None]], axis=1),
                                                    dtype=torch.float)).squeeze(1).numpy() for u in raw_u])
    out = dict(pu=pu, pi=pi, y=y, feats=feats, dur=dur.astype(np.float32), dur64=dur, normed=normed, pred=pred,
               raw_u=raw_u, raw_i=raw_i,
               emb_user=sd["embedding_dict.user_id.weight"].numpy()[raw_u], emb_item=sd["embedding_dict.photo_id.weight"].numpy()[raw_i],
               emb_feat=sd["embedding_dict.feat.weight"].numpy(),
               lin_user=sd["linear.embedding_dict.user_id.weight"].numpy()[raw_u, 0], lin_item=sd["linear.embedding_dict.photo_id.weight"].numpy()[raw_i, 0],
               lin_feat=sd["linear.embedding_dict.feat.weight"].numpy()[:, 0], lin_dense=sd["linear.weight"].numpy().reshape(-1),
               w1=sd["dnn.linears.0.weight"].numpy(), b1=sd["dnn.linears.0.bias"].numpy(),
               w2=sd["dnn.linears.1.weight"].numpy(), b2=sd["dnn.linears.1.bias"].numpy(),
               last=sd["last.weight"].numpy().reshape(-1), out_bias=sd["out.bias"].numpy().reshape(-1),
               alpha_u=sd["ab_embedding_dict.alpha_u.weight"].numpy()[raw_u, 0], beta_i=sd["ab_embedding_dict.beta_i.weight"].numpy()[raw_i, 0])
    np.savez_compressed(os.path.join(GOLDEN, "deepfm.npz"), **out)
    # the shipped checkpoint pair itself is DATA (trained weights + constructor kwargs): kept as a fixture so the
    # checkpoint interchange of SURVEY 8(f3) is tested on the real files
    import shutil
    for fn in ("DeepFM_Pair11.pt", "DeepFM_params_Pair11.pickle"):
        shutil.copy(os.path.join(ref_harness.REF_ROOT, "reproduce_results_of_our_paper", "results_alpha_beta", fn), os.path.join(GOLDEN, fn))
    print("deepfm.npz: y range", float(y.min()), float(y.max()), "pred", pred.shape, "normed range", float(normed.min()), float(normed.max()),
          "feat row0 norm", float(np.abs(out["emb_feat"][0]).max()), "keys", [k for k in sd.keys()][:30])


def gen_evalmetrics():
    """Callback_Coverage_Count.on_epoch_end + get_feat_dominate_dict + get_sorted_domination_features + LoggerCallback_Policy
    (reference evaluation.py:10-77,286-371, environments/KuaishouRec/env/data_handler.py:98-122, util/utils.py:83-137) on
    replay buffers filled exactly like Collector.collect does (one fresh VectorReplayBuffer per collector)."""
    from types import SimpleNamespace
    from evaluation import Callback_Coverage_Count, get_feat_dominate_dict
    from environments.KuaishouRec.env.data_handler import get_sorted_domination_features
    from tianshou.data import Batch, VectorReplayBuffer
    rng = np.random.RandomState(11)
    tab = make_tables(40, 150, seed=3, build_dist=False)
    I = tab.n_items
    raw_pid = tab.raw_pid
    n_raw = int(raw_pid.max()) + 1
    # df_item (indexed by RAW photo id, feat ids shifted by one, 0 = none; data_handler.py:29-33)
    feats_raw = np.zeros((n_raw, 4), np.int64)
    for rp in range(n_raw):
        f = tab.list_feat[rp]
        feats_raw[rp, :len(f)] = np.asarray(f) + 1
    df_item = pd.DataFrame(feats_raw, columns=["feat0", "feat1", "feat2", "feat3"])
    df_item.index.name = "photo_id"
    # training log -> domination list (multi-hot branch)
    n_log = 4000
    log_pid = rng.randint(0, n_raw, n_log)
    df_data = pd.DataFrame({"photo_id": log_pid, "watch_ratio": rng.gamma(2.0, 0.5, n_log)})
    df_data = df_data.join(df_item, on=["photo_id"], how="left")
    thr = np.percentile(df_data["watch_ratio"], 80)
    dom = get_sorted_domination_features(df_data, df_item, is_multi_hot=True, yname="watch_ratio", threshold=thr)
    lbe_photo = LabelEncoder().fit(raw_pid)
    out = dict(raw_pid=raw_pid, feats_raw=feats_raw, log_pid=log_pid, log_ratio=df_data["watch_ratio"].to_numpy(), log_thr=thr,
               dom_values=np.array([p[0] for p in dom["feat"]], np.int64), dom_shares=np.array([p[1] for p in dom["feat"]], np.float64),
               n_items=I)
    cases = []
    B, T = 12, 9
    for ci, (top_rate, peaked) in enumerate([(0.8, False), (0.6, True), (0.05, False), (0.95, True)]):
        names = ["FB", "NX_0", "NX_4"]
        coll = {}
        results = {"n/ep": B}
        rec = {}
        for name in names:
            buf = VectorReplayBuffer(B * (T + 2), B)
            lens = rng.randint(1, T + 1, B)
            if name == "NX_4":
                lens[:] = 4
            p = np.ones(I) / I
            if peaked:
                p = rng.dirichlet(np.full(I, 0.05))
            acts = np.full((B, T), -1, np.int64)
            idxs = np.zeros(B, np.int64)
            ready = np.arange(B)
            for t in range(T):
                live = ready[lens[ready] > t]
                if len(live) == 0:
                    break
                a = rng.choice(I, size=len(live), p=p)
                acts[live, t] = a
                done = lens[live] == t + 1
                batch = Batch(obs=np.zeros((len(live), 1)), act=a, rew=rng.uniform(size=len(live)), done=done,
                              obs_next=np.zeros((len(live), 1)), info=Batch(), policy=Batch())
                ptr, ep_rew, ep_len, ep_idx = buf.add(batch, buffer_ids=live)
                idxs[live[done]] = ep_idx[done]
            coll[name] = SimpleNamespace(buffer=buf)
            results[("" if name == "FB" else name + "_") + "idxs"] = idxs
            rec[name] = (acts, lens)
        tcs = SimpleNamespace(collector_dict=coll, env=SimpleNamespace(mat=[np.zeros((tab.n_users, I))]))
        cb = Callback_Coverage_Count(tcs, df_item, True, dom, lbe_photo, top_rate)
        res = cb.on_epoch_end(0, results=dict(results))
        for name in names:
            pre = "" if name == "FB" else name + "_"
            acts, lens = rec[name]
            out[f"c{ci}_{name}_acts"] = acts; out[f"c{ci}_{name}_lens"] = lens
            out[f"c{ci}_{name}_out"] = np.array([res[pre + "CV"], res[pre + "CV_turn"], res[pre + "ifeat_feat"]], np.float64)
        out[f"c{ci}_top_rate"] = top_rate
        cases.append(ci)
    out["n_cases"] = len(cases)
    np.savez_compressed(os.path.join(GOLDEN, "evalmetrics.npz"), **out)
    print("evalmetrics.npz:", {k: out[k] for k in out if k.endswith("_out")}, "dom", out["dom_values"][:6], out["dom_shares"][:6])


def write_kuairec_files(root, log_user, log_photo, log_ratio, list_feat, durations):
    """Tiny files in the KuaiRec layout the reference reads (kuaishouEnv.py:61-111)."""
    import json
    os.makedirs(root, exist_ok=True)
    pd.DataFrame({"user_id": log_user, "photo_id": log_photo, "play_duration": 1, "watch_ratio": log_ratio}).to_csv(
        os.path.join(root, "small_matrix.csv"), index=False)
    with open(os.path.join(root, "item_categories.json"), "w") as fh:
        json.dump({str(i): {"feature_index": [int(c) for c in f]} for i, f in enumerate(list_feat)}, fh)
    with open(os.path.join(root, "photo_mean_duration.json"), "w") as fh:
        json.dump({str(i): float(d) for i, d in enumerate(durations)}, fh)


def gen_loaders():
    """KuaishouEnv.load_mat + get_distance_mat (reference kuaishouEnv.py:61-111, core/util.py:225-273) on tiny files."""
    import tempfile
    import environments.KuaishouRec.env.kuaishouEnv as ke
    import core.util as cu
    rng = np.random.RandomState(5)
    n_raw_item, n_raw_user = 60, 40
    list_feat = [sorted(rng.choice(31, size=rng.randint(1, 5), replace=False).tolist()) for _ in range(n_raw_item)]
    durations = rng.uniform(2, 60, n_raw_item)
    users = rng.choice(n_raw_user, 25, replace=False); photos = rng.choice(n_raw_item, 33, replace=False)
    uu, pp = np.meshgrid(users, photos, indexing="ij")
    order = rng.permutation(uu.size)
    log_user, log_photo = uu.ravel()[order], pp.ravel()[order]
    log_ratio = rng.gamma(1.5, 1.2, uu.size)     # some > 5 (clipped)
    with tempfile.TemporaryDirectory() as root:
        write_kuairec_files(root, log_user, log_photo, log_ratio, list_feat, durations)
        old = ke.DATAPATH
        ke.DATAPATH = root
        try:
            mat, lbe_user, lbe_photo, lf, df_photo_env, df_dist_small = ke.KuaishouEnv.load_mat()
        finally:
            ke.DATAPATH = old
        cached = pd.read_csv(os.path.join(root, "distance_mat_photo_small.csv"), index_col=0)
    out = dict(log_user=log_user, log_photo=log_photo, log_ratio=log_ratio, durations=durations,
               list_feat=np.array([f + [-1] * (4 - len(f)) for f in list_feat], np.int64),
               mat=mat, user_classes=lbe_user.classes_, photo_classes=lbe_photo.classes_,
               photo_env_index=df_photo_env.index.to_numpy(), photo_env_values=df_photo_env.to_numpy(dtype=np.float64),
               dist=df_dist_small.to_numpy(dtype=np.float64), dist_index=df_dist_small.index.to_numpy(),
               dist_columns=df_dist_small.columns.to_numpy().astype(np.int64), dist_csv=cached.to_numpy(dtype=np.float64))
    np.savez_compressed(os.path.join(GOLDEN, "loaders.npz"), **out)
    print("loaders.npz: mat", mat.shape, "max", mat.max(), "dist inf share", float(np.isinf(out["dist"]).mean()), "cols", list(df_photo_env.columns))


def load_shipped_user_model():
    import pickle
    from core.user_model_pairwise import UserModel_Pairwise
    base = os.path.join(ref_harness.REF_ROOT, "reproduce_results_of_our_paper", "results_alpha_beta")
    with open(os.path.join(base, "DeepFM_params_Pair11.pickle"), "rb") as fh:
        params = pickle.load(fh)
    params["device"] = "cpu"
    model = UserModel_Pairwise(**params)
    model.load_state_dict(torch.load(os.path.join(base, "DeepFM_Pair11.pt"), map_location="cpu"))
    model.eval()
    return model


def gen_staticpolicy():
    """UserModel.recommend_k_item (reference core/user_model.py:254-348) and interactive_evaluation (evaluation.py:79-151)
    with the shipped DeepFM weights on a small synthetic KuaishouEnv.  Sampling modes use the shared-noise protocol:
    torch.multinomial(probs, 1) is replaced by argmax(log probs + g) with recorded Gumbel noise g."""
    import random as pyrandom
    from types import SimpleNamespace
    import evaluation as ev
    from environments.KuaishouRec.env.kuaishouEnv import KuaishouEnv
    model = load_shipped_user_model()
    tab = make_tables(24, 96, seed=9, raw_user_space=7176, raw_item_space=10729)
    kw = build_reference_env_kwargs(tab, num_leave_compute=3, leave_threshold=1, max_turn=12)
    dataset_val = SimpleNamespace(df_photo_env=kw["df_photo_env"], x_columns=list(range(7)))
    I = tab.n_items
    rng = np.random.RandomState(3)
    out = dict(raw_uid=tab.raw_uid, raw_pid=tab.raw_pid, mat=tab.mat, item_cats=tab.item_cats, duration=tab.duration, dist=tab.dist)
    # ---- recommend_k_item -------------------------------------------------------------------------------------------
    rec = []
    noise = {}
    real_multinomial = torch.multinomial

    def fake_multinomial(probs, k, replacement=False):
        g = noise["g"][:probs.numel()]
        return torch.argmax(torch.log(probs) + torch.as_tensor(g, dtype=probs.dtype)).reshape(1)

    cases = [dict(softmax=False, removed=[]), dict(softmax=False, removed=[3, 17, 40, 41, 95]), dict(softmax=True, removed=[]),
             dict(softmax=True, removed=[0, 1, 2, 64]), dict(softmax=False, removed=[], ucb=True), dict(softmax=False, removed=[], ucb=True)]
    users = rng.choice(tab.raw_uid, len(cases))
    torch.multinomial = fake_multinomial
    try:
        for ci, (c, u) in enumerate(zip(cases, users)):
            g = rng.gumbel(size=I).astype(np.float32)        # noise per ITEM id; the reference samples over the preserved items
            keep = np.ones(I, bool); keep[c["removed"]] = False
            noise["g"] = g[keep]
            if c.get("ucb") and ci == len(cases) - 1:   # second UCB call sees the counts of the first + some extra pulls
                model.n_each[rng.randint(0, I, 200)] += 3
                model.n_rec += 600
            ucb_state = (getattr(model, "n_rec", I), getattr(model, "n_each", np.ones(I)).copy()) if c.get("ucb") else (0, np.zeros(I))
            t_id, raw_id, val = model.recommend_k_item(int(u), dataset_val, k=1, is_softmax=c["softmax"], epsilon=0, is_ucb=bool(c.get("ucb")),
                                                      recommended_ids=list(c["removed"]))
            out[f"r{ci}_user"] = u; out[f"r{ci}_softmax"] = int(c["softmax"]); out[f"r{ci}_removed"] = np.array(c["removed"], np.int64)
            out[f"r{ci}_gumbel"] = g; out[f"r{ci}_ucb"] = int(bool(c.get("ucb"))); out[f"r{ci}_n_rec"] = ucb_state[0]; out[f"r{ci}_n_each"] = ucb_state[1]
            out[f"r{ci}_out"] = np.array([int(np.asarray(t_id).reshape(-1)[0]), int(np.asarray(raw_id).reshape(-1)[0])], np.int64)
            out[f"r{ci}_val"] = np.float32(np.asarray(val).reshape(-1)[0])
            rec.append(ci)
    finally:
        torch.multinomial = real_multinomial
    out["n_rec_cases"] = len(rec)
    # ---- interactive_evaluation (greedy: deterministic given the users) ----------------------------------------------
    dom = {"feat": [(1, 0.4), (2, 0.3), (5, 0.2), (7, 0.1)]}
    n_traj = 10
    for ei, (remove, fl, ucb) in enumerate([(False, 0, False), (True, 0, False), (True, 5, False), (False, 0, True), (True, 5, True)]):
        env = KuaishouEnv(**kw)
        if ucb:   # fresh arm counts (core/user_model.py:250-252, 303-306)
            for attr in ("n_rec", "n_each"):
                if hasattr(model, attr):
                    delattr(model, attr)
        pyrandom.seed(100 + ei)
        drawn = []
        orig_reset = env.reset

        def rec_reset(_orig=orig_reset, _drawn=drawn):
            o = _orig()
            _drawn.append(int(np.asarray(o).reshape(-1)[0]))
            return o
        env.reset = rec_reset
        res = ev.interactive_evaluation(model, env, dataset_val, is_softmax=False, epsilon=0, is_ucb=ucb, k=1, need_transform=True,
                                        num_trajectory=n_traj, item_feat_domination=dom, remove_recommended=remove, force_length=fl,
                                        top_rate=0.6)
        pre = f"NX_{fl}_" if remove else ""
        out[f"e{ei}_users"] = np.array(drawn, np.int64)
        out[f"e{ei}_cfg"] = np.array([int(remove), fl, int(ucb)], np.int64)
        out[f"e{ei}_res"] = np.array([float(res[pre + "click_loss"]), float(res[pre + "CV"]), float(res[pre + "CV_turn"]), float(res[pre + "ctr"]),
                                      float(res[pre + "len_tra"]), float(res[pre + "R_tra"]), float(res[pre + "ifeat_feat"])], np.float64)
    out["n_eval_cases"] = 5
    out["dom_values"] = np.array([p[0] for p in dom["feat"]], np.int64); out["dom_shares"] = np.array([p[1] for p in dom["feat"]])
    np.savez_compressed(os.path.join(GOLDEN, "staticpolicy.npz"), **out)
    print("staticpolicy.npz:", {k: out[k] for k in out if k.endswith("_out") or k.endswith("_res")})


def gen_usertrain():
    """UserModel_Pairwise training (reference core/user_model.py:87-170 fit_data, core/user_model_pairwise.py:134-151 get_loss,
    CIRS-UserModel-kuaishou.py:262-278 loss_kuaishou_pairwise): three optimiser steps on fixed batches, run BOTH through the
    reference's fit_data (shuffle off) and through its inner-loop statements one by one (per-step losses)."""
    import copy
    import importlib.util
    from types import SimpleNamespace
    from core.user_model_pairwise import UserModel_Pairwise
    from core.inputs import SparseFeatP
    from core.static_dataset import StaticDataset
    from deepctr_torch.inputs import DenseFeat
    spec = importlib.util.spec_from_file_location("cirs_usermodel_script", os.path.join(ref_harness.REF_ROOT, "CIRS-UserModel-kuaishou.py"))
    script = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(script)               # defines loss_kuaishou_pairwise (reads the module-level `args.lambda_ab`)
    out = {}
    for ci, (U, I, E, n, use_ab, lam) in enumerate([(50, 80, 8, 32, True, 10.0), (40, 60, 16, 48, True, 1.0), (30, 50, 8, 16, False, 0.0)]):
        script.args = SimpleNamespace(lambda_ab=lam)
        F = 32
        x_columns = [SparseFeatP("user_id", U, embedding_dim=E), SparseFeatP("photo_id", I, embedding_dim=E)] + \
                    [SparseFeatP(f"feat{i}", F, embedding_dim=E, embedding_name="feat", padding_idx=0) for i in range(4)] + [DenseFeat("photo_duration", 1)]
        ab_columns = [SparseFeatP("alpha_u", U, embedding_dim=1), SparseFeatP("beta_i", I, embedding_dim=1)] if use_ab else None
        y_columns = [DenseFeat("y", 1)]
        torch.manual_seed(7 + ci)
        model = UserModel_Pairwise(x_columns, y_columns, "regression", 1, dnn_hidden_units=(64, 64), seed=2022, l2_reg_dnn=0.1, device="cpu",
                                   ab_columns=ab_columns)
        rng = np.random.RandomState(ci)
        with torch.no_grad():     # the reference initialises embeddings with std 1e-4: scale up so every term of the loss matters
            for name, prm in model.named_parameters():
                if "embedding_dict" in name and "ab_" not in name:
                    prm.copy_(torch.as_tensor(rng.normal(0, 0.3, prm.shape).astype(np.float32)))
                    if name == "embedding_dict.feat.weight":
                        prm[0] = 0
                if name.startswith("ab_embedding_dict"):
                    prm.copy_(torch.as_tensor(rng.normal(1, 0.2, prm.shape).astype(np.float32)))
        init = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
        steps = 3

        def col(v):
            return np.asarray(v, np.float64)[:, None]
        N = steps * n
        feats = lambda: np.where(np.arange(4)[None, :] < rng.randint(1, 5, N)[:, None], rng.randint(1, F, (N, 4)), 0)
        u = rng.randint(0, U, N)
        x = np.concatenate([col(u), col(rng.randint(0, I, N)), feats(), col(rng.uniform(2, 60, N)),
                            col(u), col(rng.randint(0, I, N)), feats(), col(rng.uniform(2, 60, N))], axis=1)
        y = rng.uniform(0, 5, (N, 1)); score = rng.gamma(1.0, 0.5, (N, 1))
        model.compile(optimizer="adam", loss_func=script.loss_kuaishou_pairwise, metric_fun={}, metrics=None)
        model_b = copy.deepcopy(model)
        model_b.compile(optimizer="adam", loss_func=script.loss_kuaishou_pairwise, metric_fun={}, metrics=None)
        # (a) the reference's own loop
        ds = StaticDataset(x_columns, y_columns, num_workers=0)
        ds.compile_dataset(pd.DataFrame(x), pd.DataFrame(y), score)
        model.RL_eval_fun = None
        model.fit_data(ds, dataset_val=None, batch_size=n, epochs=1, shuffle=False, callbacks=[])
        final_a = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
        # (b) the same statements step by step, for the per-step numbers
        losses = []
        for st in range(steps):
            xb = torch.as_tensor(x[st * n:(st + 1) * n]).float(); yb = torch.as_tensor(y[st * n:(st + 1) * n]).float()
            sb = torch.as_tensor(score[st * n:(st + 1) * n]).float()
            loss = model_b.get_loss(xb, yb, sb).squeeze()
            model_b.optim.zero_grad()
            reg = model_b.get_regularization_loss()
            total = loss + reg + model_b.aux_loss
            total.backward()
            model_b.optim.step()
            losses.append([float(loss), float(reg)])
            if st == 0:
                first = {k: v.detach().clone().numpy() for k, v in model_b.state_dict().items()}
        final_b = {k: v.detach().clone().numpy() for k, v in model_b.state_dict().items()}
        for k in final_a:
            assert np.array_equal(final_a[k], final_b[k]), k          # the step-by-step replay IS fit_data
        pre = f"c{ci}_"
        out[pre + "cfg"] = np.array([U, I, F, E, n, int(use_ab), steps], np.int64); out[pre + "lambda_ab"] = lam
        out[pre + "x"] = x; out[pre + "y"] = y; out[pre + "score"] = score; out[pre + "losses"] = np.array(losses)
        for k, v in init.items():
            out[pre + "init_" + k] = v
        for k, v in first.items():
            out[pre + "first_" + k] = v
        for k, v in final_a.items():
            out[pre + "final_" + k] = v
    out["n_cases"] = 3
    np.savez_compressed(os.path.join(GOLDEN, "usertrain.npz"), **out)
    print("usertrain.npz:", {k: out[k] for k in out if k.endswith("losses")})


def gen_dataprep():
    """compute_exposure_effect_kuaishouRec / compute_exposure_each_user and find_negative (reference core/util.py:56-76,
    135-196; numba's njit is the identity in this harness) on a small synthetic interaction log."""
    import tempfile
    import core.util as cu
    rng = np.random.RandomState(17)
    n_users, n_items = 12, 70
    list_feat = [sorted(rng.choice(31, size=rng.randint(1, 5), replace=False).tolist()) for _ in range(n_items)]
    rows = []
    t0 = 1.6e9
    for u in range(n_users):
        L = rng.randint(1, 40)
        ts = np.sort(t0 + rng.randint(0, 5000, L).astype(np.float64))   # repeated timestamps occur (t_diff == 0 -> 1)
        for k in range(L):
            rows.append((u, int(rng.randint(0, n_items)), ts[k]))
    df = pd.DataFrame(rows, columns=["user_id", "photo_id", "timestamp"])
    out = dict(user_id=df["user_id"].to_numpy(), photo_id=df["photo_id"].to_numpy(), timestamp=df["timestamp"].to_numpy(),
               list_feat=np.array([f + [-1] * (4 - len(f)) for f in list_feat], np.int64))
    for tau in (1000.0, 50.0):
        with tempfile.TemporaryDirectory() as root:
            os.makedirs(os.path.join(root, "m", "x"))
            ex = cu.compute_exposure_effect_kuaishouRec(df[["user_id", "photo_id"]], df["timestamp"], list_feat, tau,
                                                        os.path.join(root, "m", "x"), root)
        out[f"exposure_tau{int(tau)}"] = np.asarray(ex, np.float64).reshape(-1)
    # negative sampling
    n_items2 = 1300   # includes the absent id 1225
    mat_small = rng.uniform(size=(n_users, n_items2)) < 0.3
    mat_big = rng.uniform(size=(n_users, n_items2)) < 0.3
    mat_small[3, 600:] = True; mat_big[3, :10] = True      # forces the downward search
    mat_small[4, 1220:1230] = True                          # neighbourhood of the absent id
    uids = rng.randint(0, n_users, 400); pids = rng.randint(0, n_items2, 400)
    uids[:6] = [3, 3, 4, 4, 4, 5]; pids[:6] = [700, 1299, 1223, 1224, 1226, 1299]
    df_negative = np.zeros((len(uids), 2))
    cu.find_negative(uids, pids, mat_small, mat_big, df_negative, n_items2 - 1)
    out.update(neg_users=uids, neg_items=pids, mat_small=np.packbits(mat_small, axis=1, bitorder="little"),
               mat_big=np.packbits(mat_big, axis=1, bitorder="little"), n_items2=n_items2, negatives=df_negative)
    np.savez_compressed(os.path.join(GOLDEN, "dataprep.npz"), **out)
    print("dataprep.npz: rows", len(df), "exposure max", out["exposure_tau1000"].max(), out["exposure_tau50"].max(), "neg sample", df_negative[:8, 1])


def gen_userdata():
    """load_dataset_kuaishou (reference CIRS-UserModel-kuaishou.py:86-148 with core/util.py negative_sampling and
    compute_exposure_effect_kuaishouRec) on tiny files in the KuaiRec layout: the assembled training arrays (x incl. the negative
    half, y, exposure score) and the column descriptors, for tau = 0 and tau > 0."""
    import tempfile
    spec = importlib.util.spec_from_file_location("cirs_usermodel_script2", os.path.join(ref_harness.REF_ROOT, "CIRS-UserModel-kuaishou.py"))
    script = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(script)
    rng = np.random.RandomState(23)
    n_users, n_items = 15, 1300   # ids reach past 1225, the absent photo id of the negative search
    list_feat = [sorted(rng.choice(31, size=rng.randint(1, 5), replace=False).tolist()) for _ in range(n_items)]
    durations = rng.uniform(2, 60, n_items)
    rows = []
    t0 = 1.6e9
    hot = np.r_[np.arange(1215, 1235), rng.choice(n_items, 80, replace=False)]
    for u in range(n_users):
        L = rng.randint(3, 45)
        ts = np.sort(t0 + rng.randint(0, 6000, L).astype(np.float64))
        items = rng.choice(hot, L)   # a user's log, rows contiguous and time-ordered (the layout of big_matrix.csv)
        for k in range(L):
            rows.append((u, int(items[k]), ts[k], float(rng.gamma(1.5, 1.3)), float(durations[items[k]] * 1000.0)))
    big = pd.DataFrame(rows, columns=["user_id", "photo_id", "timestamp", "watch_ratio", "photo_duration"])
    small_u = rng.randint(0, n_users, 300); small_p = rng.choice(hot, 300)
    out = dict(big_user=big["user_id"].to_numpy(), big_photo=big["photo_id"].to_numpy(), big_ts=big["timestamp"].to_numpy(),
               big_ratio=big["watch_ratio"].to_numpy(), big_dur=big["photo_duration"].to_numpy(), small_user=small_u, small_photo=small_p,
               durations=durations, list_feat=np.array([f + [-1] * (4 - len(f)) for f in list_feat], np.int64))
    for tau in (0.0, 800.0):
        with tempfile.TemporaryDirectory() as root:
            write_kuairec_files(root, small_u, small_p, np.ones(len(small_u)), list_feat, durations)
            big.to_csv(os.path.join(root, "big_matrix.csv"), index=False)
            save = os.path.join(root, "saved_models", "env", "model")
            os.makedirs(save)
            script.DATAPATH = root
            dataset, x_columns, y_columns, ab_columns = script.load_dataset_kuaishou(tau, 8, 8, save)
        tag = f"tau{int(tau)}"
        out[f"x_{tag}"] = np.asarray(dataset.x_numpy, np.float64)
        out[f"y_{tag}"] = np.asarray(dataset.y_numpy, np.float64)
        out[f"score_{tag}"] = np.asarray(dataset.score, np.float64)
        if tau == 0.0:
            out["x_col_names"] = np.array([c.name for c in x_columns])
            out["x_col_vocab"] = np.array([getattr(c, "vocabulary_size", 0) for c in x_columns], np.int64)
            out["x_col_dim"] = np.array([getattr(c, "embedding_dim", getattr(c, "dimension", 0)) for c in x_columns], np.int64)
            out["ab_col_vocab"] = np.array([c.vocabulary_size for c in ab_columns], np.int64)
    np.savez_compressed(os.path.join(GOLDEN, "userdata.npz"), **out)
    print("userdata.npz: rows", len(big), "x", out["x_tau0"].shape, "score max", out["score_tau800"].max(), "names", list(out["x_col_names"]))


def gen_userval():
    """load_static_validate_data_kuaishou + StaticDataset.set_env_items (reference core/util.py:81-133, core/static_dataset.py:19-26)
    on tiny files: the validation arrays and the per-item table of the evaluation environment."""
    import tempfile
    import core.util as cu
    rng = np.random.RandomState(29)
    n_users, n_items = 20, 90
    list_feat = [sorted(rng.choice(31, size=rng.randint(1, 5), replace=False).tolist()) for _ in range(n_items)]
    durations = rng.uniform(2, 60, n_items)
    users = rng.choice(n_users, 12, replace=False); photos = rng.choice(n_items, 25, replace=False)
    uu, pp = np.meshgrid(users, photos, indexing="ij")
    order = rng.permutation(uu.size)
    log_user, log_photo = uu.ravel()[order], pp.ravel()[order]
    log_ratio = rng.gamma(1.5, 1.4, uu.size)
    log_dur = durations[log_photo] * 1000.0 * rng.uniform(0.9, 1.1, uu.size)
    with tempfile.TemporaryDirectory() as root:
        write_kuairec_files(root, log_user, log_photo, log_ratio, list_feat, durations)
        df = pd.read_csv(os.path.join(root, "small_matrix.csv"))
        df["photo_duration"] = log_dur
        df.to_csv(os.path.join(root, "small_matrix.csv"), index=False)
        ds = cu.load_static_validate_data_kuaishou(8, 8, root)
    out = dict(log_user=log_user, log_photo=log_photo, log_ratio=log_ratio, log_dur=log_dur, durations=durations,
               list_feat=np.array([f + [-1] * (4 - len(f)) for f in list_feat], np.int64),
               x=np.asarray(ds.x_numpy, np.float64), y=np.asarray(ds.y_numpy, np.float64),
               env_index=ds.df_photo_env.index.to_numpy(), env_columns=np.array(list(ds.df_photo_env.columns)),
               env_values=ds.df_photo_env.to_numpy(dtype=np.float64),
               x_col_vocab=np.array([getattr(c, "vocabulary_size", 0) for c in ds.x_columns], np.int64))
    np.savez_compressed(os.path.join(GOLDEN, "userval.npz"), **out)
    print("userval.npz: x", out["x"].shape, "env table", out["env_values"].shape, list(out["env_columns"]))


# --------------------------------------------------------------------------------------------------
# c1rl family: the RL loop of CIRS-RL-taobao.py (BASELINE configs[0]: VirtualTaobao, 4 envs, CPU) -- two collect + update rounds
# --------------------------------------------------------------------------------------------------
def _stressed_mmoe():
    import collections
    from core.user_model_mmoe import UserModel_MMOE
    from deepctr_torch.inputs import DenseFeat
    x_columns = [DenseFeat("user_feat", 91), DenseFeat("feat_item", 27)]
    y_columns = [DenseFeat("y", 1)]
    tasks = collections.OrderedDict({f.name: "regression" for f in y_columns})
    task_logit_dim = {f.name: f.dimension for f in y_columns}
    model = UserModel_MMOE(x_columns, y_columns, len(tasks), tasks, task_logit_dim, dnn_hidden_units=(128, 128), seed=2022, device="cpu")
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for name, p_ in model.named_parameters():
            if name.startswith("dnn.") and name.endswith("weight"):
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.15)
            elif name.endswith("weight") and "linear_model" in name:
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.3)
            elif name.endswith("bias"):
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.1)
        model.tower_network[0].weight.mul_(0.05)
    return model.eval()


def gen_c1rl():
    """CIRS-RL-taobao.py:150-300 with 4 envs: SimulatedEnv(VirtualTB) x 4 -> dense-feature StateTrackerTransformer (dropout 0.1 live,
    as in the script) -> ActorProb / Critic, Independent(Normal) PPO with action scaling -> Collector.collect(n_episode=4) +
    policy.update, twice.  Recorded: initial parameters, per-round buffer rows (states, raw actions, rewards, dones), result
    dicts, loss lists, parameters after every update, the return statistics."""
    import gym
    from torch.distributions import Independent, Normal
    from core.collector import Collector
    from core.inputs import get_dataset_columns
    from core.policy.ppo import PPOPolicy
    from core.state_tracker import StateTrackerTransformer
    from core.user_model import compute_input_dim
    from tianshou.data import VectorReplayBuffer
    from tianshou.env import DummyVectorEnv
    from tianshou.utils.net.common import Net
    from tianshou.utils.net.continuous import ActorProb, Critic
    N, thr, T, B = 4, 2.4, 9, 4
    gym.register(id="VirtualTB-v0", entry_point="virtualTB.envs.virtualTB:VirtualTB", kwargs=dict(num_leave_compute=N, leave_threshold=thr, max_turn=T))
    model = _stressed_mmoe()
    gym.register(id="SimulatedEnv-v0", entry_point="core.env.simulatedEnv.simulated_env:SimulatedEnv",
                 kwargs=dict(user_model=model, task_name="VirtualTB-v0", version="v1", tau=10.0, gamma_exposure=3.0))
    sim = gym.make("SimulatedEnv-v0")
    train_envs = DummyVectorEnv([lambda: gym.make("SimulatedEnv-v0") for _ in range(B)])
    seed = 2022
    np.random.seed(seed); torch.manual_seed(seed); train_envs.seed(seed)
    dim_model, dim_state = 27, 20
    uc, ac, fc, hu, ha, hf = get_dataset_columns(dim_model, envname="VirtualTB-v0")
    assert dim_model == compute_input_dim(ac)
    tracker = StateTrackerTransformer(uc, ac, fc, dim_model=dim_model, dim_state=dim_state, dim_max_batch=B, dataset="VirtualTB-v0",
                                      has_user_embedding=hu, has_action_embedding=ha, has_feedback_embedding=hf, nhead=3, d_hid=128,
                                      nlayers=2, dropout=0.1, device="cpu", seed=seed, MAX_TURN=T)
    net = Net(dim_state, hidden_sizes=[64, 64], device="cpu")
    actor = ActorProb(net, sim.action_space.shape, max_action=sim.action_space.high[0], device="cpu")
    critic = Critic(net, device="cpu")
    for m in list(actor.modules()) + list(critic.modules()):
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight); torch.nn.init.zeros_(m.bias)
    optim = [torch.optim.Adam(list(actor.parameters()) + list(critic.parameters()), lr=1e-3), torch.optim.Adam(tracker.parameters(), lr=1e-3)]

    def dist(*logits):
        return Independent(Normal(*logits), 1)

    policy = PPOPolicy(actor, critic, optim, dist, discount_factor=0.95, max_grad_norm=0.5, eps_clip=0.2, vf_coef=0.25, ent_coef=0.0,
                       reward_normalization=1, advantage_normalization=1, recompute_advantage=0, value_clip=1, gae_lambda=0.95,
                       action_space=sim.action_space)
    collector = Collector(policy, train_envs, VectorReplayBuffer(400, B), preprocess_fn=tracker.build_state)
    out = dict(cfg=np.array([N, thr, T, B, dim_model, dim_state], np.float64), action_low=sim.action_space.low, action_high=sim.action_space.high)
    out.update({"mmoe_" + k: v.detach().numpy() for k, v in model.state_dict().items()})
    snap = lambda tag: out.update({f"{tag}_actor_{k}": v.detach().numpy().copy() for k, v in actor.state_dict().items()} |
                                  {f"{tag}_critic_{k}": v.detach().numpy().copy() for k, v in critic.state_dict().items()} |
                                  {f"{tag}_tracker_{k}": v.detach().numpy().copy() for k, v in tracker.state_dict().items()})
    snap("init")
    for rnd in range(2):
        torch.manual_seed(100 + rnd); np.random.seed(200 + rnd)
        res = collector.collect(n_episode=B)
        buf = collector.buffer
        idx = buf.sample_index(0)
        b = buf[idx]
        out.update({f"r{rnd}_idx": idx, f"r{rnd}_obs": b.obs.detach().numpy(), f"r{rnd}_obs_next": b.obs_next.detach().numpy(),
                    f"r{rnd}_act": np.asarray(b.act), f"r{rnd}_rew": np.asarray(b.rew, np.float64), f"r{rnd}_done": np.asarray(b.done),
                    f"r{rnd}_res_rews": res["rews"], f"r{rnd}_res_lens": res["lens"], f"r{rnd}_res_idxs": res["idxs"],
                    f"r{rnd}_res_n": np.array([res["n/ep"], res["n/st"]])})
        losses = policy.update(0, buf, batch_size=16, repeat=2)
        out.update({f"r{rnd}_loss_" + k.replace("/", "_"): np.array(v) for k, v in losses.items()})
        out[f"r{rnd}_ret_rms"] = np.array([policy.ret_rms.mean, policy.ret_rms.var, policy.ret_rms.count], np.float64)
        snap(f"r{rnd}")
        # where the update left torch's generator (process_fn / learn go through forward(), which samples: ppo.py:107,183); the next
        # round reseeds, so this probe draw changes nothing downstream
        out[f"r{rnd}_rng_probe"] = torch.rand(4).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "c1rl.npz"), **out)
    print("c1rl.npz: n/st", out["r0_res_n"], out["r1_res_n"], "losses", np.round(out["r0_loss_loss"][:4], 5), "lens", out["r0_res_lens"], out["r1_res_lens"],
          "act[0,:3]", out["r0_act"][0, :3])


def gen_bufferindex():
    """Index semantics of the reference's VectorReplayBuffer (tianshou/data/buffer/manager.py:91-142, base.py prev / next / unfinished_index) as DATA:
    op scripts are replayed on the reference's own class and the full index state is dumped after every op -> tests/golden/buffer_index.json.
    Script 0 is the scenario of the reference's known-answer test (tianshou/test/base/test_buffer.py:397-488: 4 sub-buffers of 5, episodes ending in
    different sub-buffers, ring wrap-around inside sub-buffer 2); scripts 1-2 are seeded random add sequences on other geometries."""
    import json
    from tianshou.data import Batch, VectorReplayBuffer

    def scenario():
        z = [0, 0, 0, 0]
        return dict(total=20, n=4, ops=[
            dict(val=[1, 2, 3], done=[0, 0, 1], ids=[0, 1, 2]),
            dict(val=[4], done=[1], ids=[3]),
            dict(val=z, done=[0, 0, 0, 0], ids=[0, 1, 2, 3]),
            dict(val=z, done=[1, 1, 1, 1], ids=[0, 1, 2, 3]),
            dict(val=z, done=[0, 0, 0, 0], ids=[0, 1, 2, 3]),
            dict(val=z, done=[0, 1, 0, 1], ids=[0, 1, 2, 3]),
            dict(val=[1], done=[1], ids=[2]),
        ])

    def random_script(seed, total, n, n_ops):
        rng = np.random.RandomState(seed)
        ops = []
        for _ in range(n_ops):
            k = rng.randint(1, n + 1)
            ids = sorted(rng.choice(n, size=k, replace=False).tolist())
            ops.append(dict(val=rng.randint(0, 9, k).tolist(), done=(rng.uniform(size=k) < 0.3).astype(int).tolist(), ids=ids))
        return dict(total=total, n=n, ops=ops)

    scripts = [scenario(), random_script(1, 12, 3, 40), random_script(2, 35, 5, 60)]
    for sc in scripts:
        buf = VectorReplayBuffer(sc["total"], sc["n"])
        for op in sc["ops"]:
            v = np.array(op["val"])
            ptr, ep_rew, ep_len, ep_idx = buf.add(Batch(obs=v, act=v, rew=v, done=np.array(op["done"])), buffer_ids=op["ids"])
            idx = np.sort(buf.sample_index(0))
            op["want"] = dict(ptr=np.asarray(ptr).tolist(), ep_rew=np.asarray(ep_rew, np.float64).tolist(), ep_len=np.asarray(ep_len).tolist(),
                              ep_idx=np.asarray(ep_idx).tolist(), len=len(buf), index=idx.tolist(), prev=buf.prev(idx).tolist(),
                              next=buf.next(idx).tolist(), unfinished=np.asarray(buf.unfinished_index()).tolist(),
                              done=np.asarray(buf.done).astype(int).tolist(), rew=np.asarray(buf.rew, np.float64).tolist(),
                              prev_last=int(buf.prev(-1)), next_last=int(buf.next(-1)), sample_minus1=buf.sample_index(-1).tolist())
    with open(os.path.join(GOLDEN, "buffer_index.json"), "w") as f:
        json.dump(dict(source="reference VectorReplayBuffer driven by oracle/gen_golden.py:gen_bufferindex", scripts=scripts), f, separators=(",", ":"))
    print("buffer_index.json:", [len(sc["ops"]) for sc in scripts], "ops; final unfinished", [sc["ops"][-1]["want"]["unfinished"] for sc in scripts])


FAMILIES = {"bufferindex": gen_bufferindex, "c1rl": gen_c1rl, "virtualtb": gen_virtualtb, "collectorset": gen_collectorset, "userval": gen_userval, "userdata": gen_userdata, "dataprep": gen_dataprep, "usertrain": gen_usertrain, "staticpolicy": gen_staticpolicy, "loaders": gen_loaders, "evalmetrics": gen_evalmetrics, "deepfm": gen_deepfm, "learn": gen_learn, "learn_opts": lambda: gen_learn("learn_opts", dual_clip=1.01, recompute=1, rounds=1), "env": gen_env, "tracker": gen_tracker, "policy": gen_policy}

if __name__ == "__main__":
    names = sys.argv[1:] or list(FAMILIES)
    for n in names:
        FAMILIES[n]()
