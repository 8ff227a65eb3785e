"""Times the REFERENCE implementation itself (its own Python, imported from /root/reference through oracle/ref_harness.py) on this
container's CPU cores for the metrics bench.py reports: rollout env-steps/s (Collector.collect), PPO minibatch steps/s
(policy.update) and DeepFM (user, item) pairs/s -> profiles/reference_python_cpu.json, which bench.py quotes inside `cpu_baseline`
(the reference cannot travel to the GPU box; SURVEY 8(d) "CPU baseline plan").  Dev container only:

    python tools/bench_reference.py [c2] [c3] [deepfm]

Synthetic KuaiRec-shaped tables (cirs_hip/synthetic.py), tau = 10, gamma_exposure = 10, max_turn = 30, recent-N = 10,
leave_threshold = 4, batch 1024 x repeat 2, dropout left ON as the reference runs it."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden as G  # noqa: E402  (installs the harness stubs and puts the reference on sys.path)

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(ROOT, "profiles", "reference_python_cpu.json")


def rl_case(U, I, B, T=30, N=10, thr=4):
    import gym
    from core.collector import Collector
    from core.policy.ppo import PPOPolicy
    from tianshou.data import VectorReplayBuffer
    from tianshou.env import DummyVectorEnv
    import warnings
    warnings.simplefilter("ignore")
    tab = G.make_tables(U, I, seed=0, with_ab=True, build_dist=True)
    G.register_envs(tab, num_leave_compute=N, leave_threshold=thr, max_turn=T, tau=10.0, gamma_exposure=10.0, version="v1", r_decay=1.0, with_ab=True)
    st = G.make_reference_tracker(U, I, T, seed=21, randomize=False)
    net, actor, critic = G.make_reference_policy(I, seed=4)
    optim_RL = torch.optim.Adam(list(actor.parameters()) + list(critic.parameters()), lr=1e-3)
    optim_state = torch.optim.Adam(st.parameters(), lr=1e-3)
    sim_env = gym.make("SimulatedEnv-v0")
    policy = PPOPolicy(actor, critic, [optim_RL, optim_state], torch.distributions.Categorical, discount_factor=0.95, max_grad_norm=0.5,
                       eps_clip=0.2, vf_coef=0.25, ent_coef=0.0, reward_normalization=1, advantage_normalization=1, recompute_advantage=0,
                       value_clip=1, gae_lambda=0.95, action_space=sim_env.action_space, action_bound_method="", action_scaling=False)
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    envs = DummyVectorEnv([lambda: gym.make("SimulatedEnv-v0") for _ in range(B)])
    coll = Collector(policy, envs, VectorReplayBuffer(B * T, B), preprocess_fn=st.build_state)
    policy.train()
    rec = {"users": U, "items": I, "envs": B, "max_turn": T, "recent_N": N, "collects": []}
    for it in range(2):
        t0 = time.perf_counter()
        res = coll.collect(n_episode=B)
        t1 = time.perf_counter()
        losses = policy.update(0, coll.buffer, batch_size=1024, repeat=2)
        t2 = time.perf_counter()
        rec["collects"].append({"env_steps": int(res["n/st"]), "collect_s": t1 - t0, "rollout_env_steps_per_s": res["n/st"] / (t1 - t0),
                                "update_s": t2 - t1, "minibatch_steps": len(losses["loss"]), "ppo_minibatch_steps_per_s": len(losses["loss"]) / (t2 - t1),
                                "env_steps_per_s_collect_plus_update": res["n/st"] / (t2 - t0)})
        print(rec["collects"][-1], flush=True)
    return rec


def deepfm_case():
    model = G.load_shipped_user_model()
    n = 10729
    rng = np.random.RandomState(0)
    X = np.concatenate([np.full((n, 1), 17.0), np.arange(n)[:, None], rng.randint(0, 32, (n, 4)), rng.uniform(2, 60, (n, 1))], axis=1)
    Xt = torch.tensor(X, dtype=torch.float)
    with torch.no_grad():
        model.forward(Xt)
        t0 = time.perf_counter()
        for _ in range(20):
            model.forward(Xt)
        dt = (time.perf_counter() - t0) / 20
    return {"pairs_per_call": n, "seconds_per_call": dt, "pairs_per_s": n / dt}


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "deepfm"]
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    out.update({"what": "the reference's own Python (chongminggao/CIRS-codes) timed in the dev container; NOT the GPU box's host",
                "cores": os.cpu_count(), "torch": torch.__version__, "threads": torch.get_num_threads()})
    if "c2" in which:
        out["c2"] = rl_case(1411, 3327, 64)
    if "c3" in which:
        out["c3"] = rl_case(7176, 10728, 1024)
    if "deepfm" in which:
        out["deepfm"] = deepfm_case()
    json.dump(out, open(OUT, "w"), indent=1)
    print(json.dumps(out, indent=1))
