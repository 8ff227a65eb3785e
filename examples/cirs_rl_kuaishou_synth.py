#!/usr/bin/env python
"""CIRS-RL-kuaishou.py's wiring (reference :141-334) on synthetic KuaiRec-shaped tables, through the mirrored plugin
surface: same module names, class names and constructor keywords; the hot loop runs in libcirs_hip.so.

    PYTHONPATH=cirs-codes_amd python examples/cirs_rl_kuaishou_synth.py --epoch 3
"""
import argparse
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cirs_hip import gymlite  # noqa: E402

gym = gymlite.install()
from gym.envs.registration import register  # noqa: E402

from core.collector import Collector  # noqa: E402
from core.collector_set import CollectorSet  # noqa: E402
from core.trainer.onpolicy import onpolicy_trainer  # noqa: E402
from core.inputs import get_dataset_columns  # noqa: E402
from core.policy.ppo import PPOPolicy  # noqa: E402
from core.state_tracker import StateTrackerTransformer  # noqa: E402
from core.user_model import compute_input_dim  # noqa: E402
from tianshou.data import VectorReplayBuffer  # noqa: E402
from tianshou.env import DummyVectorEnv  # noqa: E402
from tianshou.utils.net.common import Net  # noqa: E402
from tianshou.utils.net.discrete import Actor, Critic  # noqa: E402


def get_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--env", default="KuaishouEnv-v0")
    p.add_argument("--seed", default=2023, type=int)
    p.add_argument("--version", default="v1")
    p.add_argument("--tau", default=100, type=float)
    p.add_argument("--gamma_exposure", default=10, type=float)
    p.add_argument("--r_decay", default=1, type=float)
    p.add_argument("--leave_threshold", default=1, type=int)
    p.add_argument("--num_leave_compute", default=3, type=int)
    p.add_argument("--max_turn", default=30, type=int)
    p.add_argument("--dim_state", default=20, type=int)
    p.add_argument("--dim_model", default=32, type=int)
    p.add_argument("--nhead", default=4, type=int)
    p.add_argument("--buffer-size", type=int, default=11000)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--gamma", type=float, default=0.95)
    p.add_argument("--epoch", type=int, default=3)
    p.add_argument("--repeat-per-collect", type=int, default=2)
    p.add_argument("--batch-size", type=int, default=1024)
    p.add_argument("--hidden-sizes", type=int, nargs="*", default=[64, 64])
    p.add_argument("--episode-per-collect", type=int, default=100)
    p.add_argument("--training-num", type=int, default=100)
    p.add_argument("--vf-coef", type=float, default=0.25)
    p.add_argument("--ent-coef", type=float, default=0.0)
    p.add_argument("--eps-clip", type=float, default=0.2)
    p.add_argument("--max-grad-norm", type=float, default=0.5)
    p.add_argument("--gae-lambda", type=float, default=0.95)
    p.add_argument("--test-num", type=int, default=100)
    p.add_argument("--force_length", type=int, default=10)
    p.add_argument("--step-per-epoch", type=int, default=2000)
    p.add_argument("--top_rate", type=float, default=0.8)
    p.add_argument("--dropout", type=float, default=0.1, help="state-tracker dropout (live during rollout and update, like the reference)")
    p.add_argument("--n-users", type=int, default=1411)
    p.add_argument("--n-items", type=int, default=3327)
    return p.parse_args(argv)


def build(args, table_seed=0):
    from cirs_hip.synthetic import make_tables
    tab = make_tables(args.n_users, args.n_items, seed=table_seed)
    lbe_user = types.SimpleNamespace(classes_=tab.raw_uid)
    lbe_photo = types.SimpleNamespace(classes_=tab.raw_pid)
    device = torch.device("cuda:0")
    # %% 3. prepare envs (reference :170-221)
    register(id=args.env, entry_point="environments.KuaishouRec.env.kuaishouEnv:KuaishouEnv",
             kwargs={"mat": tab.mat, "lbe_user": lbe_user, "lbe_photo": lbe_photo, "num_leave_compute": args.num_leave_compute,
                     "leave_threshold": args.leave_threshold, "max_turn": args.max_turn, "list_feat": tab.list_feat,
                     "df_photo_env": None, "df_dist_small": tab.dist})
    env = gym.make(args.env)
    register(id="SimulatedEnv-v0", entry_point="core.env.simulatedEnv.simulated_env:SimulatedEnv",
             kwargs={"user_model": None, "task_name": args.env, "version": args.version, "tau": args.tau, "alpha_u": tab.alpha_u,
                     "beta_i": tab.beta_i, "normed_mat": tab.normed_mat, "gamma_exposure": args.gamma_exposure, "r_decay": args.r_decay})
    simulatedEnv = gym.make("SimulatedEnv-v0")
    train_envs = DummyVectorEnv([lambda: gym.make("SimulatedEnv-v0") for _ in range(args.training_num)])
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    train_envs.seed(args.seed)
    # %% 4. Setup model (reference :229-285)
    user_columns, action_columns, feedback_columns, has_user_embedding, has_action_embedding, has_feedback_embedding = \
        get_dataset_columns(args.dim_model, envname=args.env, env=env)
    assert args.dim_model == compute_input_dim(action_columns)
    state_tracker = StateTrackerTransformer(user_columns, action_columns, feedback_columns, dim_model=args.dim_model,
                                            dim_state=args.dim_state, dim_max_batch=args.training_num, dataset=args.env,
                                            has_user_embedding=has_user_embedding, has_action_embedding=has_action_embedding,
                                            has_feedback_embedding=has_feedback_embedding, nhead=args.nhead, d_hid=128, nlayers=2,
                                            dropout=args.dropout, device=device, seed=args.seed, MAX_TURN=args.max_turn).to(device)
    net = Net(args.dim_state, hidden_sizes=args.hidden_sizes, device=device)
    actor = Actor(net, env.mat.shape[1], device=device).to(device)
    critic = Critic(net, device=device).to(device)
    for m in list(actor.modules()) + list(critic.modules()):
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # duplicate trunk parameters, like the reference
        optim_RL = torch.optim.Adam(list(actor.parameters()) + list(critic.parameters()), lr=args.lr)
    optim_state = torch.optim.Adam(state_tracker.parameters(), lr=args.lr)
    policy = PPOPolicy(actor, critic, [optim_RL, optim_state], torch.distributions.Categorical, discount_factor=args.gamma,
                       max_grad_norm=args.max_grad_norm, eps_clip=args.eps_clip, vf_coef=args.vf_coef, ent_coef=args.ent_coef,
                       reward_normalization=1, advantage_normalization=1, recompute_advantage=0, value_clip=1,
                       gae_lambda=args.gae_lambda, action_space=simulatedEnv.action_space, action_bound_method="", action_scaling=False)
    # %% 5. collectors (reference :288-292)
    train_collector = Collector(policy, train_envs, VectorReplayBuffer(args.buffer_size, len(train_envs)),
                                preprocess_fn=state_tracker.build_state)
    return tab, train_envs, state_tracker, policy, train_collector


def build_test_collectors(args, policy, state_tracker):
    """The FB / NX_0 / NX_k test collectors on the real (non-simulated) env (reference :213-221, :297-299)."""
    test_envs_dict = {"FB": DummyVectorEnv([lambda: gym.make(args.env) for _ in range(args.test_num)]),
                      "NX_0": DummyVectorEnv([lambda: gym.make(args.env) for _ in range(args.test_num)]),
                      f"NX_{args.force_length}": DummyVectorEnv([lambda: gym.make(args.env) for _ in range(args.test_num)])}
    return CollectorSet(policy, test_envs_dict, args.buffer_size, args.test_num, preprocess_fn=state_tracker.build_state,
                        force_length=args.force_length)


def build_callbacks(args, tab, test_collector_set):
    """Coverage / feature-domination / log callbacks (reference :303-316); the 'training log' is synthetic."""
    import pandas as pd
    from environments.KuaishouRec.env.data_handler import get_sorted_domination_features
    from evaluation import Callback_Coverage_Count
    from util.utils import LoggerCallback_Policy
    n_raw = int(tab.raw_pid.max()) + 1
    feats = np.zeros((n_raw, 4), np.int64)  # category ids shifted by one, 0 = none (data_handler.py:29-33)
    for rp in range(n_raw):
        f = tab.list_feat[rp]
        feats[rp, :len(f)] = np.asarray(f) + 1
    df_item = pd.DataFrame(feats, columns=["feat0", "feat1", "feat2", "feat3"])
    rng = np.random.RandomState(args.seed)
    df_data = pd.DataFrame({"photo_id": rng.randint(0, n_raw, 20000), "watch_ratio": rng.gamma(2.0, 0.5, 20000)})
    df_data = df_data.join(df_item, on=["photo_id"], how="left")
    item_feat_domination = get_sorted_domination_features(df_data, df_item, is_multi_hot=True, yname="watch_ratio",
                                                          threshold=np.percentile(df_data["watch_ratio"], 80))
    lbe_photo = types.SimpleNamespace(classes_=tab.raw_pid)
    return [Callback_Coverage_Count(test_collector_set, df_item, need_transform=True, item_feat_domination=item_feat_domination,
                                    lbe_photo=lbe_photo, top_rate=args.top_rate),
            LoggerCallback_Policy("cirs_rl_kuaishou_synth.log", args.force_length)]


def main(argv=None):
    args = get_args(argv)
    tab, train_envs, state_tracker, policy, train_collector = build(args)
    test_collector_set = build_test_collectors(args, policy, state_tracker)
    policy.callbacks = build_callbacks(args, tab, test_collector_set)
    result = onpolicy_trainer(policy, train_collector, test_collector_set, state_tracker, args.epoch, args.step_per_epoch,
                              args.repeat_per_collect, args.test_num, args.batch_size, episode_per_collect=args.episode_per_collect,
                              save_model_fn=lambda epoch, policy: None)
    print(result)
    return policy, state_tracker, result


if __name__ == "__main__":
    main()
