#!/usr/bin/env python
"""The reference's RL entry point on the MI355X path, step for step.

`main(args)` walks the seven steps of reference CIRS-RL-kuaishou.py:119-345 in the same order, through the same imports, calls
and keywords (tests/golden/entrypoint_surface.json pins them; tests/test_entrypoint_surface.py checks that this file's import
block and command line are the reference's): dirs + log file, user model artefacts, envs via gym.register / gym.make, state
tracker + actor / critic + PPOPolicy, collectors + logger + callbacks, onpolicy_trainer, final checkpoint.  The reference reads
the KuaiRec files and a trained user model from the working directory and ships neither, so `prepare_workspace(root)` first writes
a synthetic data set in the KuaiRec layout and trains a small user model with the mirror's own pipeline
(core.user_model_train.train_user_model) into `<root>/saved_models/...`, exactly where step 2 looks.

    python examples/cirs_rl_kuaishou.py --workspace /tmp/cirs_ws --epoch 2 --step-per-epoch 600

Every vector step of steps 5-6 runs in libcirs_hip.so (collect = cirs_rollout_steps, policy.update = cirs_ppo_* + cirs_tracker_backward)."""
import datetime
import functools
import json
import os
import pickle
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))

# ---- the import block of the reference entry point (CIRS-RL-kuaishou.py:10-43) ---------------------------------------------------
import gym  # noqa: E402
import torch  # noqa: E402
import argparse  # noqa: E402
import numpy as np  # noqa: E402

from core.collector_set import CollectorSet  # noqa: E402
from core.inputs import get_dataset_columns  # noqa: E402

from torch.utils.tensorboard import SummaryWriter  # noqa: E402

from core.collector import Collector  # noqa: E402
from core.state_tracker import StateTrackerTransformer  # noqa: E402
from core.user_model import compute_input_dim  # noqa: E402
from core.policy.ppo import PPOPolicy  # noqa: E402
from core.user_model_pairwise import UserModel_Pairwise  # noqa: E402
from environments.KuaishouRec.env.data_handler import get_df_kuairec, get_training_item_domination, load_item_feat  # noqa: E402,F401
from environments.KuaishouRec.env.kuaishouEnv import KuaishouEnv  # noqa: E402
from tianshou.utils import BasicLogger  # noqa: E402
from tianshou.env import DummyVectorEnv  # noqa: E402
from tianshou.utils.net.common import Net  # noqa: E402
from core.trainer.onpolicy import onpolicy_trainer  # noqa: E402
from tianshou.data import VectorReplayBuffer  # noqa: E402
from tianshou.utils.net.discrete import Actor, Critic  # noqa: E402

import logzero  # noqa: E402
from logzero import logger  # noqa: E402

from evaluation import Callback_Coverage_Count  # noqa: E402
from util.utils import create_dir, LoggerCallback_RL, LoggerCallback_Policy  # noqa: E402,F401

from gym.envs.registration import register  # noqa: E402


def get_args(argv=None):
    """The reference's command line (CIRS-RL-kuaishou.py:45-115: same options, same defaults; `--cuda` defaults to 0 here, the
    reference's lab machine used GPU 1) plus --workspace."""
    p = argparse.ArgumentParser()
    p.add_argument("--env", type=str, default="KuaishouEnv-v0")
    p.add_argument("--user_model_name", type=str, default="DeepFM")
    p.add_argument("--model_name", type=str, default="CIRS")
    p.add_argument("--seed", default=2023, type=int)
    p.add_argument("--cuda", default=0, type=int)
    p.add_argument("--is_ab", dest="is_ab", action="store_true")
    p.add_argument("--no_ab", dest="is_ab", action="store_false")
    p.set_defaults(is_ab=True)
    p.add_argument("--cpu", dest="cpu", action="store_true")
    p.set_defaults(cpu=False)
    p.add_argument("--is_save", dest="is_save", action="store_true")
    p.add_argument("--no_save", dest="is_save", action="store_false")
    p.set_defaults(is_save=False)
    # env
    p.add_argument("--version", type=str, default="v1")
    p.add_argument("--tau", default=100, type=float)
    p.add_argument("--gamma_exposure", default=10, type=float)
    p.add_argument("--r_decay", default=1, type=float)
    p.add_argument("--leave_threshold", default=0, type=int)
    p.add_argument("--num_leave_compute", default=1, type=int)
    p.add_argument("--max_turn", default=30, type=int)
    # state tracker
    p.add_argument("--dim_state", default=20, type=int)
    p.add_argument("--dim_model", default=32, type=int)
    p.add_argument("--nhead", default=4, type=int)
    p.add_argument("--force_length", type=int, default=10)
    p.add_argument("--top_rate", type=float, default=0.8)
    # trainer
    p.add_argument("--buffer-size", type=int, default=11000)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--gamma", type=float, default=0.95)
    p.add_argument("--epoch", type=int, default=50)
    p.add_argument("--step-per-epoch", type=int, default=15000)
    p.add_argument("--repeat-per-collect", type=int, default=2)
    p.add_argument("--batch-size", type=int, default=1024)
    p.add_argument("--hidden-sizes", type=int, nargs="*", default=[64, 64])
    p.add_argument("--episode-per-collect", type=int, default=100)
    p.add_argument("--training-num", type=int, default=100)
    p.add_argument("--test-num", type=int, default=100)
    p.add_argument("--render", type=float, default=0)
    # ppo
    p.add_argument("--vf-coef", type=float, default=0.25)
    p.add_argument("--ent-coef", type=float, default=0.0)
    p.add_argument("--eps-clip", type=float, default=0.2)
    p.add_argument("--max-grad-norm", type=float, default=0.5)
    p.add_argument("--gae-lambda", type=float, default=0.95)
    p.add_argument("--rew-norm", type=int, default=1)
    p.add_argument("--dual-clip", type=float, default=None)
    p.add_argument("--value-clip", type=int, default=1)
    p.add_argument("--norm-adv", type=int, default=1)
    p.add_argument("--recompute-adv", type=int, default=0)
    p.add_argument("--resume", action="store_true")
    p.add_argument("--save-interval", type=int, default=1000)
    p.add_argument("--read_message", type=str, default="UserModel1")
    p.add_argument("--message", type=str, default="CIRS")
    # not in the reference: where the synthetic KuaiRec files and the user-model artefacts live (becomes the working directory)
    p.add_argument("--workspace", type=str, default=None)
    return p.parse_known_args(argv)[0]


def prepare_workspace(root, args, seed=0, um_epochs=3):
    """What the reference expects to find on disk: the KuaiRec files under environments/KuaishouRec/data and the three user-model
    artefacts of CIRS-UserModel-kuaishou.py under saved_models/<env>/<user_model_name>/.  Synthetic files + a short training run."""
    from cirs_hip.synthetic import write_kuairec_workspace
    from core.user_model_train import train_user_model
    datapath = os.path.join(root, "environments", "KuaishouRec", "data")
    os.makedirs(datapath, exist_ok=True)
    if not os.path.isfile(os.path.join(datapath, "small_matrix.csv")):
        write_kuairec_workspace(datapath, seed=seed)
    os.environ["CIRS_DATAPATH"] = datapath
    um_dir = os.path.join(root, "saved_models", args.env, args.user_model_name)
    if not os.path.isfile(os.path.join(um_dir, "{}_{}.pt".format(args.user_model_name, args.read_message))):
        train_user_model(datapath, save_root=root, env=args.env, user_model_name=args.user_model_name, message=args.read_message,
                         tau=float(args.tau) * 10, feature_dim=8, batch_size=256, epoch=um_epochs, lr=5e-3)
    return datapath


def main(args):
    # %% 1. directories and the log file (reference :120-132)
    save_dir = os.path.join(".", "saved_models", args.env, args.model_name)
    create_dir([os.path.join(".", "saved_models"), os.path.join(".", "saved_models", args.env), save_dir, os.path.join(save_dir, "logs")])
    stamp = datetime.datetime.fromtimestamp(time.time()).strftime("%Y_%m_%d-%H_%M_%S")
    logger_path = os.path.join(save_dir, "logs", "[{}]_{}.log".format(args.message, stamp))
    logzero.logfile(logger_path)
    logger.info(json.dumps(vars(args), indent=2))
    device = "cpu" if args.cpu else torch.device("cuda:{}".format(args.cuda) if torch.cuda.is_available() else "cpu")

    # %% 2. the trained user model: constructor arguments from the pickle, weights from the .pt (reference :141-165)
    um_dir = os.path.join(".", "saved_models", args.env, args.user_model_name)
    with open(os.path.join(um_dir, "{}_params_{}.pickle".format(args.user_model_name, args.read_message)), "rb") as fh:
        model_params = pickle.load(fh)
    model_params["device"] = "cpu"
    user_model = UserModel_Pairwise(**model_params)
    user_model.load_state_dict(torch.load(os.path.join(um_dir, "{}_{}.pt".format(args.user_model_name, args.read_message))))
    if hasattr(user_model, "ab_embedding_dict") and args.is_ab:
        alpha_u = user_model.ab_embedding_dict["alpha_u"].weight.detach().cpu().numpy()
        beta_i = user_model.ab_embedding_dict["beta_i"].weight.detach().cpu().numpy()
    else:
        print("Note there are no available alpha and beta!")
        alpha_u, beta_i = np.ones([7176, 1]), np.ones([10729, 1])

    # %% 3. environments (reference :170-226): the real env for testing, the simulated env (user model reward) for training
    mat, lbe_user, lbe_photo, list_feat, df_photo_env, df_dist_small = KuaishouEnv.load_mat()
    register(id=args.env, entry_point="environments.KuaishouRec.env.kuaishouEnv:KuaishouEnv",
             kwargs={"mat": mat, "lbe_user": lbe_user, "lbe_photo": lbe_photo, "num_leave_compute": args.num_leave_compute,
                     "leave_threshold": args.leave_threshold, "max_turn": args.max_turn, "list_feat": list_feat,
                     "df_photo_env": df_photo_env, "df_dist_small": df_dist_small})
    env = gym.make(args.env)
    with open(os.path.join(um_dir, "normed_mat-{}.pickle".format(args.read_message)), "rb") as fh:
        normed_mat = pickle.load(fh)
    register(id="SimulatedEnv-v0", entry_point="core.env.simulatedEnv.simulated_env:SimulatedEnv",
             kwargs={"user_model": user_model, "task_name": args.env, "version": args.version, "tau": args.tau, "alpha_u": alpha_u,
                     "beta_i": beta_i, "normed_mat": normed_mat, "gamma_exposure": args.gamma_exposure, "r_decay": args.r_decay})
    simulatedEnv = gym.make("SimulatedEnv-v0")
    state_shape = simulatedEnv.observation_space.shape or simulatedEnv.observation_space.n   # noqa: F841  (read like the reference)
    action_shape = simulatedEnv.action_space.shape or simulatedEnv.action_space.n             # noqa: F841
    max_action = simulatedEnv.action_space.high[0]                                            # noqa: F841
    train_envs = DummyVectorEnv([lambda: gym.make("SimulatedEnv-v0") for _ in range(args.training_num)])
    test_envs = DummyVectorEnv([lambda: gym.make(args.env) for _ in range(args.test_num)])
    test_envs_NX_0 = DummyVectorEnv([lambda: gym.make(args.env) for _ in range(args.test_num)])
    test_envs_NX_x = DummyVectorEnv([lambda: gym.make(args.env) for _ in range(args.test_num)])
    test_envs_dict = {"FB": test_envs, "NX_0": test_envs_NX_0, f"NX_{args.force_length}": test_envs_NX_x}
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    train_envs.seed(args.seed)

    # %% 4. models (reference :229-285)
    user_columns, action_columns, feedback_columns, has_user_embedding, has_action_embedding, has_feedback_embedding = \
        get_dataset_columns(args.dim_model, envname=args.env, env=env)
    assert args.dim_model == compute_input_dim(action_columns)
    state_tracker = StateTrackerTransformer(user_columns, action_columns, feedback_columns, dim_model=args.dim_model,
                                            dim_state=args.dim_state, dim_max_batch=max(args.training_num, args.test_num),
                                            dataset=args.env, has_user_embedding=has_user_embedding,
                                            has_action_embedding=has_action_embedding, has_feedback_embedding=has_feedback_embedding,
                                            nhead=args.nhead, d_hid=128, nlayers=2, dropout=0.1, device=device, seed=args.seed,
                                            MAX_TURN=args.max_turn).to(device)
    net = Net(args.dim_state, hidden_sizes=args.hidden_sizes, device=device)
    actor = Actor(net, env.mat.shape[1], device=device).to(device)
    critic = Critic(net, device=device).to(device)
    for m in list(actor.modules()) + list(critic.modules()):     # orthogonal initialisation
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    optim_RL = torch.optim.Adam(list(actor.parameters()) + list(critic.parameters()), lr=args.lr)   # the trunk twice
    optim_state = torch.optim.Adam(state_tracker.parameters(), lr=args.lr)
    optim = [optim_RL, optim_state]
    dist = torch.distributions.Categorical
    policy = PPOPolicy(actor, critic, optim, dist, discount_factor=args.gamma, max_grad_norm=args.max_grad_norm, eps_clip=args.eps_clip,
                       vf_coef=args.vf_coef, ent_coef=args.ent_coef, reward_normalization=args.rew_norm,
                       advantage_normalization=args.norm_adv, recompute_advantage=args.recompute_adv, value_clip=args.value_clip,
                       gae_lambda=args.gae_lambda, action_space=simulatedEnv.action_space,
                       action_bound_method="" if args.env == "KuaishouEnv-v0" else "clip",
                       action_scaling=False if args.env == "KuaishouEnv-v0" else True)

    # %% 5. collectors, logger, callbacks (reference :288-316)
    train_collector = Collector(policy, train_envs, VectorReplayBuffer(args.buffer_size, len(train_envs)),
                                preprocess_fn=state_tracker.build_state)
    test_collector_set = CollectorSet(policy, test_envs_dict, args.buffer_size, args.test_num, preprocess_fn=state_tracker.build_state,
                                      force_length=args.force_length)
    writer = SummaryWriter(os.path.join(save_dir))
    logger1 = BasicLogger(writer, save_interval=args.save_interval)
    df_item_val = load_item_feat(only_small=True)
    item_feat_domination = get_training_item_domination()
    policy.callbacks = [Callback_Coverage_Count(test_collector_set, df_item_val, need_transform=True,
                                                item_feat_domination=item_feat_domination, lbe_photo=env.lbe_photo, top_rate=args.top_rate),
                        LoggerCallback_Policy(logger_path, args.force_length)]

    # %% 6. training (reference :320-334)
    model_save_path = os.path.join(save_dir, "{}_{}.pt".format(args.model_name, args.message))
    result = onpolicy_trainer(policy, train_collector, test_collector_set, state_tracker, args.epoch, args.step_per_epoch,
                              args.repeat_per_collect, args.test_num, args.batch_size, episode_per_collect=args.episode_per_collect,
                              logger=logger1, resume_from_log=args.resume,
                              save_model_fn=functools.partial(save_model_fn, model_save_path=model_save_path, state_tracker=state_tracker,
                                                              optim=optim, is_save=args.is_save))

    # %% 7. the four-key checkpoint (reference :340-345)
    torch.save({"policy": policy.cpu().state_dict(), "optim_RL": optim[0].state_dict(), "optim_state": optim[1].state_dict(),
                "state_tracker": state_tracker.cpu().state_dict()}, model_save_path)
    return dict(result=result, policy=policy, state_tracker=state_tracker, model_save_path=model_save_path, logger_path=logger_path,
                log_dir=save_dir, callbacks=policy.callbacks)


def save_model_fn(epoch, policy, model_save_path, optim, state_tracker, is_save=False):
    """Per-epoch checkpoint `<name>-e<epoch>.pt` (reference :348-358), only with --is_save."""
    if not is_save:
        return
    path = model_save_path[:-3] + "-e{}".format(epoch) + model_save_path[-3:]
    torch.save({"policy": policy.state_dict(), "optim_RL": optim[0].state_dict(), "optim_state": optim[1].state_dict(),
                "state_tracker": state_tracker.state_dict()}, path)


def run(argv=None):
    import tempfile
    args = get_args(argv)
    ws = args.workspace or tempfile.mkdtemp(prefix="cirs_ws_")
    os.makedirs(ws, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(ws)   # the reference's paths are relative to the working directory
    try:
        prepare_workspace(ws, args)
        return main(args)
    finally:
        os.chdir(cwd)


if __name__ == "__main__":
    try:
        out = run()
        print(out["result"])
    except Exception:   # the reference logs the traceback through logzero as well (:362-368)
        var = traceback.format_exc()
        print(var)
        logzero.logger.error(var)
        raise
