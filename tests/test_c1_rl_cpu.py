"""CPU: the RL loop of BASELINE configs[0] (VirtualTaobao, 4 parallel envs, CPU plumbing; reference CIRS-RL-taobao.py:150-300) through
the mirror's public classes -- DummyVectorEnv over SimulatedEnv(VirtualTB), StateTrackerTransformer(dataset="VirtualTB-v0") with
dropout live, ActorProb / Critic, PPOPolicy with Independent(Normal) and action scaling, Collector -- against two collect + update
rounds recorded from the reference itself with the same seeds (tests/golden/c1rl.npz, oracle/gen_golden.py gen_c1rl).  The host
stack draws from torch's / numpy's generators in the reference's order, so trajectories, losses and updated parameters agree to
float round-off."""
import collections
import os

import numpy as np
import pytest
import torch


def _build(z):
    import gym
    from gym.envs.registration import register
    from torch.distributions import Independent, Normal
    from core.collector import Collector
    from core.inputs import get_dataset_columns
    from core.policy.ppo import PPOPolicy
    from core.state_tracker import StateTrackerTransformer
    from core.user_model import compute_input_dim
    from core.user_model_mmoe import UserModel_MMOE
    from deepctr_torch.inputs import DenseFeat
    from tianshou.data import VectorReplayBuffer
    from tianshou.env import DummyVectorEnv
    from tianshou.utils.net.common import Net
    from tianshou.utils.net.continuous import ActorProb, Critic
    N, thr, T, B, dim_model, dim_state = (z["cfg"][0], z["cfg"][1], z["cfg"][2], z["cfg"][3], z["cfg"][4], z["cfg"][5])
    N, T, B, dim_model, dim_state = int(N), int(T), int(B), int(dim_model), int(dim_state)
    register(id="VirtualTB-v0", entry_point="environments.VirtualTaobao.virtualTB.envs.virtualTB:VirtualTB",
             kwargs=dict(num_leave_compute=N, leave_threshold=float(thr), max_turn=T))
    x_columns, y_columns = [DenseFeat("user_feat", 91), DenseFeat("feat_item", 27)], [DenseFeat("y", 1)]
    tasks = collections.OrderedDict({f.name: "regression" for f in y_columns})
    model = UserModel_MMOE(x_columns, y_columns, len(tasks), tasks, {f.name: f.dimension for f in y_columns}, dnn_hidden_units=(128, 128),
                           seed=2022, device="cpu")
    model.load_state_dict({k[len("mmoe_"):]: torch.as_tensor(z[k]) for k in z.files if k.startswith("mmoe_")})
    model.eval()
    register(id="SimulatedEnv-v0", entry_point="core.env.simulatedEnv.simulated_env:SimulatedEnv",
             kwargs=dict(user_model=model, task_name="VirtualTB-v0", version="v1", tau=10.0, gamma_exposure=3.0))
    sim = gym.make("SimulatedEnv-v0")
    train_envs = DummyVectorEnv([lambda: gym.make("SimulatedEnv-v0") for _ in range(B)])
    uc, ac, fc, hu, ha, hf = get_dataset_columns(dim_model, envname="VirtualTB-v0")
    assert dim_model == compute_input_dim(ac)
    tracker = StateTrackerTransformer(uc, ac, fc, dim_model=dim_model, dim_state=dim_state, dim_max_batch=B, dataset="VirtualTB-v0",
                                      has_user_embedding=hu, has_action_embedding=ha, has_feedback_embedding=hf, nhead=3, d_hid=128,
                                      nlayers=2, dropout=0.1, device="cpu", seed=2022, MAX_TURN=T)
    net = Net(dim_state, hidden_sizes=[64, 64], device="cpu")
    actor = ActorProb(net, sim.action_space.shape, max_action=sim.action_space.high[0], device="cpu")
    critic = Critic(net, device="cpu")
    for mod, tag in ((actor, "actor"), (critic, "critic"), (tracker, "tracker")):
        sd = {k[len(f"init_{tag}_"):]: torch.as_tensor(z[k]) for k in z.files if k.startswith(f"init_{tag}_")}
        missing = mod.load_state_dict(sd, strict=True)
        assert not getattr(missing, "missing_keys", []) and not getattr(missing, "unexpected_keys", [])
    optim = [torch.optim.Adam(list(actor.parameters()) + list(critic.parameters()), lr=1e-3), torch.optim.Adam(tracker.parameters(), lr=1e-3)]
    policy = PPOPolicy(actor, critic, optim, lambda *logits: Independent(Normal(*logits), 1), discount_factor=0.95, max_grad_norm=0.5,
                       eps_clip=0.2, vf_coef=0.25, ent_coef=0.0, reward_normalization=1, advantage_normalization=1, recompute_advantage=0,
                       value_clip=1, gae_lambda=0.95, action_space=sim.action_space)
    collector = Collector(policy, train_envs, VectorReplayBuffer(400, B), preprocess_fn=tracker.build_state)
    return sim, tracker, actor, critic, policy, collector, B


def test_c1_rl_loop_reproduces_the_reference(golden_dir):
    import warnings
    warnings.simplefilter("ignore")
    z = np.load(os.path.join(golden_dir, "c1rl.npz"))
    sim, tracker, actor, critic, policy, collector, B = _build(z)
    # the public classes dispatched to the host stack (no GPU, no libcirs_hip call anywhere on this path)
    assert type(tracker).__name__ == "HostStateTracker" and type(policy).__name__ == "HostPPOPolicy" and type(collector).__name__ == "HostCollector"
    np.testing.assert_array_equal(sim.action_space.low, z["action_low"])
    for rnd in range(2):
        torch.manual_seed(100 + rnd); np.random.seed(200 + rnd)
        res = collector.collect(n_episode=B)
        buf = collector.buffer
        idx = buf.sample_index(0)
        b = buf[idx]
        np.testing.assert_array_equal(idx, z[f"r{rnd}_idx"])
        assert [res["n/ep"], res["n/st"]] == z[f"r{rnd}_res_n"].tolist()
        np.testing.assert_array_equal(res["lens"], z[f"r{rnd}_res_lens"])
        np.testing.assert_array_equal(res["idxs"], z[f"r{rnd}_res_idxs"])
        np.testing.assert_array_equal(np.asarray(b.done), z[f"r{rnd}_done"])
        np.testing.assert_allclose(np.asarray(b.act), z[f"r{rnd}_act"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(np.asarray(b.rew, np.float64), z[f"r{rnd}_rew"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(res["rews"], z[f"r{rnd}_res_rews"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(b.obs.detach().numpy(), z[f"r{rnd}_obs"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(b.obs_next.detach().numpy(), z[f"r{rnd}_obs_next"], rtol=1e-4, atol=1e-5)
        losses = policy.update(0, buf, batch_size=16, repeat=2)
        for k, v in losses.items():
            np.testing.assert_allclose(np.array(v), z[f"r{rnd}_loss_" + k.replace("/", "_")], rtol=2e-4, atol=2e-5, err_msg=k)
        np.testing.assert_allclose([policy.ret_rms.mean, policy.ret_rms.var, policy.ret_rms.count], z[f"r{rnd}_ret_rms"], rtol=1e-5)
        # the update leaves torch's generator where the reference's leaves it (its process_fn / learn sample through forward()): a run that
        # seeds ONCE keeps drawing the reference's actions and dropout masks in every later collect
        np.testing.assert_array_equal(torch.rand(4).numpy(), z[f"r{rnd}_rng_probe"])
        for mod, tag in ((actor, "actor"), (critic, "critic"), (tracker, "tracker")):
            for k, v in mod.state_dict().items():
                got, want = v.detach().numpy(), z[f"r{rnd}_{tag}_{k}"]
                if k.endswith("self_attn.in_proj_bias"):
                    # the key bias has an analytically zero gradient (soft-max is shift invariant): Adam turns its float round-off
                    # into +-lr steps in the reference and here alike -- not a parity target (note iv)
                    D = got.shape[0] // 3
                    got, want = np.r_[got[:D], got[2 * D:]], np.r_[want[:D], want[2 * D:]]
                np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-5, err_msg=f"round {rnd} {tag}.{k}")


def test_c1_map_action_and_deterministic_eval(golden_dir):
    z = np.load(os.path.join(golden_dir, "c1rl.npz"))
    sim, tracker, actor, critic, policy, collector, B = _build(z)
    raw = np.array([[-3.0] * 27, [0.0] * 27, [0.5] * 27], np.float32)
    got = policy.map_action(raw)
    low, high = sim.action_space.low, sim.action_space.high
    np.testing.assert_allclose(got[0], low); np.testing.assert_allclose(got[1], (low + high) / 2); np.testing.assert_allclose(got[2], low + (high - low) * 0.75)


def test_c1_trainer_epoch_on_the_host(golden_dir, tmp_path):
    """CIRS-RL-taobao.py:247-300: train / test Collectors, BasicLogger over a SummaryWriter, LoggerCallback_RL, onpolicy_trainer -- one
    short epoch end to end on the CPU."""
    import warnings
    warnings.simplefilter("ignore")
    import gym
    from torch.utils.tensorboard import SummaryWriter
    from core.collector import Collector
    from core.trainer.onpolicy import onpolicy_trainer
    from tianshou.env import DummyVectorEnv
    from tianshou.utils import BasicLogger
    from util.utils import LoggerCallback_RL
    z = np.load(os.path.join(golden_dir, "c1rl.npz"))
    sim, tracker, actor, critic, policy, train_collector, B = _build(z)
    test_envs = DummyVectorEnv([lambda: gym.make("VirtualTB-v0") for _ in range(B)])
    test_collector = Collector(policy, test_envs, preprocess_fn=tracker.build_state)
    cb = LoggerCallback_RL(str(tmp_path / "log.txt"))
    policy.callbacks = [cb]
    before = torch.cat([p.detach().reshape(-1).clone() for p in tracker.parameters()])
    torch.manual_seed(1); np.random.seed(1)
    res = onpolicy_trainer(policy, train_collector, test_collector, tracker, 1, 60, 2, B, 16, episode_per_collect=B,
                           logger=BasicLogger(SummaryWriter(str(tmp_path)), save_interval=1), verbose=False)
    # (collect() starts with reset() -> reset_stat(), like the reference: the counters describe the LAST collect)
    assert res["train_step"] == 9 * B and res["test_episode"] == B and np.isfinite(res["best_reward"])
    assert cb.last_results["num_test"] == B and float(cb.last_results["ctr"]) >= 0
    after = torch.cat([p.detach().reshape(-1) for p in tracker.parameters()])
    assert float((after - before).abs().max()) > 0      # the gradient reached the tracker through the stored states
