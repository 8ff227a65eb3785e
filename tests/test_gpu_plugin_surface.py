"""GPU: the mirrored plugin surface (core.collector / core.state_tracker / core.policy.ppo / tianshou.* /
environments.*) wired like CIRS-RL-kuaishou.py is a thin layer: collect + update give bit-identical trajectories
and parameters to driving the engines directly."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_example():
    spec = importlib.util.spec_from_file_location("cirs_rl_kuaishou_synth", os.path.join(ROOT, "examples", "cirs_rl_kuaishou_synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_script_wiring_runs_and_matches_engine():
    ex = load_example()
    args = ex.get_args(["--n-users", "200", "--n-items", "500", "--training-num", "32", "--episode-per-collect", "32",
                        "--batch-size", "64", "--max_turn", "15", "--tau", "10", "--dropout", "0"])
    tab, train_envs, st, policy, coll = ex.build(args)
    # state_dict names of the reference
    assert "embedding_dict.feat_item.weight" in st.state_dict() and "transformer_encoder.layers.1.linear2.weight" in st.state_dict()
    assert "actor.last.model.0.weight" in policy.state_dict() and "critic.preprocess.model.model.2.bias" in policy.state_dict()
    assert coll.env.mat[0].shape == (200, 500)  # BaseVectorEnv.__getattr__ fan-out the callbacks rely on (evaluation.py:289)
    tp0 = {k: v.detach().clone() for k, v in st.state_dict().items()}
    pp0 = {k: v.detach().clone() for k, v in policy.views.items()}

    users = np.random.RandomState(5).randint(0, 200, 32)
    res = coll.collect(n_episode=32, users=users)
    assert res["n/ep"] == 32 and res["n/st"] == int(res["lens"].sum()) and len(res["rews"]) == 32
    buf = coll.buffer
    assert len(buf) == res["n/st"]
    batch, idx = buf.sample(0)
    assert batch.obs.shape == (res["n/st"], 20) and batch.act.shape == (res["n/st"],)
    assert bool(batch.done[np.cumsum(buf._lengths) - 1].all()) and int(batch.done.sum()) == 32
    assert np.array_equal(np.sort(res["lens"]), np.sort(buf._lengths)) and (np.diff(res["lens"]) >= 0).all()  # completion order
    assert np.array_equal(buf._lengths[res["idxs"] // buf.size], res["lens"])
    n = res["n/st"]
    perms = [np.random.RandomState(9 + k).permutation(n) for k in range(2)]
    losses = policy.update(0, buf, batch_size=64, repeat=2, perms=perms)
    assert set(losses) == {"loss", "loss/clip", "loss/vf", "loss/ent"} and np.isfinite(losses["loss"]).all()

    # the same thing through the engine, from the same initial parameters / users / seeds / permutations
    from cirs_hip.engine import CirsEngine
    dt = coll.env.workers[0].env_task.device_tables(normed_mat=tab.normed_mat, alpha_u=tab.alpha_u, beta_i=tab.beta_i)
    eng = CirsEngine(dt, 32, max_turn=15, num_leave_compute=args.num_leave_compute, leave_threshold=args.leave_threshold, tau=10.0,
                     gamma_exposure=args.gamma_exposure, seed=0, tracker_params=tp0, policy_params=pp0)
    eng.users = torch.as_tensor(users).to(eng.device, torch.int32)
    eng.lengths = eng.rollout.collect(eng.users, seed=coll.sampler_seed(), rng_base=0)
    assert torch.equal(eng.rollout.traj.act, coll._rollout.traj.act)
    assert torch.equal(eng.rollout.traj.rew, coll._rollout.traj.rew)
    assert torch.equal(eng.rollout.traj.obs, coll._rollout.traj.obs)
    eng.update(batch_size=64, repeat=2, perms=perms)
    assert torch.equal(eng.policy_flat, policy.flat)
    assert torch.equal(eng.tracker_flat, st.flat)


def test_ppo_options_through_the_plugin_surface():
    """PPOPolicy(dual_clip=..., recompute_advantage=True) (reference core/policy/ppo.py:73-99): accepted by the mirror and routed to the device
    learner (numbers pinned in tests/test_gpu_learn.py against the reference-recorded learn_opts.npz)."""
    from core.policy.ppo import PPOPolicy
    ex = load_example()
    args = ex.get_args(["--n-users", "100", "--n-items", "300", "--training-num", "16", "--episode-per-collect", "16",
                        "--batch-size", "64", "--max_turn", "12", "--tau", "10", "--dropout", "0"])
    tab, train_envs, st, policy, coll = ex.build(args)
    with pytest.raises(AssertionError):
        PPOPolicy(policy.actor, policy.critic, policy.optim, torch.distributions.Categorical, dual_clip=0.9)
    policy._hyper["dual_clip"], policy._recompute_adv = 1.5, True      # what PPOPolicy(dual_clip=1.5, recompute_advantage=True) stores
    res = coll.collect(n_episode=16)
    losses = policy.update(0, coll.buffer, batch_size=64, repeat=3)
    n = res["n/st"]
    assert np.isfinite(losses["loss"]).all() and len(losses["loss"]) == 3 * max(n // 64, 1)
    assert policy._learner.cfg.dual_clip == 1.5
    assert int(policy._learner.rms_state.cpu()[2]) == 3 * n      # RunningMeanStd updated by process_fn and before repeats 2 and 3


def test_vector_env_protocol_and_test_envs():
    """env.reset/step with numpy in/out (venvs.py:153-252) and the bare KuaishouEnv used by the test collectors."""
    ex = load_example()
    args = ex.get_args(["--n-users", "100", "--n-items", "300", "--training-num", "8", "--max_turn", "10"])
    tab, train_envs, st, policy, coll = ex.build(args)
    obs = train_envs.reset(users=np.arange(8))
    assert obs.shape == (8, 1) and obs.dtype == np.int64 and np.array_equal(obs[:, 0], np.arange(8))
    o, r, d, info = train_envs.step(np.arange(8) * 3, np.arange(8))
    assert o.shape == (8, 1) and r.dtype == np.float64 and d.dtype == bool and "CTR" in info[0] and info[3]["env_id"] == 3
    np.testing.assert_array_equal(o[:, 0], np.arange(8) * 3)
    from tianshou.env import DummyVectorEnv
    import gym
    test_envs = DummyVectorEnv([lambda: gym.make(args.env) for _ in range(4)])
    test_envs.reset(users=np.arange(4))
    o, r, d, info = test_envs.step(np.array([5, 6, 7, 8]), np.arange(4))
    np.testing.assert_array_equal(r, tab.mat[np.arange(4), [5, 6, 7, 8]])  # reward = mat[u, a] (kuaishouEnv.py:172)
    assert "cum_reward" in info[0]


def test_trainer_loop_with_test_collector_set():
    """onpolicy_trainer + CollectorSet(FB, NX_0, NX_k): masking and forced length run inside the fused rollout."""
    ex = load_example()
    args = ex.get_args(["--n-users", "150", "--n-items", "400", "--training-num", "16", "--episode-per-collect", "16", "--test-num", "8",
                        "--batch-size", "64", "--max_turn", "20", "--tau", "10", "--epoch", "2", "--step-per-epoch", "200",
                        "--force_length", "10", "--leave_threshold", "0", "--num_leave_compute", "1"])
    tab, train_envs, st, policy, coll = ex.build(args)
    cs = ex.build_test_collectors(args, policy, st)
    events = []

    class CB:
        def on_train_begin(self): events.append("begin")
        def on_epoch_begin(self, epoch): events.append(("epoch", epoch))
        def on_epoch_end(self, epoch, results): events.append(("end", epoch, sorted(results)[:3]))
        def on_train_end(self): events.append("done")

    policy.callbacks = [CB()]
    from core.trainer.onpolicy import onpolicy_trainer
    before = policy.flat.clone(); tbefore = st.flat.clone()
    info = onpolicy_trainer(policy, coll, cs, st, args.epoch, args.step_per_epoch, args.repeat_per_collect, args.test_num, args.batch_size,
                            episode_per_collect=args.episode_per_collect, save_model_fn=lambda epoch, policy: None, verbose=False)
    assert events[0] == "begin" and events[-1] == "done" and ("epoch", 2) in events
    assert info["train_step"] >= 2 * args.step_per_epoch and info["test_episode"] == 3 * 8   # pre-training evaluation + one per epoch (reference onpolicy.py:126)
    assert float((policy.flat - before).abs().max()) > 0 and float((st.flat - tbefore).abs().max()) > 0
    res = cs.collect(n_episode=8)
    assert {"n/st", "rew", "NX_0_n/st", "NX_0_rew", "NX_10_lens"} <= set(res)
    assert (res["NX_10_lens"] == 10).all()                      # force_length
    nx = cs.collector_dict["NX_0"].buffer
    acts = nx._traj.act.cpu().numpy()
    assert np.array_equal(np.sort(res["NX_0_lens"]), np.sort(nx._lengths))     # result arrays come in completion order
    for b in range(8):                                           # remove_recommended_ids: no item twice in an episode
        a = acts[:nx._lengths[b], b]
        assert len(set(a.tolist())) == len(a) and (a >= 0).all()
    # the training tracker state (Adam step counter) survived the differently sized test engines
    assert st.adam_steps >= 2


def test_trainer_with_coverage_and_logger_callbacks():
    """The evaluation callbacks of CIRS-RL-kuaishou.py:303-316 on the device trajectories of the three test collectors."""
    ex = load_example()
    args = ex.get_args(["--n-users", "150", "--n-items", "400", "--training-num", "16", "--episode-per-collect", "16", "--test-num", "8",
                        "--batch-size", "64", "--max_turn", "20", "--tau", "10", "--epoch", "1", "--step-per-epoch", "100",
                        "--force_length", "10", "--leave_threshold", "0", "--num_leave_compute", "1"])
    tab, train_envs, st, policy, coll = ex.build(args)
    cs = ex.build_test_collectors(args, policy, st)
    policy.callbacks = cbs = ex.build_callbacks(args, tab, cs)
    from core.trainer.onpolicy import onpolicy_trainer
    onpolicy_trainer(policy, coll, cs, st, args.epoch, args.step_per_epoch, args.repeat_per_collect, args.test_num, args.batch_size,
                     episode_per_collect=args.episode_per_collect, save_model_fn=lambda epoch, policy: None, verbose=False)
    line = cbs[1].last_results
    assert {"CV", "CV_turn", "ctr", "len_tra", "R_tra", "ifeat_feat", "NX_0_CV", "NX_10_CV_turn", "NX_10_ifeat_feat"} <= set(line)
    assert line["NX_10_len_tra"] == 10.0 and 0.0 < float(line["NX_0_CV_turn"]) <= 1.0
    # cross-check the FB numbers against a host recount of the collector's trajectory
    import evalcase
    acts = cs.collector_dict["FB"].buffer._rollout.traj.act.cpu().numpy()
    flags = cbs[0]._flags["ifeat_feat"].cpu().numpy()
    hit, n, fl = evalcase.oracle_counts(acts, tab.n_items, flags)
    assert line["CV"] == f"{hit / tab.n_items:.5f}" and line["CV_turn"] == f"{hit / n:.5f}" and line["ifeat_feat"] == fl / n
    assert line["len_tra"] == n / 8


def test_policy_learns_longer_and_more_rewarding_trajectories():
    """End-to-end sanity of the whole stack (collector -> env -> tracker -> PPO -> tracker BPTT): on a small synthetic
    catalogue the exit rule punishes repeating categories, so a learning policy must lengthen its trajectories and raise
    their reward.  Thresholds are far below what the run reaches (length 12 -> 27, reward 3 -> 9 in 10 epochs)."""
    ex = load_example()
    args = ex.get_args(["--n-users", "300", "--n-items", "800", "--training-num", "128", "--episode-per-collect", "128", "--test-num", "32",
                        "--max_turn", "30", "--tau", "10", "--leave_threshold", "1", "--num_leave_compute", "3", "--epoch", "10",
                        "--step-per-epoch", "6000", "--seed", "3"])
    tab, train_envs, st, policy, coll = ex.build(args)
    first = coll.collect(n_episode=128)
    from core.trainer.onpolicy import onpolicy_trainer
    cs = ex.build_test_collectors(args, policy, st)
    onpolicy_trainer(policy, coll, cs, st, args.epoch, args.step_per_epoch, args.repeat_per_collect, args.test_num, args.batch_size,
                     episode_per_collect=args.episode_per_collect, save_model_fn=lambda epoch, policy: None, verbose=False)
    last = coll.collect(n_episode=128)
    assert last["len"] > 1.4 * first["len"], (first["len"], last["len"])
    assert last["rew"] > 1.5 * first["rew"], (first["rew"], last["rew"])


def test_policy_forward_masks_recommended_ids_through_the_protocol():
    """PPOPolicy.forward(batch, buffer, remove_recommended_ids=True) of the per-step protocol (core/policy/ppo.py:133-163 with
    core/policy/utils.py:7-58): ids already in the running episodes of the buffer are never drawn again; finished envs drop out."""
    from tianshou.data import Batch, VectorReplayBuffer
    ex = load_example()
    args = ex.get_args(["--n-users", "60", "--n-items", "70", "--training-num", "6", "--max_turn", "40", "--dropout", "0"])
    tab, train_envs, st, policy, coll = ex.build(args)
    B, I = 6, 70
    buf = VectorReplayBuffer(B * 40, B)
    rng = np.random.RandomState(0)
    live = np.arange(B)
    seen = {b: [] for b in range(B)}
    obs = torch.randn(B, 20)
    for t in range(30):
        out = policy.forward(Batch(obs=obs[live]), buf, remove_recommended_ids=True)
        act = out.act.cpu().numpy()
        assert act.shape == (len(live),)
        for b, a in zip(live, act):
            assert a not in seen[b], f"env {b} was recommended item {a} twice"
            seen[b].append(int(a))
        done = (rng.uniform(size=len(live)) < 0.12)
        buf.add(Batch(obs=obs[live].numpy(), act=act, rew=np.ones(len(live)), done=done), buffer_ids=live)
        live = live[~done]
        if len(live) == 0:
            break
    assert max(len(v) for v in seen.values()) >= 10


def test_collect_with_random_policy_through_the_step_protocol():
    """Collector.collect(n_episode, random=True) (core/collector.py:225-227): actions from the action spaces, the loop one vector
    step at a time through env.step / build_state / buffer.add; checked against the C oracle env teacher-forced with the buffer's
    actions, the torch restatement of the tracker, and the result-dict arithmetic."""
    import envcase
    import nn_oracle
    ex = load_example()
    args = ex.get_args(["--n-users", "80", "--n-items", "120", "--training-num", "12", "--episode-per-collect", "12", "--max_turn", "9",
                        "--tau", "10", "--leave_threshold", "1", "--num_leave_compute", "3", "--dropout", "0"])
    tab, train_envs, st, policy, coll = ex.build(args)
    np.random.seed(4)
    users = np.random.RandomState(2).randint(0, 80, 12)
    res = coll.collect(n_episode=12, random=True, users=users)
    buf = coll.buffer
    lens = buf._lengths
    assert res["n/ep"] == 12 and res["n/st"] == lens.sum() == len(buf) and lens.min() >= 1 and lens.max() <= 9
    assert np.array_equal(np.sort(res["lens"]), np.sort(lens)) and (np.diff(res["lens"]) >= 0).all()
    acts = np.zeros((12, 9), np.int64); rews = np.zeros((12, 9))
    for b in range(12):
        sl = slice(buf._offset[b], buf._offset[b] + lens[b])
        acts[b, :lens[b]] = buf.act[sl]; rews[b, :lens[b]] = buf.rew[sl]
        assert buf.done[sl][-1] and not buf.done[sl][:-1].any()
    assert acts.max() < 120 and len(np.unique(acts)) > 30
    a_env, b_env = envcase.ab_env_tables(tab.raw_uid, tab.raw_pid, tab.alpha_u, tab.beta_i, 80, 120)
    host = envcase.HostEnv(envcase.env_cfg(80, 120, num_leave_compute=3, leave_threshold=1, max_turn=9, tau=10.0, gamma_exposure=args.gamma_exposure,
                                           version=1, r_decay=1.0, has_ab=True), tab.mat, tab.normed_mat, tab.dist, tab.item_cats, a_env, b_env, 12)
    want = envcase.run_teacher_forced(host, users, acts, 9)
    assert np.array_equal(want["length"], lens)
    for b in range(12):
        np.testing.assert_allclose(rews[b, :lens[b]], want["rew"][b, :lens[b]], rtol=1e-12)
    order = np.argsort(lens, kind="stable")
    np.testing.assert_allclose(res["rews"], np.array([rews[b, :lens[b]].sum() for b in order]), rtol=1e-12)
    tp = {k: v.detach().cpu() for k, v in st.state_dict().items()}
    states = nn_oracle.tracker_states(tp, users, acts, rews).detach().numpy()
    for b in range(12):
        sl = slice(buf._offset[b], buf._offset[b] + lens[b])
        np.testing.assert_allclose(buf.obs[sl].cpu().numpy(), states[b, :lens[b]], atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(buf.obs_next[sl].cpu().numpy(), states[b, 1:lens[b] + 1], atol=1e-4, rtol=1e-4)
