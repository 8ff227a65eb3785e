"""GPU: PPO learner kernels (process_fn + learn) vs the reference's recorded update and vs the torch-fp32 restatement."""
import os

import numpy as np
import pytest
import torch

import nn_oracle
import rolloutcase
from test_oracle_learn import POL, load_learn

pytestmark = pytest.mark.gpu


def upload_traj(traj, acts, rews, dones, lens, obs_bts, value_bt, logp_bt):
    """Fill a device Trajectory (time-major) from env-major host arrays."""
    B, T = acts.shape
    a = np.where(np.arange(T)[None, :] < lens[:, None], acts, -1)
    traj.act.copy_(torch.as_tensor(a.T.copy()))
    traj.rew.copy_(torch.as_tensor(rews.T.copy()))
    traj.done.copy_(torch.as_tensor(dones.T.astype(np.uint8).copy()))
    traj.obs.copy_(torch.as_tensor(np.ascontiguousarray(obs_bts.transpose(1, 0, 2))))
    traj.value.copy_(torch.as_tensor(value_bt.T.copy()))
    traj.logp.copy_(torch.as_tensor(logp_bt.T.copy()))


def rollout_time_value_logp(pp, obs_bts, acts, lens):
    B, T = acts.shape
    value = np.zeros((B, T), np.float32); logp = np.zeros((B, T), np.float32)
    with torch.no_grad():
        for b in range(B):
            L = int(lens[b])
            logits, v = nn_oracle.policy_forward(pp, obs_bts[b, :L])
            _, lp, _ = nn_oracle.categorical_logp_entropy(logits, acts[b, :L])
            value[b, :L] = v.numpy(); logp[b, :L] = lp.numpy()
    return value, logp


def make_learner(pp, I, B, T, hyper):
    from cirs_hip.learner import DeviceLearner, flat_policy_params
    gamma, lam, eps_clip, vf_coef, ent_coef, mgn, lr, bs, rep = hyper
    flat, views = flat_policy_params(I, init={POL[k]: v for k, v in pp.items()})
    ln = DeviceLearner(flat, I, B, T, gamma=gamma, gae_lambda=lam, eps_clip=eps_clip, vf_coef=vf_coef, ent_coef=ent_coef,
                       max_grad_norm=mgn, lr=lr, norm_adv=True, value_clip=True, rew_norm=True)
    return ln, views


def test_learner_matches_reference_golden(golden_dir):
    from cirs_hip.rollout import Trajectory
    z, tp, pp, perms = load_learn(golden_dir)
    U, I, B, T = [int(v) for v in z["dims"]]
    lens = z["lens"]
    obs_bts = z["obs"]  # reference tracker states [B, T+1, S]
    value, logp = rollout_time_value_logp(pp, obs_bts, np.maximum(z["acts"], 0), lens)
    traj = Trajectory(B, T, 20, "cuda")
    upload_traj(traj, z["acts"], z["rews"], z["dones"], lens, obs_bts, value, logp)
    ln, views = make_learner(pp, I, B, T, z["hyper"])
    n = ln.prepare(traj, lens)
    assert n == int(lens.sum())
    np.testing.assert_allclose(ln.b_adv[:n].cpu().numpy(), z["b_adv"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ln.b_ret[:n].cpu().numpy(), z["b_returns"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ln.b_vs[:n].cpu().numpy(), z["b_v_s"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ln.b_logp[:n].cpu().numpy(), z["b_logp_old"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(ln.b_act[:n].cpu().numpy(), z["b_act"].astype(np.int64))
    np.testing.assert_allclose(ln.rms_state.cpu().numpy(), z["ret_rms"], rtol=1e-5)
    bs, rep = int(z["hyper"][7]), int(z["hyper"][8])
    losses = ln.learn(bs, rep, perms=perms).cpu().numpy()
    # per-minibatch loss / clip / vf / ent
    np.testing.assert_allclose(losses[:, 0], z["loss"], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(losses[:, 1], z["loss_clip"], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(losses[:, 2], z["loss_vf"], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(losses[:, 3], z["loss_ent"], rtol=3e-4, atol=3e-5)
    # post-update actor / critic parameters incl. the duplicated-trunk Adam / clip quirk
    for k, name in POL.items():
        post = z["post_pol_" + name]
        np.testing.assert_allclose(views[name].cpu().numpy().reshape(post.shape), post, rtol=1e-4, atol=3e-6, err_msg=name)
    # gradient handed to the tracker: d loss / d obs of the last repeat, vs autograd on the restatement
    z2, tp2, pp2, perms2 = load_learn(golden_dir)
    gamma, lam, eps_clip, vf_coef, ent_coef, mgn, lr, _, _ = z["hyper"]
    out = nn_oracle.ppo_update(tp2, pp2, z["users"], z["acts"], z["rews"], z["dones"], lens, perms2, gamma=gamma, lam=lam,
                               eps_clip=eps_clip, vf_coef=vf_coef, ent_coef=ent_coef, max_grad_norm=mgn, lr=lr, batch_size=bs, repeat=rep)
    dobs = ln.dobs.cpu().numpy()  # [T+1, B, S]
    env = ln.b_env[:n].cpu().numpy(); tt = ln.b_t[:n].cpu().numpy()
    np.testing.assert_allclose(dobs[tt, env], out["dobs_rows"], rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("I,B,T,bs,ent_coef", [(3327, 64, 30, 1024, 0.0), (10728, 160, 30, 1024, 0.01), (10728, 1024, 30, 1024, 0.0)])
def test_learner_vs_restatement_large(I, B, T, bs, ent_coef):
    """BASELINE catalogue sizes (C2: 3327 items, C3: 10728 items), merged last minibatch > 1024 rows, non-zero entropy coef;
    last case = the benchmarked learner workload: 1024 envs, ~19-27 k rows, ~2 x 19+ minibatch steps, merged last minibatch."""
    import policycase
    from cirs_hip.rollout import Trajectory
    U = 300
    rng = np.random.RandomState(I)
    tp = rolloutcase.tracker_param_dict(U, I, T, seed=1)
    arrs = policycase.random_weights(rng, I, head_scale=1.5)
    pp = {k: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)) for k, v in arrs.items()}
    lens = rng.randint(8, T + 1, size=B)
    users = rng.randint(0, U, B); acts = rng.randint(0, I, (B, T)); rews = rng.uniform(0, 1, (B, T))
    dones = np.zeros((B, T), bool); dones[np.arange(B), lens - 1] = True
    with torch.no_grad():
        obs_bts = nn_oracle.tracker_states(tp, users, acts, rews).numpy()
    value, logp = rollout_time_value_logp(pp, obs_bts, acts, lens)
    n = int(lens.sum())
    perms = [rng.permutation(n) for _ in range(2)]
    hyper = [0.95, 0.95, 0.2, 0.25, ent_coef, 0.5, 1e-3, bs, 2]
    traj = Trajectory(B, T, 20, "cuda")
    upload_traj(traj, acts, rews, dones, lens, obs_bts, value, logp)
    ln, views = make_learner(pp, I, B, T, hyper)
    assert ln.prepare(traj, lens) == n
    losses = ln.learn(bs, 2, perms=perms).cpu().numpy()
    tp_o = {k: v.clone() for k, v in tp.items()}
    pp_o = {k: v.clone() for k, v in pp.items()}
    out = nn_oracle.ppo_update(tp_o, pp_o, users, acts, rews, dones, lens, perms, gamma=0.95, lam=0.95, eps_clip=0.2, vf_coef=0.25,
                               ent_coef=ent_coef, max_grad_norm=0.5, lr=1e-3, batch_size=bs, repeat=2)
    np.testing.assert_allclose(ln.b_adv[:n].cpu().numpy(), out["adv"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ln.b_ret[:n].cpu().numpy(), out["returns"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(losses[:, 0], out["loss"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(losses[:, 3], out["ent"], rtol=1e-4, atol=1e-4)
    for k, name in POL.items():
        got = views[name].cpu().numpy().reshape(pp_o[k].shape)
        np.testing.assert_allclose(got, pp_o[k].numpy(), rtol=1e-3, atol=2e-5, err_msg=name)
    dobs = ln.dobs.cpu().numpy()
    env = ln.b_env[:n].cpu().numpy(); tt = ln.b_t[:n].cpu().numpy()
    np.testing.assert_allclose(dobs[tt, env], out["dobs_rows"], rtol=5e-3, atol=1e-6)
