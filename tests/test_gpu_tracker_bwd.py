"""GPU: analytic tracker backward (BPTT through the rollout) vs torch autograd on the fp32 restatement, and the full
update (policy + tracker Adam) vs the reference's recorded post-update tracker parameters."""
import os

import numpy as np
import pytest
import torch

import nn_oracle
from conftest import close
import rolloutcase
from test_gpu_learn import make_learner, rollout_time_value_logp, upload_traj
from test_oracle_learn import POL, load_learn

pytestmark = pytest.mark.gpu


def device_tracker_trainable(tp, U, I, B, T, nhead=4, lr=1e-3):
    from cirs_hip.tracker import DeviceTracker, flat_tracker_params, tracker_param_shapes
    shapes = tracker_param_shapes(U, I)
    flat, views = flat_tracker_params(shapes, init=tp)
    params = dict(views)
    params["pos_encoder.pe"] = tp["pos_encoder.pe"].float().cuda().contiguous()
    trk = DeviceTracker(params, U, I, B, T, nhead=nhead)
    trk.enable_training(flat, lr=lr)
    return trk, views


def replay_tracker(trk, users, acts, rews, lens):
    """Teacher-forced forward so x_hist / caches are populated exactly as a rollout would leave them."""
    B, T = acts.shape
    trk.reset()
    trk.init(torch.as_tensor(users))
    for t in range(T):
        live = np.where(lens > t)[0]
        if len(live) == 0:
            break
        trk.step(torch.as_tensor(acts[live, t]), torch.as_tensor(rews[live, t]), env_ids=torch.as_tensor(live.astype(np.int32)).cuda())


def rows_of(lens):
    B = len(lens)
    offsets = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    row_env = np.repeat(np.arange(B), lens).astype(np.int32)
    row_t = np.concatenate([np.arange(l) for l in lens]).astype(np.int32)
    return offsets, row_env, row_t


# kernels: "fused" = the round-3 default (row-chain kernels; per-episode attention when max_len <= 64 and its LDS image fits 64 KB: T = 100 and
# T = 55 take the per-row attention kernels), "rows" = one launch per op (the path of long episodes / A-B runs), selected per call
@pytest.mark.parametrize("kernels", ["fused", "rows"])
@pytest.mark.parametrize("U,I,B,T,nhead,hot", [(50, 80, 9, 12, 4, 0.0), (300, 500, 70, 30, 4, 0.0), (40, 60, 5, 100, 8, 0.0),
                                                (30, 40, 6, 7, 1, 0.0), (8, 90, 48, 30, 4, 0.6), (30, 40, 6, 20, 2, 0.0),
                                                (30, 40, 6, 14, 8, 0.0), (20, 30, 4, 40, 4, 0.0), (20, 30, 4, 55, 4, 0.0)])
def test_tracker_backward_matches_autograd(U, I, B, T, nhead, hot, kernels, monkeypatch):
    from cirs_hip.rollout import Trajectory
    if kernels == "rows":
        monkeypatch.setenv("CIRS_TRACKER_ROWS_UNFUSED", "1"); monkeypatch.setenv("CIRS_TRACKER_ATTN_ROWS", "1")
    rng = np.random.RandomState(B * T)
    tp = rolloutcase.tracker_param_dict(U, I, T, seed=3)
    lens = rng.randint(2, T + 1, size=B)
    users = rng.randint(0, U, B); acts = rng.randint(0, I, (B, T)); rews = rng.uniform(0, 1, (B, T))
    acts[:, ::5] = acts[:, :1]  # repeated items: several rows hit the same embedding row
    if hot > 0:  # a concentrated policy: one item (and few users) own hundreds of rows -> embedding-gradient segments that span
        acts[rng.uniform(size=acts.shape) < hot] = 7   # many 64-row sub-runs of the sorted scatter
    G = rng.normal(size=(T + 1, B, 20)).astype(np.float32)
    # autograd reference
    tpo = {k: v.clone() for k, v in tp.items()}
    for k, v in tpo.items():
        if k != "pos_encoder.pe":
            v.requires_grad_(True)
    states = nn_oracle.tracker_forward_all(tpo, nn_oracle.tracker_inputs(tpo, users, acts, rews), nhead)  # [B, T+1, S]
    up = torch.zeros_like(states)
    for b in range(B):
        up[b, :lens[b]] = torch.as_tensor(G[:lens[b], b])
    (states * up).sum().backward()
    # device
    trk, views = device_tracker_trainable(tp, U, I, B, T, nhead=nhead)
    replay_tracker(trk, users, acts, rews, lens)
    traj = Trajectory(B, T, 20, "cuda")
    a = np.where(np.arange(T)[None, :] < lens[:, None], acts, -1)
    traj.act.copy_(torch.as_tensor(a.T.copy())); traj.rew.copy_(torch.as_tensor(rews.T.copy()))
    offsets, row_env, row_t = rows_of(lens)
    d = lambda x: torch.as_tensor(x).cuda()  # noqa: E731
    trk.backward(torch.as_tensor(users), traj, d(row_env), d(row_t), d(offsets), d(lens.astype(np.int32)), int(lens.sum()), d(G))
    for k, gv in trk.grad_views.items():
        want = tpo[k].grad.numpy()
        got = gv.cpu().numpy()
        if k.endswith("self_attn.in_proj_bias"):
            # key-bias gradient is analytically 0: autograd returns round-off noise, we return (near) zero
            assert np.abs(got[32:64]).max() < 1e-4 * max(1.0, np.abs(got).max())
            want, got = np.delete(want, slice(32, 64)), np.delete(got, slice(32, 64))
        scale = np.abs(want).max() + 1e-12
        np.testing.assert_allclose(got / scale, want / scale, atol=2e-4, err_msg=k)


def test_full_update_matches_reference_tracker_params(golden_dir):
    """collect (teacher-forced) -> learn -> tracker backward -> optim_state.step(): post-update tracker == reference."""
    from cirs_hip.rollout import Trajectory
    z, tp, pp, perms = load_learn(golden_dir)
    U, I, B, T = [int(v) for v in z["dims"]]
    lens = z["lens"]
    trk, views = device_tracker_trainable(tp, U, I, B, T, lr=float(z["hyper"][6]))
    acts = np.maximum(z["acts"], 0)
    replay_tracker(trk, z["users"], acts, z["rews"], lens)
    obs_bts = z["obs"]
    value, logp = rollout_time_value_logp(pp, obs_bts, acts, lens)
    traj = Trajectory(B, T, 20, "cuda")
    upload_traj(traj, z["acts"], z["rews"], z["dones"], lens, obs_bts, value, logp)
    ln, pviews = make_learner(pp, I, B, T, z["hyper"])
    n = ln.prepare(traj, lens)
    ln.learn(int(z["hyper"][7]), int(z["hyper"][8]), perms=perms)
    offsets, _, _ = rows_of(lens)
    trk.backward(torch.as_tensor(z["users"]), traj, ln.b_env, ln.b_t, torch.as_tensor(offsets).cuda(),
                 torch.as_tensor(lens.astype(np.int32)).cuda(), n, ln.dobs)
    trk.adam_update()
    moved = 0
    for k, v in views.items():
        pre, post = z["trk_" + k], z["post_trk_" + k]
        got = v.cpu().numpy()
        if k.endswith("self_attn.in_proj_bias"):
            got, post, pre = (np.delete(x, slice(32, 64)) for x in (got, post, pre))
        moved += int(np.abs(post - pre).max() > 0)
        # Adam's first step is +-lr * g/(|g|+eps): entries whose gradient is ~1e-8 flip with round-off, so compare
        # where the reference moved by (almost) the full lr, and bound the rest by lr
        full = np.abs(np.abs(post - pre) - 1e-3) < 2e-5
        still = post == pre  # e.g. embedding rows of users/items that never occurred: exactly zero gradient
        close(got[full], post[full], 1e-5, 2e-6, "full update, tracker post-Adam: " + k)
        np.testing.assert_array_equal(got[still], pre[still], err_msg=k)
        assert np.abs(got - pre).max() <= 1e-3 * 1.01
        assert (full | still).mean() > 0.9, k
    assert moved >= 20


def test_two_consecutive_updates_match_reference(golden_dir):
    """TWO consecutive collect (teacher-forced) + update rounds on the same learner / tracker state (tests/golden/learn.npz
    r2_*, recorded from the reference's own second Collector.collect + policy.update): pins ret_rms carry-over, the Adam
    moments of both optimisers across updates and — through the tracker's second Adam step, whose size depends on g2/g1 —
    the MAGNITUDES of the tracker gradients (the first step only pins their signs)."""
    from cirs_hip.rollout import Trajectory
    from test_oracle_learn import compare_tracker_second_step, load_round2
    z, tp, pp, perms = load_learn(golden_dir)
    U, I, B, T = [int(v) for v in z["dims"]]
    bs, rep, lr = int(z["hyper"][7]), int(z["hyper"][8]), float(z["hyper"][6])
    trk, views = device_tracker_trainable(tp, U, I, B, T, lr=lr)
    ln, pviews = make_learner(pp, I, B, T, z["hyper"])
    traj = Trajectory(B, T, 20, "cuda")
    for pre, pr in (("", perms), ("r2_", load_round2(z))):
        lens = z[pre + "lens"]
        acts = np.maximum(z[pre + "acts"], 0)
        replay_tracker(trk, z[pre + "users"], acts, z[pre + "rews"], lens)
        pp_now = {k: pviews[name].detach().cpu().reshape(pp[k].shape).clone() for k, name in POL.items()}
        value, logp = rollout_time_value_logp(pp_now, z[pre + "obs"], acts, lens)
        traj.clear()
        upload_traj(traj, z[pre + "acts"], z[pre + "rews"], z[pre + "dones"], lens, z[pre + "obs"], value, logp)
        n = ln.prepare(traj, lens)
        close(ln.b_vs[:n].cpu().numpy(), z[pre + "b_v_s"], 1e-5, 2e-6, "two updates " + pre + "b_v_s")
        close(ln.b_ret[:n].cpu().numpy(), z[pre + "b_returns"], 1e-5, 2e-6, "two updates " + pre + "b_returns")
        close(ln.b_adv[:n].cpu().numpy(), z[pre + "b_adv"], 1e-5, 2e-6, "two updates " + pre + "b_adv")
        np.testing.assert_allclose(ln.rms_state.cpu().numpy(), z[pre + "ret_rms"], rtol=1e-5)
        losses = ln.learn(bs, rep, perms=pr).cpu().numpy()
        close(losses[:, 0], z[pre + "loss"], 1e-5, 1e-5, "two updates " + pre + "loss")
        close(losses[:, 2], z[pre + "loss_vf"], 1e-5, 1e-5, "two updates " + pre + "loss_vf")
        offsets, _, _ = rows_of(lens)
        trk.backward(torch.as_tensor(z[pre + "users"]), traj, ln.b_env, ln.b_t, torch.as_tensor(offsets).cuda(),
                     torch.as_tensor(lens.astype(np.int32)).cuda(), n, ln.dobs)
        trk.adam_update()
        for k, name in POL.items():
            post = z[pre + "post_pol_" + name]
            close(pviews[name].cpu().numpy().reshape(post.shape), post, 1e-5, 1e-5, "two updates " + pre + name)
    compare_tracker_second_step({k: v.cpu().numpy() for k, v in views.items()}, z, lr=lr)
