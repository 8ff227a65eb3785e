"""GPU: online-reward rollout (BASELINE configs[4] shape: catalogue too large for a U x I reward table).

The DeepFM pair scorer runs inside the rollout loop; checked (a) against the C oracle env fed with the same raw scores
(bit-level), (b) against the table mode whose normed_mat is the sweep of the same user model, (c) at a 2^20-item
catalogue through size-independent properties."""
import numpy as np
import pytest
import torch

import deepfmcase
import envcase
import rolloutcase

pytestmark = pytest.mark.gpu


def _item_side(rng, tab, n_feat=32):
    feats = np.where(tab.item_cats >= 0, tab.item_cats + 1, 0).astype(np.int32)  # ids shifted by one, 0 = padding
    assert feats.max() < n_feat
    dur = np.asarray(tab.duration, np.float32)
    return feats, dur


def _build(tab, B, T, seed, *, online):
    from cirs_hip.deepfm import DeviceDeepFM
    from cirs_hip.rollout import OnlineReward
    rng = np.random.RandomState(seed)
    w = deepfmcase.random_weights(rng, int(tab.raw_uid.max()) + 1, int(tab.raw_pid.max()) + 1, 16)
    feats, dur = _item_side(rng, tab)
    um = DeviceDeepFM(w)
    normed = um.normed_reward(tab.raw_uid, tab.raw_pid, feats, dur)
    _, mm = um.sweep(tab.raw_uid, tab.raw_pid, feats, dur, want_pred=False)
    kw = {}
    if online:
        kw["online"] = OnlineReward(um, tab.raw_uid, tab.raw_pid, feats, dur, mm, B)
    tab2 = type(tab)(**{**tab.__dict__, "normed_mat": normed.cpu().numpy()})
    ro, tp, arrs, envp = rolloutcase.build_device_stack(tab2, B, T, seed=seed, **kw)
    return ro, um, (w, feats, dur, mm.cpu().numpy()), envp, tab2


def test_online_rollout_matches_oracle_env_and_table_mode():
    from cirs_hip.synthetic import make_tables
    U, I, B, T = 96, 400, 64, 12
    tab = make_tables(U, I, seed=5, build_dist=False)
    ro, um, (w, feats, dur, mm), envp, tab2 = _build(tab, B, T, 11, online=True)
    users = torch.arange(B, dtype=torch.int32) % U
    ro.collect(users.cuda(), seed=77, rng_base=0)
    act = ro.traj.act.cpu().numpy().T            # [B, T], -1 after the episode end
    rew = ro.traj.rew.cpu().numpy().T
    done = ro.traj.done.cpu().numpy().T.astype(bool)
    length = ro.env.turn.cpu().numpy()
    assert length.min() >= 1 and length.min() < length.max()

    # (a) C oracle env, teacher-forced, fed with the pair scorer's raw scores (itself pinned to the reference DeepFM)
    cfg = envcase.env_cfg(U, I, num_leave_compute=envp["num_leave_compute"], leave_threshold=envp["leave_threshold"], max_turn=T,
                          tau=envp["tau"], gamma_exposure=envp["gamma_exposure"], version=1, r_decay=envp["r_decay"], has_ab=True,
                          dist_mode=1)
    host = envcase.HostEnv(cfg, None, None, None, tab.item_cats, envp["a_env"], envp["b_env"], B)
    host.reset(users.numpy())
    ready = np.arange(B)
    for t in range(T):
        if len(ready) == 0:
            break
        a = act[ready, t]
        assert (a >= 0).all()
        pred = deepfmcase.oracle_forward(w, tab.raw_uid[users.numpy()[ready]], tab.raw_pid[a], feats[a], dur[a])
        got = um.forward(tab.raw_uid[users.numpy()[ready]], tab.raw_pid[a], feats[a], dur[a]).cpu().numpy()
        np.testing.assert_allclose(got, pred, rtol=1e-5, atol=2e-6)
        host.set_online(got, mm)
        o, r, d, c, x = host.step(a, ready)
        np.testing.assert_allclose(rew[ready, t], r, rtol=1e-12, atol=1e-300)
        assert np.array_equal(done[ready, t], d)
        ready = ready[~d]
    assert (act[np.arange(B)[:, None], np.arange(T)[None, :]][np.arange(T)[None, :] >= length[:, None]] == -1).all()

    # (b) table mode with normed_mat = sweep of the same model: same rewards up to the sweep-vs-pair fp32 rounding
    host2 = envcase.HostEnv(envcase.env_cfg(U, I, num_leave_compute=envp["num_leave_compute"], leave_threshold=envp["leave_threshold"],
                                            max_turn=T, tau=envp["tau"], gamma_exposure=envp["gamma_exposure"], version=1,
                                            r_decay=envp["r_decay"], has_ab=True, dist_mode=1),
                            tab.mat, tab2.normed_mat, None, tab.item_cats, envp["a_env"], envp["b_env"], B)
    ref = envcase.run_teacher_forced(host2, users.numpy(), np.where(act < 0, 0, act), T)
    m = ~np.isnan(ref["rew"])
    assert np.array_equal(ref["length"], length)
    np.testing.assert_allclose(rew[m], ref["rew"][m], rtol=0, atol=2e-5)


def test_online_rollout_million_item_catalogue():
    """C5-lite: U = I = 2^20 (no U x I table exists), jaccard distances on the fly, DeepFM E = 16 scored online."""
    from cirs_hip.deepfm import DeviceDeepFM
    from cirs_hip.env import DeviceEnv, DeviceEnvTables
    from cirs_hip.policy import DevicePolicy
    from cirs_hip.rollout import DeviceRollout, OnlineReward
    from cirs_hip.tracker import DeviceTracker
    import policycase
    U = I = 1 << 20
    B, T = 256, 6
    rng = np.random.RandomState(3)
    cats = np.full((I, 4), -1, np.int32)
    ncat = rng.randint(1, 5, I)
    for q in range(4):
        col = rng.randint(0, 31, I)
        cats[:, q] = np.where(q < ncat, col, -1)
    feats = np.where(cats >= 0, cats + 1, 0).astype(np.int32)
    dur = rng.uniform(2, 60, I).astype(np.float32)
    w = deepfmcase.random_weights(rng, U, I, 16)
    um = DeviceDeepFM(w)
    # global (min, max): bounds from a sample of pairs, widened (the exact sweep over 2^40 pairs is the offline job)
    su = rng.randint(0, U, 1 << 16); si = rng.randint(0, I, 1 << 16)
    sample = um.forward(su, si, feats[si], dur[si])
    lo, hi = float(sample.min()) - 1.0, float(sample.max()) + 1.0
    dt = DeviceEnvTables(None, None, cats, n_users=U, n_items=I)
    env = DeviceEnv(dt, B, num_leave_compute=3, leave_threshold=1, max_turn=T, tau=10.0, gamma_exposure=10.0, dist_mode=1)
    tp = rolloutcase.tracker_param_dict(U, I, T, 4)
    trk = DeviceTracker({k: v.float().cuda().contiguous() for k, v in tp.items()}, U, I, B, T)
    arrs = policycase.random_weights(rng, I)
    pol = DevicePolicy({rolloutcase.POLICY_NAMES[k]: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)).cuda()
                        for k, v in arrs.items()}, I)
    ident = np.arange(I, dtype=np.int64)
    ro = DeviceRollout(env, trk, pol, online=OnlineReward(um, ident, ident, feats, dur, (lo, hi), B))
    users = torch.as_tensor(rng.randint(0, U, B).astype(np.int32))
    ro.collect(users.cuda(), seed=5)
    act = ro.traj.act.cpu().numpy().T; rew = ro.traj.rew.cpu().numpy().T
    length = ro.env.turn.cpu().numpy()
    valid = np.arange(T)[None, :] < length[:, None]
    assert (act[valid] >= 0).all() and (act[valid] < I).all() and (act[~valid] == -1).all()
    assert act[valid].max() > (1 << 19)                      # the sampler reaches the far end of the catalogue
    # first-step reward has no exposure effect: r = normalised online score exactly
    a0 = act[:, 0]
    raw = um.forward(users.numpy().astype(np.int64), a0, feats[a0], dur[a0]).cpu().numpy().astype(np.float64)
    want = (raw - np.float64(np.float32(lo))) / (np.float64(np.float32(hi)) - np.float64(np.float32(lo)))
    np.testing.assert_allclose(rew[:, 0], want, rtol=1e-12)
    assert (rew[valid] > 0).all() and (rew[valid] < 1).all()
    # determinism: the same seed reproduces the rollout bit for bit
    ro.collect(users.cuda(), seed=5)
    assert np.array_equal(ro.traj.act.cpu().numpy().T, act)
    assert np.array_equal(ro.traj.rew.cpu().numpy().T, rew)


def test_engine_collect_update_million_item_catalogue():
    """C5-lite end to end: collect (online reward) + PPO update + BPTT into the tracker at I = 2^20."""
    from cirs_hip.deepfm import DeviceDeepFM
    from cirs_hip.engine import CirsEngine
    from cirs_hip.env import DeviceEnvTables
    from cirs_hip.rollout import OnlineReward
    U = I = 1 << 20
    B, T = 128, 5
    rng = np.random.RandomState(9)
    cats = np.where(np.arange(4)[None, :] < rng.randint(1, 5, I)[:, None], rng.randint(0, 31, (I, 4)), -1).astype(np.int32)
    feats = np.where(cats >= 0, cats + 1, 0).astype(np.int32)
    dur = rng.uniform(2, 60, I).astype(np.float32)
    um = DeviceDeepFM(deepfmcase.random_weights(rng, U, I, 16))
    ident = np.arange(I, dtype=np.int64)

    def run():
        np.random.seed(123)   # (the learner draws its minibatch permutations from a seeded device generator)
        dt = DeviceEnvTables(None, None, cats, n_users=U, n_items=I)
        eng = CirsEngine(dt, B, max_turn=T, num_leave_compute=3, leave_threshold=1, tau=10.0, gamma_exposure=10.0, seed=1,
                         online_reward=OnlineReward(um, ident, ident, feats, dur, (-30.0, 30.0), B), batch_size_hint=128)
        before_p = eng.policy_flat.clone(); before_t = eng.tracker_flat.clone()
        lengths = eng.collect()
        losses, n = eng.update(batch_size=128, repeat=2)
        return eng, lengths, losses, n, before_p, before_t

    eng, lengths, losses, n, bp, bt = run()
    assert n == int(lengths.sum()) and n >= B
    assert torch.isfinite(losses).all() and losses.shape[1] == 4
    dp = (eng.policy_flat - bp); dtk = (eng.tracker_flat - bt)
    assert torch.isfinite(eng.policy_flat).all() and torch.isfinite(eng.tracker_flat).all()
    assert float(dp.abs().max()) > 0 and float(dp.abs().max()) <= 2 * 1e-3 * 2 * (n // 128 + 1)   # Adam: |step| <= ~lr each
    assert float(dtk.abs().max()) > 0
    eng2, lengths2, losses2, n2, _, _ = run()
    assert torch.equal(eng.policy_flat, eng2.policy_flat) and torch.equal(eng.tracker_flat, eng2.tracker_flat)


def test_c5_shape_hashed_ids_emb64():
    """configs[4] shape on one GPU: ids from an open vocabulary hashed into 2^20-row tables (splitmix64, bit-exact vs the
    oracle), DeepFM emb_dim = 64, online reward inside the rollout."""
    import ctypes as C
    import oracle_lib
    from cirs_hip.deepfm import DeviceDeepFM, hash_ids
    from cirs_hip.env import DeviceEnv, DeviceEnvTables
    from cirs_hip.policy import DevicePolicy
    from cirs_hip.rollout import DeviceRollout, OnlineReward
    from cirs_hip.tracker import DeviceTracker
    import policycase
    NB = 1 << 20
    rng = np.random.RandomState(21)
    raw = rng.randint(0, 1 << 62, 50000, dtype=np.int64)
    got = hash_ids(raw, NB).cpu().numpy()
    want = np.zeros_like(raw)
    assert oracle_lib.lib().oracle_hash_ids(raw.ctypes.data, raw.size, NB, want.ctypes.data) == 0
    np.testing.assert_array_equal(got, want)
    assert got.min() >= 0 and got.max() < NB and len(np.unique(got)) > 48000          # well spread
    # env over the hashed id space: user u / item i of the env ARE bucket ids
    U = I = NB
    B, T = 128, 4
    cats = np.where(np.arange(4)[None, :] < rng.randint(1, 5, I)[:, None], rng.randint(0, 31, (I, 4)), -1).astype(np.int32)
    feats = np.where(cats >= 0, cats + 1, 0).astype(np.int32)
    dur = rng.uniform(2, 60, I).astype(np.float32)
    um = DeviceDeepFM(deepfmcase.random_weights(rng, U, I, 64))
    dt = DeviceEnvTables(None, None, cats, n_users=U, n_items=I)
    env = DeviceEnv(dt, B, num_leave_compute=3, leave_threshold=1, max_turn=T, tau=10.0, gamma_exposure=10.0, dist_mode=1)
    tp = rolloutcase.tracker_param_dict(U, I, T, 4)
    trk = DeviceTracker({k: v.float().cuda().contiguous() for k, v in tp.items()}, U, I, B, T)
    arrs = policycase.random_weights(rng, I)
    pol = DevicePolicy({rolloutcase.POLICY_NAMES[k]: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)).cuda() for k, v in arrs.items()}, I)
    ident = np.arange(I, dtype=np.int64)
    ro = DeviceRollout(env, trk, pol, online=OnlineReward(um, ident, ident, feats, dur, (-60.0, 60.0), B))
    users = hash_ids(raw[:B], NB).to(torch.int32)
    ro.collect(users, seed=9)
    act = ro.traj.act.cpu().numpy().T; rew = ro.traj.rew.cpu().numpy().T
    a0 = act[:, 0]
    raw_s = um.forward(users.cpu().numpy().astype(np.int64), a0, feats[a0], dur[a0]).cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(rew[:, 0], (raw_s + 60.0) / 120.0, rtol=1e-12)
    # the pair scorer at E = 64 against the C oracle on the chosen pairs
    want_s = deepfmcase.oracle_forward(um_weights(um), users.cpu().numpy().astype(np.int64), a0, feats[a0], dur[a0])
    np.testing.assert_allclose(raw_s, want_s, rtol=2e-5, atol=5e-6)


def um_weights(um):
    from cirs_hip import abi
    return {f: um.t[f].cpu().numpy() for f in abi.DEEPFM_FIELDS}
