"""GPU: the fused device-resident rollout, checked stage by stage against the oracle on the SAME inputs:
   actions  <- C oracle actor_sample on the rollout's own states (bit-exact, same counter RNG)
   rew/done <- C oracle env step teacher-forced with the rollout's actions (done bit-exact, rew 1e-12)
   states   <- torch-fp32 tracker restatement on the rollout's actions/rewards (1e-4)"""
import os

import numpy as np
import pytest
import torch

import envcase
import nn_oracle
from conftest import close
import policycase
import rolloutcase

pytestmark = pytest.mark.gpu


def check_rollout(U, I, B, T, seed, *, sync_every=None, **kw):
    from cirs_hip.synthetic import make_tables
    tab = make_tables(U, I, seed=0, build_dist=False)
    ro, tp, arrs, envp = rolloutcase.build_device_stack(tab, B, T, **kw)
    rng = np.random.RandomState(1)
    users = rng.randint(0, U, B)
    lengths = ro.collect(torch.as_tensor(users), seed=seed, rng_base=100, sync_every=sync_every).cpu().numpy()
    tr = ro.traj
    act = tr.act.cpu().numpy(); rew = tr.rew.cpu().numpy(); done = tr.done.cpu().numpy().astype(bool)
    obs = tr.obs.cpu().numpy(); logp = tr.logp.cpu().numpy(); value = tr.value.cpu().numpy()
    assert lengths.min() >= 1 and lengths.max() <= T
    # structure: act >= 0 exactly for t < length, done exactly at t == length-1
    tt = np.arange(T)[:, None]
    assert np.array_equal(act >= 0, tt < lengths[None, :])
    assert np.array_equal(done & (act >= 0), (tt == lengths[None, :] - 1))
    # --- env: teacher-forced oracle ---
    a_env, b_env = envp.pop("a_env"), envp.pop("b_env")
    cfg = envcase.env_cfg(U, I, dist_mode=1, **envp)
    host = envcase.HostEnv(cfg, tab.mat, tab.normed_mat, None, tab.item_cats, a_env, b_env, B)
    want = envcase.run_teacher_forced(host, users, np.maximum(act.T, 0), T)
    assert np.array_equal(want["length"], lengths)
    m = (act >= 0).T
    np.testing.assert_allclose(rew.T[m], want["rew"][m], rtol=1e-12)
    assert np.array_equal(done.T[m], want["done"][m])
    np.testing.assert_allclose(tr.ctr.cpu().numpy().T[m], want["ctr"][m], rtol=1e-12)
    # --- policy: oracle on the device's own states, step by step ---
    #     (SURVEY 8(c)'s protocol: ids identical wherever the draw's top-2 margin exceeds 1e-6; violations reported, expected and observed 0)
    draws = mism = 0
    for t in range(int(lengths.max())):
        live = act[t] >= 0
        oa, ol, ov, mg = policycase.oracle_sample(arrs, obs[t], seed=seed, rng_step=100 + t, skip=(~live).astype(np.uint8), want_margins=True)
        d, m = policycase.assert_draws_match(act[t][live], oa[live], mg[live], f"rollout step {t}")
        draws += d; mism += m
        assert np.array_equal(ov[live], value[t][live])
        same = live & (oa == act[t])
        close(logp[t][same], ol[same], 1e-4, 1e-4, "rollout: log-prob of the sampled action vs oracle")
    print(f"rollout U={U} I={I} B={B}: {draws} draws, {mism} ids differ inside the 1e-6 margin, 0 violations")
    # --- tracker: restatement over the recorded episodes ---
    states = nn_oracle.tracker_states(tp, users, np.maximum(act.T, 0), rew.T).numpy()  # [B, T+1, S]
    for b in range(B):
        L = lengths[b]
        close(obs[:L + 1, b], states[b, :L + 1], 1e-4, 1e-4, "rollout: tracker states vs restatement")
    # --- the fused step reads its weights from the packed image ([k/4][O][4], coalesced), the stand-alone tracker step from the
    #     row-major matrices: same fma order, so a teacher-forced replay through cirs_tracker_init / cirs_tracker_step must
    #     reproduce the states BIT FOR BIT
    from cirs_hip.tracker import DeviceTracker
    trk2 = DeviceTracker({k: v.float().cuda().contiguous() for k, v in tp.items()}, U, I, B, T)
    replay = np.zeros_like(obs)
    replay[0] = trk2.init(torch.as_tensor(users)).cpu().numpy()
    for t in range(int(lengths.max())):
        live = np.where(lengths > t)[0]
        out = trk2.step(torch.as_tensor(act[t, live]), torch.as_tensor(rew[t, live]), env_ids=torch.as_tensor(live.astype(np.int32)).cuda())
        replay[t + 1, live] = out.cpu().numpy()
    for b in range(B):
        assert np.array_equal(obs[:lengths[b] + 1, b], replay[:lengths[b] + 1, b]), f"env {b}: fused step (packed weight image) != stand-alone tracker step"
    return lengths


def test_rollout_small():
    lengths = check_rollout(60, 150, 24, 12, seed=5, N=3, thr=1)
    assert lengths.min() < lengths.max()


def test_rollout_ragged_env_counts():
    """env counts that are not a multiple of the four envs of a step workgroup, long episodes (positions beyond the 32 the attention keeps in registers)"""
    check_rollout(60, 150, 7, 12, seed=6, N=3, thr=1)
    check_rollout(40, 90, 5, 40, seed=8, N=2, thr=8)


def test_rollout_c2_shapes():
    check_rollout(1411, 3327, 64, 30, seed=7)


def test_rollout_c3_shapes():
    """The benchmarked workload (BASELINE configs[2]: 7176 x 10728, 1024 envs, T = 30): the fused 2-launch path
    (TailFuse / TrunkFuse in tracker_step_kernel) against the oracle, stage by stage."""
    lengths = check_rollout(7176, 10728, 1024, 30, seed=11)
    assert lengths.min() < lengths.max()


def test_rollout_more_envs_than_one_wave_per_simd():
    """2309 envs (not a multiple of anything): two step-kernel workgroups per CU, seven chunks per mass-kernel workgroup, ragged last
    row block -- the launch geometries the 1024-env benchmark shape never takes."""
    check_rollout(7176, 10728, 2309, 8, seed=13)


def test_rollout_early_stop_polling_is_equivalent():
    a = check_rollout(200, 400, 40, 30, seed=9, N=4, thr=2, sync_every=4)
    b = check_rollout(200, 400, 40, 30, seed=9, N=4, thr=2)
    assert np.array_equal(a, b)


def test_rollout_remove_recommended_ids_and_force_length():
    from cirs_hip.synthetic import make_tables
    U, I, B, T = 60, 90, 16, 20
    tab = make_tables(U, I, seed=0, build_dist=False)
    ro, tp, arrs, envp = rolloutcase.build_device_stack(tab, B, T, N=2, thr=10, remove_recommended_ids=True, force_length=10)
    users = np.random.RandomState(0).randint(0, U, B)
    lengths = ro.collect(torch.as_tensor(users), seed=3).cpu().numpy()
    assert (lengths == 10).all()  # force_length overrides the env's own done (collector.py:253-258)
    act = ro.traj.act.cpu().numpy()
    for b in range(B):
        a = act[:10, b]
        assert len(set(a.tolist())) == 10, "an id was recommended twice although remove_recommended_ids is on"


def test_small_count_switches_keep_the_draws(monkeypatch):
    """The two small-env-count variants of the sampler (read per call since round 6): CIRS_ROLLOUT_MASS_SMALL=0 runs actor_mass_kernel
    instead of actor_mass_small_kernel -- the same logits, exponentials and summation order, so the whole trajectory is BIT-identical; CIRS_ROLLOUT_ZSTORE=0
    lets the step kernel's pick recompute the drawn chunk's 128 logits as fp32 fma chains instead of reading the mass kernel's (bf16-pipe) accumulators
    -- values 1e-7 apart, so ids agree except inside the 1e-6 top-2 margin (none here) and log-probs to round-off."""
    from cirs_hip.synthetic import make_tables
    U, I, B, T = 1411, 3327, 64, 30
    tab = make_tables(U, I, seed=0, build_dist=False)
    users = torch.as_tensor(np.random.RandomState(1).randint(0, U, B))
    runs = {}
    for name, zs, ms in (("default", "1", "1"), ("mass_generic", "1", "0"), ("no_store", "0", "1")):
        monkeypatch.setenv("CIRS_ROLLOUT_ZSTORE", zs); monkeypatch.setenv("CIRS_ROLLOUT_MASS_SMALL", ms)
        ro, _, _, _ = rolloutcase.build_device_stack(tab, B, T)
        ro.collect(users, seed=7, rng_base=100)
        tr = ro.traj
        runs[name] = (tr.act.clone(), tr.logp.clone(), tr.obs.clone(), tr.rew.clone())
    for a, b in zip(runs["default"], runs["mass_generic"]):
        assert torch.equal(a, b)
    assert torch.equal(runs["default"][0], runs["no_store"][0]), "ids differ between the logit store and the recomputed chunk (a 1e-6 margin case?)"
    np.testing.assert_allclose(runs["default"][1].cpu().numpy(), runs["no_store"][1].cpu().numpy(), rtol=0, atol=1e-5)      # (observed 3.1e-6 on log-probs of -8)


@pytest.mark.parametrize("U,I,B,T,kw", [(60, 150, 24, 12, dict(N=3, thr=1)), (1411, 3327, 64, 30, {}), (300, 2000, 200, 9, dict(N=2, thr=2, remove_recommended_ids=True))])
def test_collect_from_one_call_equals_reset_plus_steps(U, I, B, T, kw, monkeypatch):
    """cirs_rollout_collect (env reset, the tracker's first position from the packed weight image with the first trunk in its launch, all vector steps; no
    clears) leaves the trajectory, the tracker's slots and the env state of reset() + cirs_rollout_steps(0, T), bit for bit -- also on a trajectory buffer
    that still holds another collect's entries."""
    from cirs_hip.synthetic import make_tables
    tab = make_tables(U, I, seed=0, build_dist=False)
    users = torch.as_tensor(np.random.RandomState(2).randint(0, U, B))
    users2 = torch.as_tensor(np.random.RandomState(3).randint(0, U, B))
    got = {}
    for mode in ("one call", "stepwise"):
        if mode == "stepwise":
            monkeypatch.setenv("CIRS_ROLLOUT_STEPWISE_RESET", "1")
        else:
            monkeypatch.delenv("CIRS_ROLLOUT_STEPWISE_RESET", raising=False)
        ro, tp, arrs, envp = rolloutcase.build_device_stack(tab, B, T, **kw)
        ro.collect(users2, seed=9, rng_base=3)          # (a previous collect: its entries must not shine through)
        lens = ro.collect(users, seed=21, rng_base=50)
        tr = ro.traj
        got[mode] = dict(lens=lens.clone(), act=tr.act.clone(), rew=tr.rew.clone(), done=tr.done.clone(), logp=tr.logp.clone(), value=tr.value.clone(),
                         ctr=tr.ctr.clone(), obs=tr.obs.clone(), x_hist=ro.tracker.x_hist.clone(), tlen=ro.tracker.len.clone(), turn=ro.env.turn.clone(),
                         edone=ro.env.done.clone())
    a, b = got["one call"], got["stepwise"]
    for k in ("lens", "act", "rew", "done", "logp", "ctr", "tlen", "turn", "edone"):
        assert torch.equal(a[k], b[k]), k
    live = a["act"] >= 0
    assert torch.equal(a["value"][live], b["value"][live])
    lens = a["lens"].cpu().numpy()
    for e in range(B):
        assert torch.equal(a["obs"][:lens[e] + 1, e], b["obs"][:lens[e] + 1, e]), e
        assert torch.equal(a["x_hist"][e, :lens[e] + 1], b["x_hist"][e, :lens[e] + 1]), e
