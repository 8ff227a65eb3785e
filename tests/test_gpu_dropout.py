"""GPU: production-mode dropout of the state tracker (the reference trains, tests and back-propagates with
nn.Dropout(0.1) live -- core/state_tracker.py:155-156,176; CIRS-RL-kuaishou.py:235-243 never calls eval()).  Counter-based masks
keyed (key, global env, position, layer, site, element):
  * the K/V-cached decode steps (tracker_step_kernel<NH, true>) == the torch restatement with the SAME masks injected;
  * cirs_tracker_backward (masks regenerated from the counters) == autograd through that restatement;
  * through the engine: key = (seed, collect tag), env ids offset per rank; collect + update are deterministic;
  * p = 0 is the dropout-free kernel instantiation (every other GPU test)."""
import numpy as np
import pytest
import torch

import nn_oracle
import rolloutcase
from test_gpu_tracker_bwd import device_tracker_trainable, replay_tracker, rows_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("U,I,B,T,nhead,p", [(50, 80, 9, 12, 4, 0.1), (40, 60, 70, 30, 4, 0.1), (30, 40, 6, 7, 2, 0.3), (20, 30, 5, 40, 8, 0.1)])
def test_forward_and_backward_with_masks_match_restatement(U, I, B, T, nhead, p):
    from cirs_hip.rollout import Trajectory
    rng = np.random.RandomState(B * T + nhead)
    tp = rolloutcase.tracker_param_dict(U, I, T, seed=3)
    lens = rng.randint(2, T + 1, size=B)
    users = rng.randint(0, U, B); acts = rng.randint(0, I, (B, T)); rews = rng.uniform(0, 1, (B, T))
    G = rng.randn(T + 1, B, 20).astype(np.float32)
    env_base, seed, tag = 1000, 99, 17
    d = dict(p=p, key=nn_oracle.dropout_key(seed, tag), envs=np.arange(B) + env_base)
    # device: incremental decode with dropout on
    trk, views = device_tracker_trainable(tp, U, I, B, T, nhead=nhead)
    trk.set_dropout(p)
    trk.set_dropout_key(seed, tag, env_base)
    states_dev = np.zeros((B, T + 1, 20), np.float32)
    trk.reset()
    states_dev[:, 0] = trk.init(torch.as_tensor(users)).cpu().numpy()
    for t in range(T):
        live = np.where(lens > t)[0]
        if len(live) == 0:
            break
        out = trk.step(torch.as_tensor(acts[live, t]), torch.as_tensor(rews[live, t]), env_ids=torch.as_tensor(live.astype(np.int32)).cuda())
        states_dev[live, t + 1] = out.cpu().numpy()
    # restatement with the same masks + autograd
    tpo = {k: v.clone() for k, v in tp.items()}
    for k, v in tpo.items():
        if k != "pos_encoder.pe":
            v.requires_grad_(True)
    states = nn_oracle.tracker_forward_all(tpo, nn_oracle.tracker_inputs(tpo, users, acts, rews), nhead, dropout=d)  # [B, T+1, S]
    sn = states.detach().numpy()
    for b in range(B):
        np.testing.assert_allclose(states_dev[b, :lens[b] + 1], sn[b, :lens[b] + 1], atol=3e-5, rtol=1e-4)
    plain = nn_oracle.tracker_forward_all(tp, nn_oracle.tracker_inputs(tp, users, acts, rews), nhead).detach().numpy()
    assert np.abs(plain - sn).max() > 1e-3, "dropout must change the states"
    up = torch.zeros_like(states)
    for b in range(B):
        up[b, :lens[b]] = torch.as_tensor(G[:lens[b], b])
    (states * up).sum().backward()
    traj = Trajectory(B, T, 20, "cuda")
    a = np.where(np.arange(T)[None, :] < lens[:, None], acts, -1)
    traj.act.copy_(torch.as_tensor(a.T.copy())); traj.rew.copy_(torch.as_tensor(rews.T.copy()))
    offsets, row_env, row_t = rows_of(lens)
    dd = lambda x: torch.as_tensor(x).cuda()  # noqa: E731
    trk.backward(torch.as_tensor(users), traj, dd(row_env), dd(row_t), dd(offsets), dd(lens.astype(np.int32)), int(lens.sum()), dd(G))
    for k, gv in trk.grad_views.items():
        want = tpo[k].grad.numpy()
        got = gv.cpu().numpy()
        if k.endswith("self_attn.in_proj_bias"):   # key-bias gradient: analytically 0 (softmax shift invariance survives the mask)
            want, got = np.delete(want, slice(32, 64)), np.delete(got, slice(32, 64))
        scale = np.abs(want).max() + 1e-12
        np.testing.assert_allclose(got / scale, want / scale, atol=3e-4, err_msg=k)


def test_engine_dropout_mode_keys_and_determinism():
    from cirs_hip.engine import CirsEngine
    from cirs_hip.env import DeviceEnvTables
    from cirs_hip.synthetic import make_tables
    U, I, B, T = 90, 200, 48, 12
    tab = make_tables(U, I, seed=0, build_dist=False)
    a_env = tab.alpha_u[tab.raw_uid, 0].astype(np.float64); b_env = tab.beta_i[tab.raw_pid, 0].astype(np.float64)

    def run(dropout):
        dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, alpha_env=a_env, beta_env=b_env)
        eng = CirsEngine(dt, B, max_turn=T, num_leave_compute=3, leave_threshold=1, tau=10.0, gamma_exposure=10.0, seed=11, dropout=dropout)
        users = torch.as_tensor(np.random.RandomState(1).randint(0, U, B))
        eng.collect(users)                       # collect #0: rng_base 0
        eng.update(batch_size=64, repeat=2)
        eng.collect(users)                       # collect #1: rng_base T -> different masks
        return eng, users

    eng, users = run(0.1)
    tr = eng.rollout.traj
    lens = eng.lengths.cpu().numpy()
    act = tr.act.cpu().numpy().T; rew = tr.rew.cpu().numpy().T
    tp = {k: v.detach().cpu() for k, v in eng.tracker.params.items()}
    d = dict(p=0.1, key=nn_oracle.dropout_key(11, 1 * T), envs=np.arange(B))
    with torch.no_grad():
        want = nn_oracle.tracker_states(tp, users.numpy(), np.maximum(act, 0), rew, dropout=d).numpy()
        want_off = nn_oracle.tracker_states(tp, users.numpy(), np.maximum(act, 0), rew).numpy()
    obs = tr.obs.cpu().numpy()
    for b in range(B):
        np.testing.assert_allclose(obs[:lens[b] + 1, b], want[b, :lens[b] + 1], atol=5e-5, rtol=1e-4)
    assert np.abs(want - want_off).max() > 1e-3
    losses, n = eng.update(batch_size=64, repeat=2)
    assert torch.isfinite(losses).all() and torch.isfinite(eng.tracker_flat).all()
    eng2, _ = run(0.1)
    eng2.update(batch_size=64, repeat=2)
    assert torch.equal(eng.policy_flat, eng2.policy_flat) and torch.equal(eng.tracker_flat, eng2.tracker_flat)
    eng0, _ = run(0.0)
    assert not torch.equal(eng0.rollout.traj.obs, eng.rollout.traj.obs)


def test_stepwise_protocol_draws_fresh_masks_per_episode():
    """the per-step build_state protocol (Collector.collect(random=True), external preprocess_fn users) must key
    the dropout masks per reset -- two consecutive episodes with the SAME users / actions / rewards must not reuse the same masks
    (the reference draws fresh nn.Dropout noise on every call); eval() stays deterministic."""
    from core.inputs import SparseFeatP
    from core.state_tracker import StateTrackerTransformer
    U, I, B, T = 30, 40, 6, 8
    st = StateTrackerTransformer([SparseFeatP("feat_user", U, embedding_dim=32)], [SparseFeatP("feat_item", I, embedding_dim=32)],
                                 [], 32, 20, B, dropout=0.1, nhead=4, device="cuda", MAX_TURN=T, init_std=0.5)
    rng = np.random.RandomState(0)
    users, acts, rews = rng.randint(0, U, B), rng.randint(0, I, (T, B)), rng.uniform(0, 1, (T, B))

    def episode():
        st.build_state(reset=True, dim_batch=B)
        out = [st.build_state(obs=users, env_id=np.arange(B))["obs"].cpu().numpy()]
        for t in range(T):
            out.append(st.build_state(obs_next=acts[t], rew=rews[t], env_id=np.arange(B))["obs_next"].cpu().numpy())
        return np.stack(out)

    st.train()
    a, b = episode(), episode()
    assert np.abs(a - b).max() > 1e-6, "two stepwise episodes reused the same dropout masks"
    st.eval()
    c, d = episode(), episode()
    assert np.array_equal(c, d)


@pytest.mark.parametrize("U,I,B,T", [(60, 150, 12, 7), (300, 3327, 64, 30)])
def test_exact_redraw_collect_from_one_call_equals_the_stepwise_collect(U, I, B, T, monkeypatch):
    """cirs_rollout_steps_redraw (the whole exact-redraw collect from one call: per vector step the batched prefix pass of build_state call t inside the
    fused rollout) == the stage-by-stage loop of round 4 (RedrawRollout.collect_stepwise): actions, rewards, done flags, log-probs, values, episode
    lengths and the tracker's input slots bit for bit, the states of the envs alive at a call bit for bit.
    (The fused rollout's logit store is off here: at these env counts it would hand the pick the mass kernel's bf16-pipe logits, while the stage-by-stage
    loop's stand-alone pick recomputes them as fp32 chains -- 1e-7 apart, tests/test_gpu_rollout.py::test_small_count_switches_keep_the_draws;
    this test is about the redraw plumbing and keeps its bit-for-bit bar.)"""
    monkeypatch.setenv("CIRS_ROLLOUT_ZSTORE", "0")
    from cirs_hip.engine import CirsEngine
    from cirs_hip.env import DeviceEnvTables
    from cirs_hip.synthetic import make_tables
    tab = make_tables(U, I, seed=0, build_dist=False)
    a_env = tab.alpha_u[tab.raw_uid, 0].astype(np.float64); b_env = tab.beta_i[tab.raw_pid, 0].astype(np.float64)
    users = torch.as_tensor(np.random.RandomState(1).randint(0, U, B))
    out = []
    for stepwise in (False, True):
        dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, alpha_env=a_env, beta_env=b_env)
        eng = CirsEngine(dt, B, max_turn=T, num_leave_compute=3 if T < 30 else 10, leave_threshold=1 if T < 30 else 4, tau=10.0, gamma_exposure=10.0,
                         seed=11, dropout=0.1, dropout_redraw=True)
        ro = eng.rollout
        lens = (ro.collect_stepwise(users.cuda(), seed=(11 << 8), rng_base=0) if stepwise else ro.collect(users.cuda(), seed=(11 << 8), rng_base=0))
        torch.cuda.synchronize()
        tr = ro.traj
        out.append(dict(lens=lens.clone(), act=tr.act.clone(), rew=tr.rew.clone(), done=tr.done.clone(), logp=tr.logp.clone(), value=tr.value.clone(),
                        ctr=tr.ctr.clone(), obs=tr.obs.clone(), x_hist=eng.tracker.x_hist.clone()))
    a, b = out
    assert torch.equal(a["lens"], b["lens"]) and int(a["lens"].min()) >= 1
    live = (a["act"] >= 0)                                   # [T, B]: env alive at step t
    assert torch.equal(a["act"], b["act"])
    for k in ("done", "rew", "logp", "value", "ctr"):           # (rows of finished envs are padding: the two loops fill them differently)
        assert torch.equal(a[k][live], b[k][live]), k
    assert torch.equal(a["obs"][:T][live], b["obs"][:T][live])
    lens = a["lens"].long()
    for e in range(B):                                       # the final state of every env (call len[e]) and its input slots
        assert torch.equal(a["obs"][lens[e], e], b["obs"][lens[e], e])
        assert torch.equal(a["x_hist"][e, :lens[e] + 1], b["x_hist"][e, :lens[e] + 1])


def test_exact_redraw_mode_matches_the_reference_procedure():
    """the reference redraws the masks of the WHOLE prefix at every build_state call (core/state_tracker.py:170-186,
    243-246).  The exact-redraw option (cirs_hip/redraw.py, CirsEngine(dropout_redraw=True)) must (a) produce, for every call t, the
    state of the restatement run with call t's masks over positions 0..t, (b) back-propagate d loss / d s_t through call t's graph only
    (sum over the calls == autograd through the per-call restatement), (c) differ from the sticky production mode for t >= 1 and
    coincide with it when p = 0."""
    from cirs_hip.engine import CirsEngine
    from cirs_hip.env import DeviceEnvTables
    from cirs_hip.redraw import call_tag, redraw_tracker_backward
    from cirs_hip.synthetic import make_tables
    U, I, B, T, p, seed = 60, 150, 12, 7, 0.1, 11
    tab = make_tables(U, I, seed=0, build_dist=False)
    a_env = tab.alpha_u[tab.raw_uid, 0].astype(np.float64); b_env = tab.beta_i[tab.raw_pid, 0].astype(np.float64)
    users = torch.as_tensor(np.random.RandomState(1).randint(0, U, B))

    def make(dropout, redraw):
        dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, alpha_env=a_env, beta_env=b_env)
        return CirsEngine(dt, B, max_turn=T, num_leave_compute=3, leave_threshold=1, tau=10.0, gamma_exposure=10.0, seed=seed, dropout=dropout,
                          dropout_redraw=redraw)

    eng = make(p, True)
    lens = eng.collect(users).cpu().numpy()
    tr = eng.rollout.traj
    act = tr.act.cpu().numpy().T; rew = tr.rew.cpu().numpy().T; obs = tr.obs.cpu().numpy()
    assert lens.min() >= 1 and (act[np.arange(T)[None, :] < lens[:, None]] >= 0).all()
    tp = {k: v.detach().cpu().clone() for k, v in eng.tracker.params.items()}
    acts0 = np.maximum(act, 0)
    # (a) call by call against the restatement
    sticky_like = None
    for t in range(int(lens.max()) + 1):
        d = dict(p=p, key=nn_oracle.dropout_key(seed, call_tag(0, 0)), envs=t * B + np.arange(B))      # call t's masks: pseudo-env ids t * B + e
        with torch.no_grad():
            want = nn_oracle.tracker_states(tp, users.numpy(), acts0, rew, dropout=d).numpy()
        live = lens >= t
        np.testing.assert_allclose(obs[t][live], want[live, t], atol=5e-5, rtol=1e-4, err_msg=f"call {t}")
        if t == 2:
            d1 = dict(p=p, key=nn_oracle.dropout_key(seed, call_tag(0, 0)), envs=1 * B + np.arange(B))
            with torch.no_grad():
                other = nn_oracle.tracker_states(tp, users.numpy(), acts0, rew, dropout=d1).numpy()
            assert np.abs(other[live, 2] - want[live, 2]).max() > 1e-3, "a call's masks must differ from the previous call's"
    # (b) backward: sum over calls
    G = np.random.RandomState(3).randn(T + 1, B, 20).astype(np.float32)
    tpo = {k: v.clone() for k, v in tp.items()}
    for k, v in tpo.items():
        if k != "pos_encoder.pe":
            v.requires_grad_(True)
    total = 0.0
    for t in range(int(lens.max())):
        d = dict(p=p, key=nn_oracle.dropout_key(seed, call_tag(0, 0)), envs=t * B + np.arange(B))      # call t's masks: pseudo-env ids t * B + e
        st = nn_oracle.tracker_forward_all(tpo, nn_oracle.tracker_inputs(tpo, users.numpy(), acts0, rew), 4, dropout=d)
        w = torch.zeros(B, 20)
        sel = torch.as_tensor(lens > t)
        w[sel] = torch.as_tensor(G[t])[sel]
        total = total + (st[:, t] * w).sum()
    total.backward()
    offsets, row_env, row_t = rows_of(lens)
    dd = lambda x: torch.as_tensor(x).cuda()  # noqa: E731
    redraw_tracker_backward(eng.rollout, dd(row_env), dd(row_t), dd(offsets), dd(lens.astype(np.int32)), int(lens.sum()), dd(G))
    for k, gv in eng.tracker.grad_views.items():
        want, got = tpo[k].grad.numpy(), gv.cpu().numpy()
        if k.endswith("self_attn.in_proj_bias"):
            want, got = np.delete(want, slice(32, 64)), np.delete(got, slice(32, 64))
        scale = np.abs(want).max() + 1e-12
        np.testing.assert_allclose(got / scale, want / scale, atol=5e-4, err_msg=k)
    # the engine's update runs through it
    losses, n = eng.update(batch_size=32, repeat=2)
    assert n == int(lens.sum()) and torch.isfinite(losses).all() and torch.isfinite(eng.tracker_flat).all()
    # (c) p = 0: the option is the ordinary rollout
    e0, e1 = make(0.0, True), make(0.0, False)
    l0, l1 = e0.collect(users), e1.collect(users)
    assert torch.equal(l0, l1) and torch.equal(e0.rollout.traj.act, e1.rollout.traj.act)
    live = (e1.rollout.traj.act >= 0)
    torch.testing.assert_close(e0.rollout.traj.obs[:-1][live], e1.rollout.traj.obs[:-1][live], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(e0.rollout.traj.rew[live], e1.rollout.traj.rew[live], rtol=1e-12, atol=0)


@pytest.mark.parametrize("W,B,I,U,T", [(2, 12, 300, 90, 10), (4, 8, 500, 90, 8)])
def test_exact_redraw_with_replicated_ranks(monkeypatch, W, B, I, U, T):
    """The exact-redraw procedure under world_size > 1: W virtual ranks (threads + the thread-synchronised collectives of
    test_gpu_engine_dp), learner "replicated" -- every rank rolls out its own envs with call t's masks keyed by the GLOBAL pseudo-env id
    t * B_total + rank * B + e, gathers all trajectories, and runs the identical update incl. the one batched backward over all calls of all
    envs.  Checks: a rank's states are the restatement's under exactly those ids; ranks stay bit-identical; the result equals one device
    holding all W * B envs and running the same procedure on the gathered buffer."""
    import threading
    import torch.distributed as dist
    from test_gpu_engine_dp import FakeCollectives
    from cirs_hip.engine import CirsEngine
    from cirs_hip.env import DeviceEnvTables
    from cirs_hip.redraw import call_tag
    from cirs_hip.synthetic import make_tables
    tab = make_tables(U, I, seed=0, build_dist=False)
    a_env = tab.alpha_u[tab.raw_uid, 0].astype(np.float64); b_env = tab.beta_i[tab.raw_pid, 0].astype(np.float64)
    dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, alpha_env=a_env, beta_env=b_env)
    fake = FakeCollectives(W)
    monkeypatch.setattr(dist, "all_reduce", fake.all_reduce)
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake.all_gather_into_tensor)
    monkeypatch.setattr(dist, "reduce_scatter_tensor", fake.reduce_scatter_tensor)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: W)
    p, seed, bs = 0.1, 5, 32
    kw = dict(max_turn=T, num_leave_compute=3, leave_threshold=1, tau=10.0, gamma_exposure=10.0, seed=seed, batch_size_hint=bs, dropout=p, dropout_redraw=True)
    engines = [CirsEngine(dt, B, world_size=W, rank=r, learner_mode="replicated", tracker_backward="replicated", **kw) for r in range(W)]
    rng = np.random.RandomState(2)
    users = [torch.as_tensor(rng.randint(0, U, B)) for _ in range(W)]
    for r, eng in enumerate(engines):
        eng.collect(users[r])
    torch.cuda.synchronize()
    # a rank's states against the restatement with the global pseudo-env ids
    r = W - 1
    tr = engines[r].rollout.traj
    lens = engines[r].lengths.cpu().numpy()
    act = np.maximum(tr.act.cpu().numpy().T, 0); rew = tr.rew.cpu().numpy().T; obs = tr.obs.cpu().numpy()
    tp = {k: v.detach().cpu().clone() for k, v in engines[r].tracker.params.items()}
    for t in range(int(lens.max()) + 1):
        d = dict(p=p, key=nn_oracle.dropout_key(seed, call_tag(0, 0)), envs=t * (W * B) + r * B + np.arange(B))
        with torch.no_grad():
            want = nn_oracle.tracker_states(tp, users[r].numpy(), act, rew, dropout=d).numpy()
        live = lens >= t
        np.testing.assert_allclose(obs[t][live], want[live, t], atol=5e-5, rtol=1e-4, err_msg=f"rank {r} call {t}")
    n_total = int(sum(int(e.lengths.sum()) for e in engines))
    perms = [rng.permutation(n_total) for _ in range(2)]
    gathered, results = {}, [None] * W

    def run(q):
        try:
            fake.local.rank = q
            g = engines[q]._gather()
            if q == 0:
                gathered["traj"] = {k: getattr(g[0], k).clone() for k in ("obs", "act", "rew", "done", "logp", "value", "ctr")}
                gathered["x_hist"], gathered["lens"], gathered["users"] = g[1].clone(), g[2].clone(), g[3].clone()
            results[q] = engines[q].update(bs, 2, perms=perms)
        except Exception as exc:  # noqa: BLE001
            fake.errors.append(exc)
            fake.bar.abort()

    threads = [threading.Thread(target=run, args=(q,)) for q in range(W)]
    [t.start() for t in threads]; [t.join(timeout=120) for t in threads]
    assert not fake.errors, fake.errors
    for q in range(1, W):
        assert torch.equal(engines[0].policy_flat, engines[q].policy_flat) and torch.equal(engines[0].tracker_flat, engines[q].tracker_flat)
    monkeypatch.undo()
    ref = CirsEngine(dt, B * W, world_size=1, rank=0, **kw)
    for k, v in gathered["traj"].items():
        getattr(ref.rollout.traj, k).copy_(v)
    ref.tracker.x_hist.copy_(gathered["x_hist"])
    ref.lengths, ref.users = gathered["lens"].to(torch.int32), gathered["users"].to(torch.int32)
    ref.rollout._users, ref.rollout._key = ref.users, engines[0].rollout._key
    ref.update(bs, 2, perms=perms)
    assert torch.equal(ref.policy_flat, engines[0].policy_flat)
    assert torch.equal(ref.tracker_flat, engines[0].tracker_flat)
    assert not torch.equal(ref.tracker_flat, CirsEngine(dt, B * W, world_size=1, rank=0, **kw).tracker_flat), "the update must have moved the tracker"


@pytest.mark.parametrize("B,T,nhead,p", [(70, 30, 4, 0.1), (9, 31, 4, 0.0), (11, 12, 2, 0.3), (7, 20, 8, 0.1), (5, 9, 1, 0.2), (6, 40, 4, 0.1)])
def test_prefix_states_from_one_launch_equal_the_multi_launch_pass(B, T, nhead, p, monkeypatch):
    """cirs_tracker_prefix_states runs as ONE launch (prefix_env_kernel: one wavefront per env, prefixes of at most 32 rows: max_len = T + 1 = 32 is the largest; T = 40 takes the
    multi-launch pass in both runs) that keeps the arithmetic of the stand-alone kernels: its states are BIT-identical to the multi-launch pass
    (CIRS_TRACKER_PREFIX_LAUNCHES=1), and equal to the restatement under the same masks."""
    U, I = 40, 60
    rng = np.random.RandomState(B * T + nhead)
    tp = rolloutcase.tracker_param_dict(U, I, T, seed=3)
    lens = rng.randint(0, T + 2, size=B)          # rows per env (at most max_len = T + 1), some envs without rows
    lens[0] = T + 1
    users = rng.randint(0, U, B); acts = rng.randint(0, I, (B, T)); rews = rng.uniform(0, 1, (B, T))
    env_base, seed, tag = 500, 7, 3
    trk, _ = device_tracker_trainable(tp, U, I, B, T, nhead=nhead)
    trk.reset()
    trk.init(torch.as_tensor(users))
    for t in range(T):          # the input slots 0 .. T of every env (slot 0 = the user's, slot t + 1 = (action, reward) of step t)
        trk.step(torch.as_tensor(acts[:, t]), torch.as_tensor(rews[:, t]))
    trk.set_dropout(p)
    trk.set_dropout_key(seed, tag, env_base)
    offsets, row_env, row_t = rows_of(lens)
    dd = lambda x: torch.as_tensor(x).cuda()  # noqa: E731
    outs = []
    for launches in ("0", "1"):
        if launches == "1":
            monkeypatch.setenv("CIRS_TRACKER_PREFIX_LAUNCHES", "1")
        else:
            monkeypatch.delenv("CIRS_TRACKER_PREFIX_LAUNCHES", raising=False)
        out = torch.full((B, 20), 7.0, device="cuda")
        trk.prefix_states(dd(row_env), dd(row_t), dd(offsets), dd(lens.astype(np.int32)), int(lens.sum()), out)
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert (outs[0][lens == 0] == 7.0).all(), "envs without rows must be left untouched"
    d = dict(p=p, key=nn_oracle.dropout_key(seed, tag), envs=np.arange(B) + env_base) if p > 0 else None
    with torch.no_grad():
        want = nn_oracle.tracker_forward_all(tp, nn_oracle.tracker_inputs(tp, users, acts, rews), nhead, dropout=d).numpy()   # [B, T + 1, S]
    for b in range(B):
        if lens[b] > 0:
            np.testing.assert_allclose(outs[0][b], want[b, lens[b] - 1], atol=3e-5, rtol=1e-4)


@pytest.mark.parametrize("B,T,nhead,p,nlayers", [(70, 30, 4, 0.1, 2), (9, 12, 4, 0.0, 2), (11, 12, 2, 0.3, 2), (7, 20, 8, 0.1, 2), (5, 9, 1, 0.2, 2),
                                                 (6, 40, 4, 0.1, 2), (13, 10, 4, 0.1, 1), (8, 14, 4, 0.1, 3)])
def test_last_row_pass_equals_the_general_pass(B, T, nhead, p, nlayers, monkeypatch):
    """cirs_tracker_backward_last (the backward of a build_state call: the upstream gradient sits on every env's LAST row, core/state_tracker.py:243-246)
    runs its top layer on one row per env -- one-query attention, compact row chain and weight-gradient problems.  It must produce the gradients of
    cirs_tracker_backward fed a dstate that is zero except on those rows (only the summation order over rows differs), including envs without rows,
    and its fallbacks (one layer; CIRS_TRACKER_LAST_FULL; max_len > 64) are that general pass exactly."""
    from cirs_hip.rollout import Trajectory
    U, I = 40, 60
    rng = np.random.RandomState(B * T + nhead)
    tp = rolloutcase.tracker_param_dict(U, I, T, seed=5, nlayers=nlayers)
    lens = rng.randint(1, T + 1, size=B)
    lens[rng.randint(0, B)] = 0            # an env without rows (a pseudo-env of a call its env did not live to see)
    lens[rng.randint(0, B)] = T
    users = rng.randint(0, U, B); acts = rng.randint(0, I, (B, T)); rews = rng.uniform(0, 1, (B, T))
    g_last = rng.randn(B, 20).astype(np.float32)
    G = np.zeros((T + 1, B, 20), np.float32)
    for b in range(B):
        if lens[b] > 0:
            G[lens[b] - 1, b] = g_last[b]
    from cirs_hip.tracker import DeviceTracker, flat_tracker_params, tracker_param_shapes
    flat, views = flat_tracker_params(tracker_param_shapes(U, I, nlayers=nlayers), init=tp)
    trk = DeviceTracker({**views, "pos_encoder.pe": tp["pos_encoder.pe"].float().cuda().contiguous()}, U, I, B, T, nhead=nhead, nlayers=nlayers)
    trk.enable_training(flat)
    if p > 0:
        trk.set_dropout(p)
        trk.set_dropout_key(99, 17, 1000)
    trk.reset()
    trk.init(torch.as_tensor(users))
    for t in range(T):
        live = np.where(lens > t + 1)[0]          # slots 0 .. lens - 1 are what the rows read
        if len(live):
            trk.step(torch.as_tensor(acts[live, t]), torch.as_tensor(rews[live, t]), env_ids=torch.as_tensor(live.astype(np.int32)).cuda())
    traj = Trajectory(B, T, 20, "cuda")
    a = np.where(np.arange(T)[None, :] < lens[:, None], acts, -1)
    traj.act.copy_(torch.as_tensor(a.T.copy())); traj.rew.copy_(torch.as_tensor(rews.T.copy()))
    offsets, row_env, row_t = rows_of(lens)
    dd = lambda x: torch.as_tensor(x).cuda()  # noqa: E731
    args = (torch.as_tensor(users), traj, dd(row_env), dd(row_t), dd(offsets), dd(lens.astype(np.int32)), int(lens.sum()))
    trk.backward(*args, dd(G))
    want = {k: v.clone() for k, v in trk.grad_views.items()}
    trk.flat_grad.fill_(float("nan"))
    trk.backward(*args, dd(g_last), last_rows_only=True)
    for k, gv in trk.grad_views.items():
        scale = want[k].abs().max().item() + 1e-12
        err = ((gv - want[k]).abs().max().item()) / scale
        assert torch.isfinite(gv).all() and err < 2e-5, (k, err)
    monkeypatch.setenv("CIRS_TRACKER_LAST_FULL", "1")
    trk.flat_grad.fill_(float("nan"))
    trk.backward(*args, dd(g_last), last_rows_only=True)
    for k, gv in trk.grad_views.items():
        assert torch.equal(gv, want[k]), k
