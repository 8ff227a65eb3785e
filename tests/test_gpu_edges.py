"""GPU: edge cases of the hot path -- empty / ragged batches, tiny and non-multiple-of-32 sizes, max_turn = 100,
finished envs, single-minibatch and merged-minibatch updates."""
import ctypes as C

import numpy as np
import pytest
import torch

import envcase
import nn_oracle
import policycase
import rolloutcase

pytestmark = pytest.mark.gpu


def test_empty_batches_are_noops():
    from cirs_hip import abi
    from cirs_hip.env import DeviceEnv, DeviceEnvTables
    from cirs_hip.synthetic import make_tables
    tab = make_tables(20, 30, seed=0)
    dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, dist=tab.dist)
    env = DeviceEnv(dt, 4, num_leave_compute=2, leave_threshold=1, max_turn=5)
    env.reset(torch.arange(4))
    turn0 = env.turn.clone()
    o, r, d, c, _ = env.step(torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int32))
    assert o.numel() == 0 and torch.equal(env.turn, turn0)
    lib = abi.lib()
    assert lib.cirs_env_step(C.byref(env.cfg), C.byref(env._tab), C.byref(env._st), None, None, 0, None, None, None, None, None, None) == 0
    # a real null pointer with n > 0 is rejected on the host before any launch
    assert lib.cirs_env_step(C.byref(env.cfg), C.byref(env._tab), C.byref(env._st), None, None, 2, None, None, None, None, None, None) == -1


def test_finished_env_is_inert_and_out_of_range_action_is_rejected():
    from cirs_hip.env import DeviceEnv, DeviceEnvTables
    from cirs_hip.synthetic import make_tables
    tab = make_tables(20, 30, seed=0)
    dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, dist=tab.dist)
    env = DeviceEnv(dt, 2, num_leave_compute=2, leave_threshold=30, max_turn=3)
    env.reset(torch.tensor([1, 2]))
    for t in range(3):
        o, r, d, c, _ = env.step(torch.tensor([3, 4]), torch.tensor([0, 1]))
    assert d.cpu().numpy().all() and int(env.turn[0]) == 3          # t >= max_turn-1 -> done (kuaishouEnv.py:168-169)
    o, r, d, c, _ = env.step(torch.tensor([5, 6]), torch.tensor([0, 1]))
    assert int(env.turn[0]) == 3 and float(r.abs().sum()) == 0.0 and d.cpu().numpy().all()   # stepping a finished env: no-op
    env.reset(torch.tensor([1, 2]))
    o, r, d, c, _ = env.step(torch.tensor([-1, 30]), torch.tensor([0, 1]))   # invalid ids never touch state
    assert int(env.turn.sum()) == 0


@pytest.mark.parametrize("U,I,B,T,N,thr", [(9, 5, 1, 4, 1, 0), (40, 33, 3, 100, 10, 50), (64, 129, 70, 7, 3, 2), (30, 31, 33, 30, 5, 100)])
def test_rollout_and_update_odd_shapes(U, I, B, T, N, thr):
    """Tiny / ragged shapes: I < one MFMA tile, I = 4 tiles + 1, B = 1, B not a multiple of 32, T = 100 (tracker history
    of 101 positions).  Rollout is checked against the oracle stage-wise, then a full update must run and move weights."""
    from cirs_hip.engine import CirsEngine
    from cirs_hip.env import DeviceEnvTables
    from cirs_hip.synthetic import make_tables
    tab = make_tables(U, I, seed=3)
    a_env, b_env = envcase.ab_env_tables(tab.raw_uid, tab.raw_pid, tab.alpha_u, tab.beta_i, U, I)
    dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, dist=tab.dist, alpha_env=a_env, beta_env=b_env)
    tp = rolloutcase.tracker_param_dict(U, I, T, seed=4)
    eng = CirsEngine(dt, B, max_turn=T, num_leave_compute=N, leave_threshold=thr, tau=10.0, gamma_exposure=10.0, seed=11,
                     tracker_params=tp)
    users = torch.as_tensor(np.random.RandomState(0).randint(0, U, B))
    lens = eng.collect(users).cpu().numpy()
    tr = eng.rollout.traj
    act = tr.act.cpu().numpy(); rew = tr.rew.cpu().numpy(); obs = tr.obs.cpu().numpy()
    assert lens.min() >= 1 and lens.max() <= T
    names = rolloutcase.POLICY_NAMES
    arrs = {k: eng.policy_views[v].cpu().numpy() for k, v in names.items()}
    for t in range(int(lens.max())):
        live = act[t] >= 0
        oa, _, ov, mg = policycase.oracle_sample(arrs, obs[t], seed=(11 << 8), rng_step=t, skip=(~live).astype(np.uint8), want_margins=True)
        policycase.assert_draws_match(act[t][live], oa[live], mg[live], f"step {t}")
    host = envcase.HostEnv(envcase.env_cfg(U, I, num_leave_compute=N, leave_threshold=thr, max_turn=T, tau=10.0, gamma_exposure=10.0,
                                           version=1, r_decay=1.0, has_ab=True), tab.mat, tab.normed_mat, tab.dist, tab.item_cats, a_env, b_env, B)
    want = envcase.run_teacher_forced(host, users.numpy(), np.maximum(act.T, 0), T)
    assert np.array_equal(want["length"], lens)
    m = (act >= 0).T
    np.testing.assert_allclose(rew.T[m], want["rew"][m], rtol=1e-12)
    states = nn_oracle.tracker_states(tp, users.numpy(), np.maximum(act.T, 0), rew.T).numpy()
    for b in range(B):
        np.testing.assert_allclose(obs[:lens[b] + 1, b], states[b, :lens[b] + 1], atol=2e-4, rtol=2e-4)
    n = int(lens.sum())
    if n >= 2:
        before, tbefore = eng.policy_flat.clone(), eng.tracker_flat.clone()
        losses, nn_ = eng.update(batch_size=max(2, n // 3), repeat=2)
        assert nn_ == n and torch.isfinite(losses).all()
        assert float((eng.policy_flat - before).abs().max()) > 0 and float((eng.tracker_flat - tbefore).abs().max()) > 0
        assert torch.isfinite(eng.policy_flat).all() and torch.isfinite(eng.tracker_flat).all()


def test_minibatch_schedule_matches_batch_split_merge_last():
    """Batch.split(size, merge_last=True) (tianshou/data/batch.py:734-744): slices for awkward lengths."""
    from cirs_hip.learner import minibatch_slices
    assert minibatch_slices(55, 16) == [(0, 16), (16, 32), (32, 55)]
    assert minibatch_slices(64, 16) == [(0, 16), (16, 32), (32, 48), (48, 64)]
    assert minibatch_slices(10, 16) == [(0, 10)]
    assert minibatch_slices(17, 16) == [(0, 17)]
    assert minibatch_slices(33, 16) == [(0, 16), (16, 33)]


@pytest.mark.parametrize("n", [1, 2, 5, 1000, 28673, 300001])
def test_device_permutation_is_bit_exact_vs_oracle(n):
    """cirs_random_permutation (the minibatch shuffle) vs its C restatement, and it is a permutation."""
    import ctypes as C
    import oracle_lib
    from cirs_hip import abi
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    abi.check(abi.lib().cirs_random_permutation(n, 20230, 11, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "perm")
    got = out.cpu().numpy()
    want = np.empty(n, np.int32)
    assert oracle_lib.lib().oracle_random_permutation(n, 20230, 11, want.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(got, want)
    assert np.array_equal(np.sort(got), np.arange(n))


@pytest.mark.parametrize("n,count", [(1, 1), (1000, 2), (28673, 3), (501, 11)])
def test_permutations_of_an_update_from_one_launch(n, count):
    """cirs_random_permutations (all repeats of an update from one launch) = one cirs_random_permutation per tag, i.e. the oracle's, bit for bit."""
    import ctypes as C
    import oracle_lib
    from cirs_hip import abi
    out = torch.empty((count, n), dtype=torch.int32, device="cuda")
    abi.check(abi.lib().cirs_random_permutations(n, 777, 40, count, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "perms")
    got = out.cpu().numpy()
    for c in range(count):
        want = np.empty(n, np.int32)
        assert oracle_lib.lib().oracle_random_permutation(n, 777, 40 + c, want.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(got[c], want), c


def test_update_readback_writes_pinned_host_words():
    """cirs_ppo_update_readback: the lengths and the hand-off count land in pinned host memory from one launch (no copy)."""
    from cirs_hip import abi
    n = 1500
    lens = torch.randint(1, 31, (n,), dtype=torch.int32, device="cuda")
    host = torch.full((n,), -1, dtype=torch.int32).pin_memory()
    lost = torch.full((1,), -1, dtype=torch.int32).pin_memory()
    abi.check(abi.lib().cirs_ppo_update_readback(lens.data_ptr(), n, host.data_ptr(), lost.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream), "readback")
    torch.cuda.synchronize()
    assert torch.equal(host, lens.cpu()) and int(lost[0]) == 0
    # ... with process_fn's offsets / row count from the same launch
    host.fill_(-1)
    off = torch.full((n,), -1, dtype=torch.int32, device="cuda"); nrow = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    abi.check(abi.lib().cirs_ppo_update_readback(lens.data_ptr(), n, host.data_ptr(), lost.data_ptr(), off.data_ptr(), nrow.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream), "readback")
    torch.cuda.synchronize()
    lc = lens.cpu()
    assert torch.equal(host, lc) and int(nrow.cpu()) == int(lc.sum()) and torch.equal(off.cpu(), (torch.cumsum(lc, 0) - lc).to(torch.int32))
