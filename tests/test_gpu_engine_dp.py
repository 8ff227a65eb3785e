"""GPU (single device): the engine's data-parallel update (`CirsEngine._update_dp`, the path the 2/4/8-GPU bench runs) with
W virtual ranks = W threads on one GPU and a thread-synchronised stand-in for the collectives it uses (all_gather_into_tensor of
the packed trajectory records; all_reduce of gradients for learner "dp"; reduce_scatter_tensor + two all-gathers per minibatch for
"dp_sharded").  Checks: ranks stay bit-identical, and the result equals a single-device update of the gathered buffer with the SAME
batch_size -- the global minibatch is the reference's batch_size (CIRS-RL-kuaishou.py:89) for every W (VERDICT r02 next #1b)."""
import threading

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


class FakeCollectives:
    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.local = threading.local()
        self.errors = []

    def rank(self):
        return self.local.rank

    def all_reduce(self, t, op=None, group=None, async_op=False):
        torch.cuda.synchronize()
        self.slots[self.rank()] = t
        self.bar.wait()
        total = self.slots[0].clone()
        for r in range(1, self.world):      # same order on every rank -> identical bits
            total += self.slots[r]
        self.bar.wait()
        t.copy_(total)
        torch.cuda.synchronize()
        self.bar.wait()

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        torch.cuda.synchronize()
        self.slots[self.rank()] = inp.clone()     # the input may alias this rank's slice of `out` (in-place all-gather)
        self.bar.wait()
        out.copy_(torch.cat([s.reshape(-1) for s in self.slots]).view_as(out))
        torch.cuda.synchronize()
        self.bar.wait()

    def reduce_scatter_tensor(self, out, inp, op=None, group=None, async_op=False):
        torch.cuda.synchronize()
        self.slots[self.rank()] = inp
        self.bar.wait()
        n = out.numel()
        r = self.rank()
        total = self.slots[0][r * n:(r + 1) * n].clone()
        for q in range(1, self.world):      # rank order: every element is reduced exactly once, by its owner
            total += self.slots[q][r * n:(r + 1) * n]
        self.bar.wait()
        out.copy_(total)
        torch.cuda.synchronize()
        self.bar.wait()


@pytest.mark.parametrize("mode", ["dp", "dp_sharded", "tp", "replicated"])
@pytest.mark.parametrize("W,B,I,U,T,bs", [(2, 24, 300, 90, 10, 32), (4, 16, 1000, 90, 10, 32), (8, 12, 500, 90, 10, 64),
                                          (2, 512, 10728, 7176, 30, 1024), (4, 256, 10728, 7176, 30, 1024), (8, 128, 10728, 7176, 30, 1024)])
def test_engine_dp_update_with_virtual_ranks(monkeypatch, W, B, I, U, T, bs, mode):
    _virtual_rank_update(monkeypatch, W, B, I, U, T, bs, mode)


@pytest.mark.parametrize("mode", ["replicated", "dp"])
def test_engine_c4_size_update_with_8_virtual_ranks(monkeypatch, mode):
    """BASELINE configs[3] at its own size: 8 ranks x 1024 envs on the 7176 x 10728 tables, one update of the gathered
    buffer (8192 episodes; a fresh policy plays ~12-turn episodes: ~100 k rows, ~190 optimiser steps of 1024 rows; 245 k rows / ~480 steps once
    episodes run to max_turn).  Ranks bit-identical; `replicated` = the single-device update of the
    gathered buffer; `dp` tracks it for as long as two fp32 evaluations of this update can (~19 free-running steps), then
    stays on the same trajectory statistically."""
    _virtual_rank_update(monkeypatch, 8, 1024, 10728, 7176, 30, 1024, mode, c4=True)


def _virtual_rank_update(monkeypatch, W, B, I, U, T, bs, mode, c4=False):
    from cirs_hip.engine import CirsEngine
    from cirs_hip.env import DeviceEnvTables
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
    kw = dict(max_turn=T, num_leave_compute=3 if T < 30 else 10, leave_threshold=1 if T < 30 else 4, tau=10.0, gamma_exposure=10.0, seed=5, batch_size_hint=bs)
    engines = [CirsEngine(dt, B, world_size=W, rank=r, learner_mode=mode, tracker_backward="sharded", **kw) for r in range(W)]
    rng = np.random.RandomState(2)
    users = [torch.as_tensor(rng.randint(0, U, B)) for _ in range(W)]
    for r, eng in enumerate(engines):
        eng.collect(users[r])
    torch.cuda.synchronize()
    init_policy = engines[0].policy_flat.clone()
    n_total = int(sum(int(e.lengths.sum()) for e in engines))
    perms = [rng.permutation(n_total) for _ in range(2)]
    gathered = {}
    results = [None] * W

    def run(r):
        try:
            fake.local.rank = r
            if r == 0:
                pass
            g = engines[r]._gather()   # (traj, x_hist, lens, users) of ALL ranks
            if r == 0:
                gathered["traj"] = {k: getattr(g[0], k).clone() for k in ("obs", "act", "rew", "done", "logp", "value", "ctr")}
                gathered["x_hist"], gathered["lens"], gathered["users"] = g[1].clone(), g[2].clone(), g[3].clone()
            results[r] = engines[r].update(bs, 2, perms=perms)
        except Exception as exc:  # noqa: BLE001
            fake.errors.append(exc)
            fake.bar.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600 if c4 else 120)
    assert not fake.errors, fake.errors
    assert all(r is not None for r in results)
    for r in range(1, W):                                           # every rank applied the identical update
        assert torch.equal(engines[0].policy_flat, engines[r].policy_flat)
        assert torch.equal(engines[0].tracker_flat, engines[r].tracker_flat)
        assert torch.equal(results[0][0], results[r][0])
    assert results[0][1] == n_total
    # the collectives the mode promises, per minibatch step (+ the trajectory all-gather: once by this test's explicit _gather(), once
    # inside update(); + the d obs and tracker-gradient all-reduces per update)
    n_mb = results[0][0].shape[0]
    c = engines[0].coll.calls
    if mode == "dp":
        assert c == {"all_reduce": n_mb + 2, "reduce_scatter": 0, "all_gather": 2}, c
    elif mode == "replicated":     # replicated policy learner (no per-minibatch collective) + sharded tracker BPTT: one gradient all-reduce
        assert c == {"all_reduce": 1, "reduce_scatter": 0, "all_gather": 2}, c
    elif mode == "tp":     # per minibatch: statistics all-gather + d h2 all-reduce; per update: head shards all-gather + tracker gradients
        assert c == {"all_reduce": n_mb + 1, "reduce_scatter": 0, "all_gather": 2 + n_mb + 1}, c
    else:
        assert c == {"all_reduce": 2, "reduce_scatter": n_mb, "all_gather": 2 + 2 * n_mb}, c
    # single device: the gathered buffer, the SAME batch size
    monkeypatch.undo()
    ref = CirsEngine(dt, B * W, world_size=1, rank=0, **kw)
    for k, v in gathered["traj"].items():
        getattr(ref.rollout.traj, k).copy_(v)
    ref.tracker.x_hist.copy_(gathered["x_hist"])
    ref.lengths, ref.users = gathered["lens"].to(torch.int32), gathered["users"].to(torch.int32)
    ref_losses, ref_n = ref.update(bs, 2, perms=perms)
    assert ref_n == n_total
    if c4:
        assert n_total >= 8 * 8192 and n_mb >= 2 * (n_total // bs)      # the C4 buffer: 8192 episodes, >= 128 optimiser steps per update
        got_l, want_l = results[0][0].cpu().numpy(), ref_losses.cpu().numpy()
        got_p, want_p = engines[0].policy_flat.cpu().numpy(), ref.policy_flat.cpu().numpy()
        np.testing.assert_allclose(got_l[:16], want_l[:16], rtol=3e-4, atol=3e-5)       # free-running, while it means something
        if mode == "replicated":    # the same kernels on the same rows in the same order; only the tracker gradient is summed over rank shards
            np.testing.assert_allclose(got_l, want_l, rtol=3e-4, atol=3e-5)
            np.testing.assert_allclose(got_p, want_p, rtol=3e-4, atol=3e-6)
        else:                       # row-sharded sums: another fp32 evaluation of the same ~480 chained Adam steps
            assert np.isfinite(got_l).all() and abs(got_l[:, 0].mean() - want_l[:, 0].mean()) < 0.05 * abs(want_l[:, 0]).mean() + 1e-3
            p0 = init_policy.cpu().numpy()
            da, db = got_p - p0, want_p - p0
            assert float((da * db).sum() / np.sqrt((da * da).sum() * (db * db).sum())) > 0.95      # the two updates point the same way
        return
    np.testing.assert_allclose(results[0][0].cpu().numpy(), ref_losses.cpu().numpy(), rtol=3e-4, atol=3e-5)
    # (tp: the item-sharded learner takes the action's logit from a scalar fp32 chain on the owning shard, the single-device step from the
    #  bf16x6 accumulator of its statistics kernel -- 1e-7 apart, which Adam turns into a few 1e-6 on near-zero gradients)
    got_p, want_p = engines[0].policy_flat.cpu().numpy(), ref.policy_flat.cpu().numpy()
    if mode == "tp":     # a strict bar again (observed since the 1-ulp Adam arithmetic of round 5: max |error| 1.6e-6, nothing beyond the bar; gpurun_out/parity_margins.json)
        import conftest
        conftest.close(got_p, want_p, 3e-4, 8e-6, f"engine tp W={W}, {got_p.size} parameters after the update vs single device")
    else:
        np.testing.assert_allclose(got_p, want_p, rtol=3e-4, atol=3e-6)
    got_t, want_t = engines[0].tracker_flat.cpu().numpy(), ref.tracker_flat.cpu().numpy()
    # Adam turns tiny gradient differences into +-lr steps where the gradient is ~0 (cf. the key-bias note in DESIGN.md):
    # compare where the reference actually moved a parameter by a clear margin
    np.testing.assert_allclose(got_t, want_t, rtol=0, atol=2.5e-3)
    assert np.mean(np.abs(got_t - want_t) < 1e-5) > 0.97
