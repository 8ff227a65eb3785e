"""CPU: index semantics of the mirror's VectorReplayBuffer (add() return values, sample_index(0), prev / next / unfinished_index incl. ring
wrap-around inside a sub-buffer) against states RECORDED from the reference's own class: tests/golden/buffer_index.json, written by
oracle/gen_golden.py:gen_bufferindex (op scripts replayed on the reference's VectorReplayBuffer, full index state dumped after every op; script 0 is
the scenario of the reference's known-answer test, tianshou/test/base/test_buffer.py:397-488, scripts 1-2 are seeded random add sequences)."""
import json
import os

import numpy as np
import pytest

from tianshou.data import Batch, VectorReplayBuffer

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "buffer_index.json")) as _f:
    _SCRIPTS = json.load(_f)["scripts"]


@pytest.mark.parametrize("k", range(len(_SCRIPTS)))
def test_index_state_after_every_add_equals_the_reference(k):
    sc = _SCRIPTS[k]
    buf = VectorReplayBuffer(sc["total"], sc["n"])
    for n_op, op in enumerate(sc["ops"]):
        v, want, at = np.array(op["val"]), op["want"], f"script {k}, op {n_op}"
        ptr, ep_rew, ep_len, ep_idx = buf.add(Batch(obs=v, act=v, rew=v, done=np.array(op["done"])), buffer_ids=op["ids"])
        assert np.asarray(ptr).tolist() == want["ptr"] and np.asarray(ep_len).tolist() == want["ep_len"] and np.asarray(ep_idx).tolist() == want["ep_idx"], at
        np.testing.assert_allclose(np.asarray(ep_rew, np.float64), want["ep_rew"], rtol=0, atol=0, err_msg=at)
        assert len(buf) == want["len"], at
        idx = np.sort(buf.sample_index(0))
        assert idx.tolist() == want["index"], at
        assert np.asarray(buf.prev(idx)).tolist() == want["prev"], at
        assert np.asarray(buf.next(idx)).tolist() == want["next"], at
        assert np.asarray(buf.unfinished_index()).tolist() == want["unfinished"], at
        assert np.asarray(buf.done).astype(int).tolist() == want["done"], at
        np.testing.assert_array_equal(np.asarray(buf.rew, np.float64), want["rew"], err_msg=at)
        # corner cases: a scalar index, -1, an empty sample
        assert int(buf.prev(-1)) == want["prev_last"] == int(buf.prev([buf.maxsize - 1])[0]), at
        assert int(buf.next(-1)) == want["next_last"] == int(buf.next([buf.maxsize - 1])[0]), at
        assert buf.sample_index(-1).tolist() == want["sample_minus1"], at


def test_fill_from_trajectory_equals_sequence_of_adds():
    """The one-shot fill from a (host stand-in of a) device trajectory leaves the same index state as the per-step adds of
    Collector.collect (core/collector.py:278): offsets, lengths, last_index, prev / next / unfinished_index, sample(0)."""
    import torch
    rng = np.random.RandomState(0)
    B, T, S = 6, 7, 3
    lens = np.array([7, 3, 1, 5, 7, 2])

    class Traj:
        pass
    tr = Traj()
    tr.obs = torch.as_tensor(rng.randn(T + 1, B, S).astype(np.float32))
    tr.act = torch.as_tensor(np.where(np.arange(T)[:, None] < lens[None, :], rng.randint(0, 50, (T, B)), -1))
    tr.rew = torch.as_tensor(rng.uniform(0, 1, (T, B)))
    tr.done = torch.as_tensor((np.arange(T)[:, None] == lens[None, :] - 1).astype(np.uint8))
    tr.ctr = torch.as_tensor(rng.uniform(0, 1, (T, B)))
    a = VectorReplayBuffer(B * T, B)
    a.fill_from_trajectory(tr, lens)
    b = VectorReplayBuffer(B * T, B)
    ep = []
    for t in range(T):
        live = np.where(lens > t)[0]
        ptr, ep_rew, ep_len, ep_idx = b.add(Batch(obs=tr.obs[t, live].numpy(), obs_next=tr.obs[t + 1, live].numpy(), act=tr.act[t, live].numpy(),
                                                  rew=tr.rew[t, live].numpy(), done=tr.done[t, live].numpy().astype(bool),
                                                  info=Batch(CTR=tr.ctr[t, live].numpy(), env_id=live)), buffer_ids=live)
        fin = tr.done[t, live].numpy().astype(bool)
        ep += [(int(e), float(r), int(l), int(i)) for e, r, l, i in zip(live[fin], ep_rew[fin], ep_len[fin], ep_idx[fin])]
    assert len(a) == len(b) == lens.sum()
    assert np.array_equal(a.last_index, b.last_index) and np.array_equal(a._lengths, b._lengths)
    idx = a.sample_index(0)
    assert np.array_equal(idx, b.sample_index(0))
    for k in ("act", "rew", "done"):
        assert np.array_equal(np.asarray(a.__getattr__(k))[idx], np.asarray(b.__getattr__(k))[idx]), k
    np.testing.assert_array_equal(a.obs[idx].numpy(), b.obs[idx]); np.testing.assert_array_equal(a.obs_next[idx].numpy(), b.obs_next[idx])
    assert np.array_equal(a.info.CTR[idx], b.info.CTR[idx]) and np.array_equal(a.info.env_id[idx], b.info.env_id[idx])
    assert np.array_equal(a.prev(idx), b.prev(idx)) and np.array_equal(a.next(idx), b.next(idx))
    assert np.array_equal(a.unfinished_index(), b.unfinished_index()) and len(a.unfinished_index()) == 0
    # episode accounting of add(): reward sums / lengths / start indices per env
    for e, r, l, i in ep:
        assert l == lens[e] and i == a._offset[e]
        np.testing.assert_allclose(r, tr.rew[:lens[e], e].sum().item(), rtol=1e-14)
