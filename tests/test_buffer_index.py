"""CPU: index semantics of the mirror's VectorReplayBuffer against the KNOWN ANSWERS of the reference's own test
(tianshou/test/base/test_buffer.py:397-488, test_replaybuffermanager): add() return values, sample_index(0), prev / next /
unfinished_index incl. ring wrap-around inside a sub-buffer — the expected arrays below are that test's literals."""
import numpy as np

from tianshou.data import Batch, VectorReplayBuffer


def test_replaybuffermanager_known_answers():
    buf = VectorReplayBuffer(20, 4)
    batch = Batch(obs=np.array([1, 2, 3]), act=np.array([1, 2, 3]), rew=np.array([1, 2, 3]), done=np.array([0, 0, 1]))
    ptr, ep_rew, ep_len, ep_idx = buf.add(batch, buffer_ids=[0, 1, 2])
    assert np.all(ep_len == [0, 0, 1]) and np.all(ep_rew == [0, 0, 3])
    assert np.all(ptr == [0, 5, 10]) and np.all(ep_idx == [0, 5, 10])
    batch, indice = buf.sample(0)
    assert np.allclose(indice, [0, 5, 10])
    assert np.allclose(buf.prev(indice), indice)
    assert np.allclose(buf.next(indice), indice)
    assert np.allclose(buf.unfinished_index(), [0, 5])
    buf.add(Batch(obs=np.array([4]), act=np.array([4]), rew=np.array([4]), done=np.array([1])), buffer_ids=[3])
    assert np.allclose(buf.unfinished_index(), [0, 5])
    batch, indice = buf.sample(0)
    assert np.allclose(indice, [0, 5, 10, 15])
    assert np.allclose(buf.prev(indice), indice)
    assert np.allclose(buf.next(indice), indice)
    data = np.array([0, 0, 0, 0])
    buf.add(Batch(obs=data, act=data, rew=data, done=data), buffer_ids=[0, 1, 2, 3])
    buf.add(Batch(obs=data, act=data, rew=data, done=1 - data), buffer_ids=[0, 1, 2, 3])
    assert len(buf) == 12
    buf.add(Batch(obs=data, act=data, rew=data, done=data), buffer_ids=[0, 1, 2, 3])
    buf.add(Batch(obs=data, act=data, rew=data, done=np.array([0, 1, 0, 1])), buffer_ids=[0, 1, 2, 3])
    assert len(buf) == 20
    indice = buf.sample_index(0)
    assert np.allclose(indice, np.arange(len(buf)))
    assert np.allclose(buf.done, [
        0, 0, 1, 0, 0,
        0, 0, 1, 0, 1,
        1, 0, 1, 0, 0,
        1, 0, 1, 0, 1,
    ])
    assert np.allclose(buf.prev(indice), [
        0, 0, 1, 3, 3,
        5, 5, 6, 8, 8,
        10, 11, 11, 13, 13,
        15, 16, 16, 18, 18,
    ])
    assert np.allclose(buf.next(indice), [
        1, 2, 2, 4, 4,
        6, 7, 7, 9, 9,
        10, 12, 12, 14, 14,
        15, 17, 17, 19, 19,
    ])
    assert np.allclose(buf.unfinished_index(), [4, 14])
    ptr, ep_rew, ep_len, ep_idx = buf.add(Batch(obs=np.array([1]), act=np.array([1]), rew=np.array([1]), done=np.array([1])), buffer_ids=[2])
    assert np.all(ep_len == [3]) and np.all(ep_rew == [1])
    assert np.all(ptr == [10]) and np.all(ep_idx == [13])
    assert np.allclose(buf.unfinished_index(), [4])
    indice = list(sorted(buf.sample_index(0)))
    assert np.allclose(indice, np.arange(len(buf)))
    assert np.allclose(buf.prev(indice), [
        0, 0, 1, 3, 3,
        5, 5, 6, 8, 8,
        14, 11, 11, 13, 13,
        15, 16, 16, 18, 18,
    ])
    assert np.allclose(buf.next(indice), [
        1, 2, 2, 4, 4,
        6, 7, 7, 9, 9,
        10, 12, 12, 14, 10,
        15, 17, 17, 19, 19,
    ])
    # corner case: list, int and -1
    assert buf.prev(-1) == buf.prev([buf.maxsize - 1])[0]
    assert buf.next(-1) == buf.next([buf.maxsize - 1])[0]
    assert buf.sample_index(-1).tolist() == []


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
