"""CPU, world_size 2 over gloo: the single all-gather of packed trajectory records reconstructs the global buffer in
rank-major env order on every rank, and the minibatch schedule is identical across ranks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_fields(rank, T, B, S, D):
    g = torch.Generator().manual_seed(100 + rank)
    return dict(obs=torch.randn(T + 1, B, S, generator=g), act=torch.randint(-1, 50, (T, B), generator=g),
                rew=torch.rand(T, B, generator=g, dtype=torch.float64), done=torch.randint(0, 2, (T, B), generator=g).to(torch.uint8),
                logp=torch.randn(T, B, generator=g), value=torch.randn(T, B, generator=g),
                ctr=torch.rand(T, B, generator=g, dtype=torch.float64), x_hist=torch.randn(B, T + 1, D, generator=g),
                lens=torch.randint(1, T + 1, (B,), generator=g).to(torch.int32), users=torch.randint(0, 9, (B,), generator=g).to(torch.int32))


def worker(rank, world, port, T, B, S, D, ret):
    sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cirs_hip import distributed
    from cirs_hip.learner import minibatch_slices
    mine = make_fields(rank, T, B, S, D)
    g = distributed.all_gather_records(mine, T, B, S, D)
    ok = True
    for r in range(world):
        ref = make_fields(r, T, B, S, D)
        for name in ("obs", "act", "rew", "done", "logp", "value", "ctr"):
            ok &= torch.equal(g[name][:, r * B:(r + 1) * B], ref[name])
        for name in ("x_hist", "lens", "users"):
            ok &= torch.equal(g[name][r * B:(r + 1) * B], ref[name])
    n = int(g["lens"].sum())
    perm = np.random.RandomState(1234).permutation(n)
    sched = torch.tensor([hash((tuple(perm[:8].tolist()), tuple(minibatch_slices(n, 16)))) & 0x7FFFFFFF])
    other = [torch.zeros_like(sched) for _ in range(world)]
    dist.all_gather(other, sched)
    ok &= all(int(o) == int(sched) for o in other)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_all_gather_records_world2():
    world, T, B, S, D = 2, 5, 3, 20, 32
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(worker, args=(world, port, T, B, S, D, ret), nprocs=world, join=True)
    assert ret[0] and ret[1]


def test_pack_unpack_single_rank_roundtrip():
    sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
    from cirs_hip import distributed
    T, B, S, D = 4, 6, 20, 32
    f = make_fields(0, T, B, S, D)
    g = distributed.unpack_records(distributed.pack_records(f).unsqueeze(0), 1, T, B, S, D)
    for k, v in f.items():
        assert torch.equal(g[k], v), k


def test_emulated_peers_tile_the_local_message_and_count_traffic():
    """tools/emulate_world.py's stand-in for W - 1 peers: all-gather tiles the local message, reductions keep the local contribution,
    calls and bytes are counted like Collectives does; all_gather_records takes its world size from it."""
    import torch
    from cirs_hip import distributed
    from cirs_hip.distributed import EmulatedPeers
    W = 4
    coll = EmulatedPeers(W)
    inp = torch.arange(6, dtype=torch.float32)
    out = torch.zeros(W * 6)
    coll.all_gather(out, inp)
    assert torch.equal(out.view(W, 6), inp.expand(W, 6))
    t = torch.ones(5); coll.all_reduce(t)
    assert torch.equal(t, torch.ones(5))
    shard = torch.zeros(3); coll.reduce_scatter(shard, torch.arange(12, dtype=torch.float32))
    assert torch.equal(shard, torch.arange(3, dtype=torch.float32))
    assert coll.calls == {"all_reduce": 1, "reduce_scatter": 1, "all_gather": 1} and coll.bytes["all_gather"] == W * 6 * 4
    T, B, S, D = 3, 2, 20, 32
    f = dict(obs=torch.randn(T + 1, B, S), act=torch.randint(0, 9, (T, B)), rew=torch.rand(T, B, dtype=torch.float64),
             done=torch.zeros(T, B, dtype=torch.uint8), logp=torch.randn(T, B), value=torch.randn(T, B), ctr=torch.rand(T, B, dtype=torch.float64),
             x_hist=torch.randn(B, T + 1, D), lens=torch.tensor([3, 2], dtype=torch.int32), users=torch.tensor([5, 7], dtype=torch.int32))
    g = distributed.all_gather_records(f, T, B, S, D, coll=coll)
    assert g["obs"].shape == (T + 1, W * B, S) and torch.equal(g["obs"][:, B:2 * B], f["obs"]) and g["lens"].tolist() == [3, 2] * W
