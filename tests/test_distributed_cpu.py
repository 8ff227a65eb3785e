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
