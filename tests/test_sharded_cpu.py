"""CPU, world_size 2 (gloo): the exchange layout of the row-sharded embedding lookup (cirs_hip.sharded.ShardedTable: all-to-all of
requested local row numbers, all-to-all of returned rows, fixed-capacity messages) returns exactly table[ids] on every rank -- for
uneven owner distributions, repeated ids, an id set owned by ONE rank (worst-case capacity) and a reduced capacity that still fits.
The local row gather is injected (torch indexing here; cirs_gather_rows on the GPU)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torch_gather(table, idx):
    out = table[idx.clamp(min=0)]
    out[idx < 0] = 0
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cirs_hip.sharded import DistComm, ShardedTable
        comm = DistComm()
        n_rows, R = 1001, 8
        full = torch.arange(n_rows * R, dtype=torch.float32).reshape(n_rows, R) * 0.5 + 1
        tab = ShardedTable(ShardedTable.shard_of(full, rank, world), n_rows, comm, gather=_torch_gather)
        rng = np.random.RandomState(7 + rank)
        cases = [torch.as_tensor(rng.randint(0, n_rows, 37)), torch.as_tensor(rng.randint(0, n_rows, 37) // world * world),   # all owned by rank 0
                 torch.as_tensor(np.repeat(rng.randint(0, n_rows, 5), 7)[:33]), torch.as_tensor(rng.randint(0, n_rows, 1))]
        ok = True
        for ids in cases:
            got = tab.lookup(ids)
            ok &= bool(torch.equal(got, full[ids]))
        ids = torch.as_tensor(np.arange(40) * 3 + rank)      # evenly spread over the owners: a capacity of 24 slots fits 40 ids
        ok &= bool(torch.equal(tab.lookup(ids, cap=24), full[ids]))
        # all-gather / all-to-all contracts of the comm itself
        g = comm.all_gather(torch.full((3,), float(rank)))
        ok &= bool(torch.equal(g, torch.arange(world, dtype=torch.float32)[:, None].expand(world, 3)))
        a = comm.all_to_all(torch.arange(world, dtype=torch.float32)[:, None] * 10 + rank)     # t[d] = 10 d + rank
        ok &= bool(torch.equal(a[:, 0], torch.arange(world, dtype=torch.float32) + 10 * rank))   # r[s] = 10 rank + s
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_sharded_table_lookup_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(0, True), (1, True)], res


class _LoopComm:
    """world 1: every collective is the identity (single process, no process group)."""
    world, rank = 1, 0

    def all_gather(self, t):
        return t.unsqueeze(0)

    def all_to_all(self, t):
        return t


def test_lookup_reduced_capacity_overflow_is_detected_not_out_of_range():
    """A capacity smaller than the largest per-owner count must not index out of range: overflowing requests are
    dropped (zero rows), counted on the device, and the first reduced-capacity call raises a clear error."""
    sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
    from cirs_hip.sharded import ShardedTable
    full = torch.arange(50 * 4, dtype=torch.float32).reshape(50, 4) + 1
    tab = ShardedTable(full.clone(), 50, _LoopComm(), gather=_torch_gather)
    ids = torch.arange(10)
    assert torch.equal(tab.lookup(ids, cap=10), full[ids])
    assert torch.equal(tab.lookup(ids, cap=12), full[ids])
    tab2 = ShardedTable(full.clone(), 50, _LoopComm(), gather=_torch_gather)
    with pytest.raises(RuntimeError, match="capacity"):
        tab2.lookup(ids, cap=6)
    # later calls do not sync; the rows that fitted are right, the dropped ones are zero rows, the counter keeps counting
    got = tab2.lookup(ids, cap=6)
    assert torch.equal(got[:6], full[:6]) and float(got[6:].abs().sum()) == 0.0
    assert int(tab2.overflow) == 8
