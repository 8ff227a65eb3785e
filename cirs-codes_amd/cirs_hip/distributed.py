"""Multi-GPU glue: environments shard across ranks (one process per GPU), ONE all-gather of the packed trajectory
records per PPO epoch, replicated learner.  Works on any torch device/backend (RCCL on MI355X, gloo in
the CPU tests).  No collective is used on the rollout path: envs are independent units.

Exactness: RunningMeanStd needs the variance of ALL returns (a2c.py:101-104) and minibatches are drawn from a
permutation of the WHOLE buffer (ppo.py:180), so every rank must see every transition.  After the gather each rank
holds the same buffer; with the same permutation seed the replicated learners stay in lock-step.
"""
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist

# (name, trailing shape factory, dtype, rows: 'T+1' | 'T' | 'B' )
FIELDS = [("obs", "T1BS", torch.float32), ("act", "TB", torch.int64), ("rew", "TB", torch.float64),
          ("done", "TB", torch.uint8), ("logp", "TB", torch.float32), ("value", "TB", torch.float32),
          ("ctr", "TB", torch.float64), ("x_hist", "BLD", torch.float32), ("lens", "B", torch.int32),
          ("users", "B", torch.int32)]


def _shape(kind, T, B, S, D):
    return {"T1BS": (T + 1, B, S), "TB": (T, B), "BLD": (B, T + 1, D), "B": (B,)}[kind]


def pack_records(fields: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Concatenate every field's bytes into one contiguous uint8 record buffer (one message per rank)."""
    parts = [fields[name].contiguous().view(torch.uint8).reshape(-1) for name, _, _ in FIELDS]
    return torch.cat(parts)


def unpack_records(buf: torch.Tensor, world: int, T: int, B_local: int, S: int, D: int) -> Dict[str, torch.Tensor]:
    """buf: [world, bytes_per_rank] uint8 -> global tensors with the env axis = rank-major concatenation."""
    out = {}
    off = 0
    for name, kind, dtype in FIELDS:
        shp = _shape(kind, T, B_local, S, D)
        n = 1
        for v in shp:
            n *= v
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        chunk = buf[:, off:off + nbytes].contiguous().view(dtype).reshape((world,) + shp)
        off += nbytes
        if kind in ("TB", "T1BS"):  # [W, T, Bl, ...] -> [T, W*Bl, ...]
            perm = (1, 0, 2) + tuple(range(3, chunk.dim()))
            chunk = chunk.permute(perm).reshape((shp[0], world * B_local) + shp[2:])
        else:  # [W, Bl, ...] -> [W*Bl, ...]
            chunk = chunk.reshape((world * B_local,) + shp[1:])
        out[name] = chunk.contiguous()
    return out


def all_gather_records(fields: Dict[str, torch.Tensor], T: int, B_local: int, S: int, D: int, group=None, coll=None) -> Dict[str, torch.Tensor]:
    """The single collective of the data path: all-gather of the packed per-rank trajectory records (through `coll`, a
    Collectives, when given: same explicit stream ordering as the learner's collectives)."""
    world = getattr(coll, "world", None) or (dist.get_world_size(group) if dist.is_initialized() else 1)
    local = pack_records(fields)
    if world == 1:
        return unpack_records(local.unsqueeze(0), 1, T, B_local, S, D)
    gathered = torch.empty(world * local.numel(), dtype=torch.uint8, device=local.device)
    if coll is not None:
        coll.all_gather(gathered, local)
    else:
        dist.all_gather_into_tensor(gathered, local, group=group)
    return unpack_records(gathered.view(world, local.numel()), world, T, B_local, S, D)


class Collectives:
    """The learner's collectives over torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, gloo on CPU / shared-GPU tests).

    Stream ordering, made explicit: every libcirs_hip launch goes to torch's CURRENT stream of the device (the `_stream()` of the
    engines reads it at every call) and every collective here is issued with async_op=True followed by `work.wait()`, i.e.
      * c10d records an event on the current stream when the collective is enqueued and makes its communication stream wait for it
        (the gradients written by the preceding launches are complete before RCCL reads them), and
      * `work.wait()` makes the current stream wait for the collective's completion event -- without blocking the host --, so the next
        library launch on that stream sees the reduced data.
    Every call asserts that the stream current at the call is the one the first call saw (the one the engines launch on).
    `check=True` (CIRS_DIST_CHECK=1) additionally records an event on that stream after `work.wait()`, waits for it on the host and
    asserts that the collective's work handle reports completion by then: the second bullet, observed (a `wait()` that did not order
    the stream would let the event complete first).  The first bullet is c10d's own contract and is not observable from here; the
    two-process RCCL test covers it end to end (ranks stay bit-identical over several updates)."""

    def __init__(self, group=None, device=None, check=None):
        import os
        self.group, self.device = group, (torch.device(device) if device is not None else None)
        self.check = (os.environ.get("CIRS_DIST_CHECK", "0") == "1") if check is None else bool(check)
        self._bound = None
        self.calls = {"all_reduce": 0, "reduce_scatter": 0, "all_gather": 0}
        self.bytes = {"all_reduce": 0, "reduce_scatter": 0, "all_gather": 0}

    def _backend(self):
        try:
            return dist.get_backend(self.group)
        except Exception:   # noqa: BLE001  (patched / uninitialised in the virtual-rank tests)
            return "nccl"

    def _ordered(self, kind, t, issue):
        self.calls[kind] += 1
        self.bytes[kind] += t.numel() * t.element_size()
        cuda = t.is_cuda
        if cuda:
            cur = torch.cuda.current_stream(t.device)
            if self._bound is None:
                self._bound = cur.cuda_stream
            assert cur.cuda_stream == self._bound, "collective issued on another stream than the library launches it orders against"
        work = issue()
        if work is not None:
            work.wait()          # current stream waits for the collective; the host does not
        if cuda and self.check:
            after = torch.cuda.Event(); after.record(cur)
            after.synchronize()
            assert work is None or work.is_completed(), "stream order violated: the launch stream ran past a collective that is still pending"

    def all_reduce(self, t):
        self._ordered("all_reduce", t, lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def all_gather(self, out, inp):
        self._ordered("all_gather", out, lambda: dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True))

    def reduce_scatter(self, out, inp):
        if self._backend() == "gloo":     # gloo has no reduce_scatter: all-reduce + own slice (shared-GPU / CPU tests only)
            def issue():
                tmp = inp.clone()
                dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
                r = dist.get_rank(self.group)
                out.copy_(tmp[r * out.numel():(r + 1) * out.numel()])
                return None
            self._ordered("reduce_scatter", inp, issue)
            return
        self._ordered("reduce_scatter", inp, lambda: dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group, async_op=True))


class EmulatedPeers:
    """Collectives of ONE real rank whose W - 1 peers are copies of itself (tools/emulate_world.py): all_gather tiles the local message W
    times, all_reduce / reduce_scatter leave the local contribution in place.  Shapes, launch sequence and the per-rank kernel work of a
    W-rank job are the real ones; the wire time of the collectives is NOT in the measurement (calls and bytes are counted like
    Collectives does, so that it can be added from a link model)."""

    def __init__(self, world: int, rank: int = 0):
        self.world, self.rank = int(world), int(rank)
        self.calls = {"all_reduce": 0, "reduce_scatter": 0, "all_gather": 0}
        self.bytes = {"all_reduce": 0, "reduce_scatter": 0, "all_gather": 0}

    def _count(self, kind, t):
        self.calls[kind] += 1
        self.bytes[kind] += t.numel() * t.element_size()

    def all_reduce(self, t):
        self._count("all_reduce", t)

    def all_gather(self, out, inp):
        self._count("all_gather", out)
        lo, hi = out.data_ptr(), out.data_ptr() + out.numel() * out.element_size()
        if lo <= inp.data_ptr() < hi:
            return       # in-place gather of this rank's shard of `out` (parameter shards): the peers' shards keep what they hold
        out.view(self.world, -1).copy_(inp.reshape(1, -1).expand(self.world, -1))

    def reduce_scatter(self, out, inp):
        self._count("reduce_scatter", inp)
        n = out.numel()
        out.copy_(inp.reshape(-1)[self.rank * n:(self.rank + 1) * n])
