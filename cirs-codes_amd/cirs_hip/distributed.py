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


def all_gather_records(fields: Dict[str, torch.Tensor], T: int, B_local: int, S: int, D: int, group=None) -> Dict[str, torch.Tensor]:
    """The single collective of the data path: all-gather of the packed per-rank trajectory records."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    local = pack_records(fields)
    if world == 1:
        return unpack_records(local.unsqueeze(0), 1, T, B_local, S, D)
    gathered = torch.empty(world * local.numel(), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(gathered, local, group=group)
    return unpack_records(gathered.view(world, local.numel()), world, T, B_local, S, D)
