"""Dataset preparation of the user-model training on the device (csrc/dataprep.hip).

Host-side counterpart of reference core/util.py:56-76,135-196 (compute_exposure_effect_kuaishouRec / compute_exposure_each_user,
negative_sampling / find_negative)."""
from typing import Optional

import numpy as np
import torch

from . import abi
from .synthetic import pack_item_cats


def exposure_history(user_id, photo_id, timestamp, tau: float, *, dist: Optional[np.ndarray] = None, list_feat=None, device="cuda"):
    """exposure_pos [n_rows] float64 of every logged interaction (rows in file order, a user's rows contiguous).
    Pass either the distance table `dist` (1 / similarity, [n_items, n_items]) or `list_feat` (category lists per item id)."""
    user_id = np.asarray(user_id); n = len(user_id)
    # first row of each row's user (df_user.index[0], util.py:158): rows of a user are contiguous in the log
    change = np.r_[True, user_id[1:] != user_id[:-1]]
    start = np.maximum.accumulate(np.where(change, np.arange(n), 0)).astype(np.int64)
    dev = torch.device(device)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a)).to(dev, dt).contiguous()
    start_d, photo_d, ts_d = t(start, torch.int64), t(np.asarray(photo_id), torch.int32), t(np.asarray(timestamp, np.float64), torch.float64)
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    dist_d = cats_d = None
    if dist is not None:
        dist_d = t(np.asarray(dist, np.float64), torch.float64)
        n_items = dist_d.shape[0]
    else:
        cats = np.full((len(list_feat), 4), -1, np.int32)
        for i, f in enumerate(list_feat):
            cats[i, :len(f)] = sorted(set(int(c) for c in f))
        cats_d = torch.as_tensor(np.ascontiguousarray(pack_item_cats(cats)).view(np.int32)).to(dev)
        n_items = len(list_feat)
    abi.check(abi.lib().cirs_exposure_history(start_d.data_ptr(), photo_d.data_ptr(), ts_d.data_ptr(), n, abi.ptr(dist_d), abi.ptr(cats_d),
                                              n_items, float(tau), out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
              "cirs_exposure_history")
    return out


def bitmap_rows(mat_bool: np.ndarray) -> np.ndarray:
    """[n_users, n_items] bool -> [n_users, ceil(n_items/32)] uint32 (bit i of word i>>5)"""
    n_u, n_i = mat_bool.shape
    padded = np.zeros((n_u, ((n_i + 31) // 32) * 32), dtype=bool)
    padded[:, :n_i] = mat_bool
    return np.packbits(padded, axis=1, bitorder="little").view(np.uint32)


def find_negative(user_ids, photo_ids, seen_small_bits, seen_big_bits, n_items: int, absent_id: int = 1225, device="cuda"):
    """negative item per (user, positive item) row; seen_*_bits: bitmap_rows() of the two interaction matrices."""
    dev = torch.device(device)
    u = torch.as_tensor(np.asarray(user_ids)).to(dev, torch.int64).contiguous()
    p = torch.as_tensor(np.asarray(photo_ids)).to(dev, torch.int64).contiguous()
    a = torch.as_tensor(np.ascontiguousarray(seen_small_bits).view(np.int32)).to(dev).contiguous()
    b = torch.as_tensor(np.ascontiguousarray(seen_big_bits).view(np.int32)).to(dev).contiguous()
    out = torch.empty_like(u)
    abi.check(abi.lib().cirs_find_negative(u.data_ptr(), p.data_ptr(), u.numel(), a.data_ptr(), b.data_ptr(), int(n_items), int(absent_id),
                                           out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "cirs_find_negative")
    return out
