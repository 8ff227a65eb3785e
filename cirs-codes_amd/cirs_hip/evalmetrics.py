"""Coverage / feature-domination counts on device trajectories (csrc/evalmetrics.hip, cirs_eval_coverage).

Host-side counterpart of the buffer walks in reference evaluation.py:303-352 (Callback_Coverage_Count) and of the row
test in get_feat_dominate_dict (evaluation.py:36-44)."""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import abi


def dominated_values(sorted_items, top_rate: float) -> np.ndarray:
    """Leading feature values whose normalised cumulative share stays <= top_rate, at least one
    (evaluation.py:20-31).  sorted_items: [(value, share), ...] sorted by share, descending."""
    values = np.array([pair[1] for pair in sorted_items], dtype=np.float64)
    values = values / sum(values)
    cumsum = values.cumsum()
    ind = 0
    for v in cumsum:
        if v > top_rate:
            break
        ind += 1
    if ind == 0:
        ind += 1
    return np.array([pair[0] for pair in sorted_items])[:ind]


def item_flags(feat_matrix: np.ndarray, dom_values: np.ndarray) -> np.ndarray:
    """[n_items, n_feat_cols] int feature ids -> u8 flag: the item has one of the dominating values (evaluation.py:39-42)."""
    return np.isin(np.asarray(feat_matrix).astype(np.int64), np.asarray(dom_values).astype(np.int64)).any(axis=1).astype(np.uint8)


class CoverageCounter:
    """Reusable scratch for one catalogue size."""

    def __init__(self, n_items: int, device="cuda"):
        self.n_items = int(n_items)
        self.device = torch.device(device)
        self.bitmap = torch.zeros((self.n_items + 31) // 32, dtype=torch.int32, device=self.device)
        self.out = torch.zeros(3, dtype=torch.int64, device=self.device)
        self._lib = abi.lib()

    def count(self, act: torch.Tensor, item_flag: Optional[torch.Tensor] = None):
        """act: int64 tensor of any shape (-1 = no recommendation) -> (hit_item, n_acts, n_flagged) as Python ints."""
        act = act.to(self.device, torch.int64).contiguous()
        if item_flag is not None:
            item_flag = item_flag.to(self.device, torch.uint8).contiguous()
            assert item_flag.numel() == self.n_items
        stream = torch.cuda.current_stream(self.device).cuda_stream
        abi.check(self._lib.cirs_eval_coverage(act.data_ptr(), act.numel(), self.n_items, abi.ptr(item_flag), self.bitmap.data_ptr(),
                                               self.out.data_ptr(), stream), "cirs_eval_coverage")
        hit, n, fl = self.out.cpu().tolist()
        return int(hit), int(n), int(fl)
