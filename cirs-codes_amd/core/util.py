"""core/util.py pieces of the reference that sit next to the hot path.

compute_action_distance / compute_exposure / clip0 (reference core/util.py:21-54) are fused into csrc/env.hip.  Here:
the item-item distance table builder / loader (reference core/util.py:225-273), which the reference computes with an
O(I^2) Python double loop and caches as CSV; this build computes it on the device (cirs_dist_jaccard) and reads / writes
the same CSV layout (index and columns = original photo ids, values = 1 / Jaccard similarity, inf when disjoint)."""
import os

import numpy as np
import pandas as pd
import torch

from cirs_hip import abi
from cirs_hip.synthetic import pack_item_cats


def _device_distance(list_feat_sub) -> np.ndarray:
    cats = np.full((len(list_feat_sub), 4), -1, dtype=np.int32)
    for i, lst in enumerate(list_feat_sub):
        lst = sorted(set(int(x) for x in lst))
        assert len(lst) <= 4, "KuaiRec items carry at most 4 categories (feat0..feat3)"
        cats[i, :len(lst)] = lst
    packed = torch.as_tensor(np.ascontiguousarray(pack_item_cats(cats)).view(np.int32)).cuda()
    n = len(list_feat_sub)
    dist = torch.empty((n, n), dtype=torch.float64, device="cuda")
    abi.check(abi.lib().cirs_dist_jaccard(packed.data_ptr(), n, dist.data_ptr(), torch.cuda.current_stream().cuda_stream), "cirs_dist_jaccard")
    return dist.cpu().numpy()


def get_distance_mat(list_feat, sub_index_list, DATAPATH="environments/KuaishouRec/data"):
    """reference core/util.py:225-244: load `distance_mat_photo_small.csv` or build it (and cache it)."""
    if sub_index_list is None:
        return None
    path = os.path.join(DATAPATH, "distance_mat_photo_small.csv")
    if os.path.isfile(path):
        df_dist_small = pd.read_csv(path, index_col=0)
        df_dist_small.columns = df_dist_small.columns.astype(int)
        return df_dist_small
    sub = np.asarray(sub_index_list)
    dist = _device_distance([list_feat[int(i)] for i in sub])
    df_dist_small = pd.DataFrame(dist, index=sub, columns=sub)
    df_dist_small.to_csv(path)
    return df_dist_small


def get_similarity_mat(list_feat, DATAPATH="environments/KuaishouRec/data"):
    """reference core/util.py:246-273: the full I x I Jaccard similarity (1 / distance; 0 when disjoint)."""
    path = os.path.join(DATAPATH, "similarity_mat_photo.csv")
    if os.path.isfile(path):
        df_sim = pd.read_csv(path, index_col=0)
        df_sim.columns = df_sim.columns.astype(int)
        return df_sim.to_numpy()
    sim = 1.0 / _device_distance(list_feat)
    pd.DataFrame(sim).to_csv(path)
    return sim
