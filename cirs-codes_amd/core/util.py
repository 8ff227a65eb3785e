"""core/util.py pieces of the reference that sit next to the hot path.

compute_action_distance / compute_exposure / clip0 (reference core/util.py:21-54) are fused into csrc/env.hip for the rollout;
the Python names below are the host-side helpers of the same formulas for callers outside the rollout (notebooks, analysis
scripts) — nothing on the product path calls them.  Also here:
the item-item distance table builder / loader (reference core/util.py:225-273), which the reference computes with an
O(I^2) Python double loop and caches as CSV; this build computes it on the device (cirs_dist_jaccard) and reads / writes
the same CSV layout (index and columns = original photo ids, values = 1 / Jaccard similarity, inf when disjoint)."""
import os

import numpy as np
import pandas as pd
import torch

from cirs_hip import abi
from cirs_hip.synthetic import pack_item_cats


def compute_action_distance(action, actions_hist, env_name="VirtualTB-v0", realenv=None):
    """reference core/util.py:21-38.  KuaishouEnv: positional lookup dist[action, hist] in the item-item table (env-encoded
    ids); VirtualTB: Euclidean distance between the 27-d action and every earlier action."""
    if env_name == "VirtualTB-v0":
        diff = np.asarray(action) - np.asarray(actions_hist)
        return np.sqrt((diff * diff).sum(axis=-1))
    if env_name == "KuaishouEnv-v0":
        df_dist_small = realenv.df_dist_small
        table = df_dist_small.to_numpy() if hasattr(df_dist_small, "to_numpy") else np.asarray(df_dist_small)
        return table[int(action), np.asarray(actions_hist, dtype=int)]
    raise ValueError(env_name)


def compute_exposure(t_diff, dist, tau):
    """reference core/util.py:40-50: sum_k exp(-t_diff_k * dist_k / tau); tau <= 0 switches the exposure effect off."""
    if tau <= 0:
        return 0
    return float(np.exp(-np.asarray(t_diff, dtype=np.float64) * np.asarray(dist, dtype=np.float64) / tau).sum())


def clip0(x):
    """reference core/util.py:53-54 — np.amax over axis 0, NOT max(x, 0): the identity on scalars."""
    return np.amax(x, 0)


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


# ---- dataset assembly of the user-model training (reference core/util.py:135-169, 277-309) --------------------------------
def compute_exposure_effect_kuaishouRec(df_x, timestamp, list_feat, tau, MODEL_SAVE_PATH, DATAPATH):
    """Exposure effect of every logged interaction, [n, 1] float64: sum over the user's EARLIER interactions of
    exp(-dt * dist(item, earlier item) / tau) (compute_exposure_each_user), cached as
    <MODEL_SAVE_PATH>/../saved_exposure/exposure_pos_<tau>.csv like the reference.  One launch of cirs_exposure_history over the
    whole log (a user's rows are contiguous, as in big_matrix.csv); distances come from the category lists."""
    from cirs_hip.dataprep import exposure_history
    cache = os.path.join(MODEL_SAVE_PATH, "..", "saved_exposure", "exposure_pos_{:.1f}.csv".format(tau))
    if os.path.isfile(cache):
        return pd.read_csv(cache).to_numpy()
    expo = exposure_history(df_x["user_id"].to_numpy(), df_x["photo_id"].to_numpy(), np.asarray(timestamp, dtype=np.float64), float(tau),
                            list_feat=list_feat).cpu().numpy().reshape(-1, 1)
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    pd.DataFrame(expo).to_csv(cache, index=False)
    return expo


def negative_sampling(df_big, df_feat, DATAPATH):
    """One negative item per logged (user, item) pair: the nearest higher (else lower) photo id the user has interacted with in
    neither matrix (find_negative), with that item's features, mean duration and watch_ratio 0.  The two user x item
    interaction matrices are bitmaps; the search is cirs_find_negative."""
    import json
    from cirs_hip.dataprep import bitmap_rows, find_negative
    small = pd.read_csv(os.path.join(DATAPATH, "small_matrix.csv"), header=0, usecols=["user_id", "photo_id"])
    n_user, n_item = int(df_big["user_id"].max()) + 1, int(df_big["photo_id"].max()) + 1
    seen = []
    for log in (small, df_big):
        m = np.zeros((n_user, n_item), dtype=bool)
        m[log["user_id"].to_numpy(), log["photo_id"].to_numpy()] = True
        seen.append(bitmap_rows(m))
    users = df_big["user_id"].to_numpy()
    neg = find_negative(users, df_big["photo_id"].to_numpy(), seen[0], seen[1], n_item).cpu().numpy()
    out = pd.DataFrame({"user_id": users.astype(int), "photo_id": neg.astype(int)})
    out = out.merge(df_feat, on=["photo_id"], how="left")
    with open(os.path.join(DATAPATH, "photo_mean_duration.json")) as fh:
        mean_dur = {int(k): v for k, v in json.load(fh).items()}
    out["photo_duration"] = out["photo_id"].map(mean_dur)
    out["watch_ratio"] = 0.0
    return out
