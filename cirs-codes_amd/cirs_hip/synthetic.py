"""Deterministic synthetic KuaiRec-shaped tables.

The real KuaiRec CSVs are not shipped with the reference (`.gitignore:8-10`), so every parity test and the
benchmark run on tables generated here.  Shapes/semantics mirror what `KuaishouEnv.load_mat`
(reference environments/KuaishouRec/env/kuaishouEnv.py:61-111) returns:

  mat          U x I float64 watch ratio, clipped to [0, 5]                       (kuaishouEnv.py:66)
  raw_uid      U   sorted raw (big-matrix) user ids  == lbe_user.classes_         (kuaishouEnv.py:71-72)
  raw_pid      I   sorted raw photo ids              == lbe_photo.classes_        (kuaishouEnv.py:68-69)
  list_feat    list indexed by RAW photo id -> 1..4 distinct categories in [0,30] (kuaishouEnv.py:88-90)
  duration     I   photo_mean_duration of the env items                           (kuaishouEnv.py:98-106)
  dist         I x I float64 = 1/Jaccard(categories), inf when disjoint            (core/util.py:225-273)
  normed_mat   U x I float64 in [0,1]                                              (kuaishouEnv.py:139-143)
  alpha_u/beta_i  (U_raw,1)/(I_raw,1) float32, indexed by RAW ids                  (CIRS-RL-kuaishou.py:159-165)
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

N_CATEGORIES = 31  # KuaiRec item_categories.json: feature_index in [0, 30]


@dataclass
class KuaiTables:
    n_users: int
    n_items: int
    mat: np.ndarray
    normed_mat: np.ndarray
    raw_uid: np.ndarray
    raw_pid: np.ndarray
    list_feat: List[List[int]]          # indexed by raw photo id
    item_cats: np.ndarray               # (I, 4) int32, -1 padded, env-encoded item order
    duration: np.ndarray                # (I,) float64
    alpha_u: Optional[np.ndarray]       # (U_raw, 1) float32
    beta_i: Optional[np.ndarray]        # (I_raw, 1) float32
    dist: Optional[np.ndarray] = None   # (I, I) float64, built lazily (I^2 * 8 bytes)
    meta: dict = field(default_factory=dict)

    def list_feat_small(self) -> List[List[int]]:
        """Categories per env-encoded item (reference kuaishouEnv.py:49)."""
        return [self.list_feat[int(r)] for r in self.raw_pid]


def pack_item_cats(item_cats: np.ndarray) -> np.ndarray:
    """(I,4) int (-1 = none) -> (I,) uint32, four u8 lanes, 0xFF = none.  This is the HBM layout the HIP
    exit-rule / Jaccard kernels read (one 4-byte load per item)."""
    c = np.asarray(item_cats, dtype=np.int64)
    assert c.ndim == 2 and c.shape[1] == 4
    assert c.max(initial=-1) < 255, "category ids must fit in a byte (0xFF is the pad marker)"
    b = np.where(c < 0, 255, c).astype(np.uint32)
    return (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16) | (b[:, 3] << 24)).astype(np.uint32)


def jaccard_distance(item_cats: np.ndarray, rows: Optional[np.ndarray] = None) -> np.ndarray:
    """1/Jaccard over category *sets* (reference core/util.py:261-268: len(set & set)/len(set | set), then 1.0/sim).

    Computed exactly as the reference does in float64: sim = inter/union, dist = 1.0/sim (inf when inter == 0)."""
    c = np.asarray(item_cats)
    n = c.shape[0]
    masks = np.zeros(n, dtype=np.uint64)
    for k in range(c.shape[1]):
        valid = c[:, k] >= 0
        masks[valid] |= (np.uint64(1) << c[valid, k].astype(np.uint64))
    sel = np.arange(n) if rows is None else np.asarray(rows)

    def popcount(x):
        x = x - ((x >> np.uint64(1)) & np.uint64(0x5555555555555555))
        x = (x & np.uint64(0x3333333333333333)) + ((x >> np.uint64(2)) & np.uint64(0x3333333333333333))
        x = (x + (x >> np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
        return ((x * np.uint64(0x0101010101010101)) >> np.uint64(56)).astype(np.float64)

    out = np.empty((len(sel), n), dtype=np.float64)
    chunk = max(1, (1 << 22) // max(n, 1))
    for s in range(0, len(sel), chunk):
        m = masks[sel[s:s + chunk], None]
        inter = popcount(m & masks[None, :])
        union = popcount(m | masks[None, :])
        with np.errstate(divide="ignore", invalid="ignore"):
            sim = inter / union
            out[s:s + chunk] = 1.0 / sim
    return out


def make_tables(n_users: int, n_items: int, seed: int = 0, *, with_ab: bool = True, build_dist: bool = True,
                raw_user_space: Optional[int] = None, raw_item_space: Optional[int] = None,
                zipf_a: float = 1.2, normed: str = "uniform") -> KuaiTables:
    """Generate tables.  `seed` follows SURVEY.md §8(d): seed 0 for tables."""
    rng = np.random.RandomState(seed)
    raw_user_space = raw_user_space or max(7176, int(np.ceil(1.3 * n_users)))
    raw_item_space = raw_item_space or max(10729, int(np.ceil(1.3 * n_items)))
    raw_uid = np.sort(rng.choice(raw_user_space, size=n_users, replace=False)).astype(np.int64)
    raw_pid = np.sort(rng.choice(raw_item_space, size=n_items, replace=False)).astype(np.int64)

    mat = rng.uniform(0.0, 5.0, size=(n_users, n_items))

    # categories for every raw photo id (list_feat is indexed by raw id in the reference)
    pop = 1.0 / np.arange(1, N_CATEGORIES + 1) ** zipf_a
    pop /= pop.sum()
    n_cat = rng.randint(1, 5, size=raw_item_space)
    # Gumbel top-k == sampling without replacement proportional to `pop`
    keys = np.log(pop)[None, :] + rng.gumbel(size=(raw_item_space, N_CATEGORIES))
    order = np.argsort(-keys, axis=1)[:, :4]
    cats_raw = np.where(np.arange(4)[None, :] < n_cat[:, None], order, -1).astype(np.int32)
    list_feat = [[int(c) for c in row if c >= 0] for row in cats_raw]
    item_cats = cats_raw[raw_pid]

    duration = rng.uniform(2.0, 60.0, size=n_items)
    if normed == "uniform":
        normed_mat = rng.uniform(0.0, 1.0, size=(n_users, n_items))
    else:
        normed_mat = np.zeros((n_users, n_items))
    if with_ab:
        alpha_u = rng.normal(1.0, 0.1, size=(raw_user_space, 1)).astype(np.float32)
        beta_i = rng.normal(1.0, 0.1, size=(raw_item_space, 1)).astype(np.float32)
    else:
        alpha_u = beta_i = None
    dist = jaccard_distance(item_cats) if build_dist else None
    return KuaiTables(n_users=n_users, n_items=n_items, mat=mat, normed_mat=normed_mat, raw_uid=raw_uid,
                      raw_pid=raw_pid, list_feat=list_feat, item_cats=item_cats, duration=duration,
                      alpha_u=alpha_u, beta_i=beta_i, dist=dist,
                      meta=dict(seed=seed, raw_user_space=raw_user_space, raw_item_space=raw_item_space))


def write_kuairec_workspace(datapath: str, *, n_users: int = 48, n_items: int = 1400, n_env_users: int = 24, n_env_items: int = 160,
                            log_len=(20, 60), seed: int = 0) -> dict:
    """Write synthetic files in the KuaiRec on-disk layout under `datapath` -- what the reference expects in
    environments/KuaishouRec/data/ and does not ship (.gitignore): `big_matrix.csv` (the training log: user_id, photo_id,
    timestamp, watch_ratio, photo_duration [ms]; a user's rows contiguous and time-ordered), `small_matrix.csv` (the fully
    observed user x item block the env is built from), `item_categories.json`, `photo_mean_duration.json`.  Raw ids run to
    n_users / n_items (> 1225: the absent photo id of the reference's negative search, core/util.py:173-196); the env block is a
    subset of both.  Returns the generated arrays (for tests)."""
    import json
    import os

    import pandas as pd
    assert n_items > 1300 and n_env_users <= n_users and n_env_items <= n_items
    rng = np.random.RandomState(seed)
    os.makedirs(datapath, exist_ok=True)
    pop = 1.0 / np.arange(1, N_CATEGORIES + 1) ** 1.2
    pop /= pop.sum()
    list_feat = [sorted(rng.choice(N_CATEGORIES, size=rng.randint(1, 5), replace=False, p=pop).tolist()) for _ in range(n_items)]
    durations = rng.uniform(2.0, 60.0, n_items)
    env_users = np.sort(rng.choice(n_users, n_env_users, replace=False))
    env_items = np.sort(rng.choice(n_items, n_env_items, replace=False))
    hot = np.unique(np.r_[env_items, np.arange(1215, 1235), rng.choice(n_items, 3 * n_env_items, replace=False), [n_items - 1]])
    affinity = rng.gamma(2.0, 0.6, size=(n_users, N_CATEGORIES))      # users like categories: the watch ratio is learnable
    rows, t0 = [], 1.6e9
    for u in range(n_users):
        L = int(rng.randint(log_len[0], log_len[1]))
        ts = np.sort(t0 + rng.randint(0, 60000, L).astype(np.float64))
        items = rng.choice(hot, L)
        for k in range(L):
            ratio = float(np.mean(affinity[u, list_feat[items[k]]]) * rng.gamma(4.0, 0.25))
            rows.append((u, int(items[k]), ts[k], ratio, float(durations[items[k]] * 1000.0)))
    rows.append((n_users - 1, n_items - 1, t0 + 70000.0, 1.0, float(durations[n_items - 1] * 1000.0)))   # the largest ids occur in the log
    big = pd.DataFrame(rows, columns=["user_id", "photo_id", "timestamp", "watch_ratio", "photo_duration"])
    big.to_csv(os.path.join(datapath, "big_matrix.csv"), index=False)
    uu, pp = np.meshgrid(env_users, env_items, indexing="ij")
    order = rng.permutation(uu.size)
    su, sp = uu.ravel()[order], pp.ravel()[order]
    sr = np.array([np.mean(affinity[u, list_feat[p]]) for u, p in zip(su, sp)]) * rng.gamma(4.0, 0.25, su.size)
    pd.DataFrame({"user_id": su, "photo_id": sp, "play_duration": 1, "watch_ratio": sr,
                  "photo_duration": durations[sp] * 1000.0}).to_csv(os.path.join(datapath, "small_matrix.csv"), index=False)
    with open(os.path.join(datapath, "item_categories.json"), "w") as fh:
        json.dump({str(i): {"feature_index": f} for i, f in enumerate(list_feat)}, fh)
    with open(os.path.join(datapath, "photo_mean_duration.json"), "w") as fh:
        json.dump({str(i): float(d) for i, d in enumerate(durations)}, fh)
    return dict(list_feat=list_feat, durations=durations, env_users=env_users, env_items=env_items, big=big)
