"""Scratch probe: cirs_exposure_history at KuaiRec big-matrix proportions (about 1750 interactions per user)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np, torch
from cirs_hip.dataprep import exposure_history
rng = np.random.RandomState(0)
n_users, per, n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 1750, 10729
cats = np.where(np.arange(4)[None, :] < rng.randint(1, 5, n_items)[:, None], rng.randint(0, 31, (n_items, 4)), -1)
lf = [sorted(set(int(c) for c in r if c >= 0)) for r in cats]
users = np.repeat(np.arange(n_users), per); photos = rng.randint(0, n_items, n_users * per)
ts = (1.6e9 + np.sort(rng.randint(0, 5_000_000, (n_users, per)), axis=1)).reshape(-1).astype(np.float64)
exposure_history(users[:per], photos[:per], ts[:per], 100.0, list_feat=lf)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = exposure_history(users, photos, ts, 100.0, list_feat=lf)
torch.cuda.synchronize()
t = time.perf_counter() - t0
terms = n_users * per * (per - 1) / 2
print(json.dumps(dict(rows=n_users * per, pair_terms=terms, seconds=t, terms_per_s=terms / t, rows_per_s=n_users * per / t,
                      extrapolated_seconds_for_kuairec_big=1.1e10 / (terms / t))))
