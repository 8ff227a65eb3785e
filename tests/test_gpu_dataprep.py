"""GPU: cirs_exposure_history / cirs_find_negative against the reference's recorded outputs and the C oracle at scale."""
import os

import numpy as np
import pytest

import prepcase

pytestmark = pytest.mark.gpu


def test_exposure_history_golden_and_scale(golden_dir):
    from cirs_hip.dataprep import exposure_history
    z = np.load(os.path.join(golden_dir, "dataprep.npz"))
    list_feat = [[int(c) for c in f if c >= 0] for f in z["list_feat"]]
    for tau in (1000, 50):
        got = exposure_history(z["user_id"], z["photo_id"], z["timestamp"], float(tau), list_feat=list_feat).cpu().numpy()
        np.testing.assert_allclose(got, z[f"exposure_tau{tau}"], rtol=1e-12, atol=0)
    # larger log: 200 users x up to 1500 interactions, table and on-the-fly distances agree with the oracle
    rng = np.random.RandomState(0)
    n_items = 500
    cats = np.where(np.arange(4)[None, :] < rng.randint(1, 5, n_items)[:, None], rng.randint(0, 31, (n_items, 4)), -1)
    lf = [sorted(set(int(c) for c in r if c >= 0)) for r in cats]
    users, photos, ts = [], [], []
    for u in range(200):
        L = rng.randint(1, 1500)
        users += [u] * L; photos += rng.randint(0, n_items, L).tolist(); ts += np.sort(1.6e9 + rng.randint(0, 100000, L)).astype(np.float64).tolist()
    words = prepcase.cats_words([r + [-1] * (4 - len(r)) for r in lf])
    want = prepcase.oracle_exposure(users, photos, ts, 300.0, words=words)
    got = exposure_history(users, photos, ts, 300.0, list_feat=lf).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-300)
    from cirs_hip.synthetic import jaccard_distance
    dist = jaccard_distance(np.array([r + [-1] * (4 - len(r)) for r in lf], np.int32))
    got_t = exposure_history(users, photos, ts, 300.0, dist=dist).cpu().numpy()
    np.testing.assert_allclose(got_t, want, rtol=1e-12, atol=1e-300)


def test_find_negative_golden_and_scale(golden_dir):
    from cirs_hip.dataprep import bitmap_rows, find_negative
    z = np.load(os.path.join(golden_dir, "dataprep.npz"))
    I = int(z["n_items2"])
    small, big = prepcase.unpack_bits(z["mat_small"], I), prepcase.unpack_bits(z["mat_big"], I)
    got = find_negative(z["neg_users"], z["neg_items"], bitmap_rows(small), bitmap_rows(big), I).cpu().numpy()
    np.testing.assert_array_equal(got, z["negatives"][:, 1].astype(np.int64))
    rng = np.random.RandomState(1)
    U, I = 300, 10729
    small = rng.uniform(size=(U, I)) < 0.4; big = rng.uniform(size=(U, I)) < 0.5
    small[0] = True; big[0] = True; small[0, 1225] = False          # a user who has seen everything but the absent id -> -1
    users = rng.randint(0, U, 50000); items = rng.randint(0, I, 50000); users[:3] = 0
    a, b = bitmap_rows(small), bitmap_rows(big)
    got = find_negative(users, items, a, b, I).cpu().numpy()
    np.testing.assert_array_equal(got, prepcase.oracle_negative(users, items, a, b, I))
    assert (got[:3] == -1).all()
