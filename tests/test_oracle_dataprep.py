"""CPU: dataset preparation of the user-model training -- the C oracle against outputs of the reference's
compute_exposure_effect_kuaishouRec and find_negative."""
import os

import numpy as np

import prepcase
from cirs_hip.dataprep import bitmap_rows


def test_exposure_history_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "dataprep.npz"))
    words = prepcase.cats_words(z["list_feat"])
    for tau in (1000, 50):
        got = prepcase.oracle_exposure(z["user_id"], z["photo_id"], z["timestamp"], float(tau), words=words)
        np.testing.assert_allclose(got, z[f"exposure_tau{tau}"], rtol=1e-13, atol=0)
        first = np.r_[True, z["user_id"][1:] != z["user_id"][:-1]]
        assert (got[first] == 0).all()


def test_find_negative_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "dataprep.npz"))
    I = int(z["n_items2"])
    small, big = prepcase.unpack_bits(z["mat_small"], I), prepcase.unpack_bits(z["mat_big"], I)
    got = prepcase.oracle_negative(z["neg_users"], z["neg_items"], bitmap_rows(small), bitmap_rows(big), I)
    np.testing.assert_array_equal(got, z["negatives"][:, 1].astype(np.int64))
    np.testing.assert_array_equal(z["neg_users"], z["negatives"][:, 0].astype(np.int64))
    assert 1225 not in got and not small[z["neg_users"], got].any() and not big[z["neg_users"], got].any()
