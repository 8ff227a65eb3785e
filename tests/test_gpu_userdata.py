"""GPU: assembly of the user-model training set (core.user_data.load_dataset_kuaishou: negative sampling and exposure effect on
the device) vs the arrays the reference's load_dataset_kuaishou produced from the same tiny KuaiRec-layout files
(tests/golden/userdata.npz, recorded by oracle/gen_golden.py:gen_userdata)."""
import json
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def write_files(root, z):
    os.makedirs(root, exist_ok=True)
    pd.DataFrame({"user_id": z["big_user"], "photo_id": z["big_photo"], "timestamp": z["big_ts"], "watch_ratio": z["big_ratio"],
                  "photo_duration": z["big_dur"]}).to_csv(os.path.join(root, "big_matrix.csv"), index=False)
    pd.DataFrame({"user_id": z["small_user"], "photo_id": z["small_photo"], "play_duration": 1, "watch_ratio": 1.0,
                  "photo_duration": z["durations"][z["small_photo"]] * 1000.0}).to_csv(os.path.join(root, "small_matrix.csv"), index=False)
    feats = [[int(c) for c in row if c >= 0] for row in z["list_feat"]]
    with open(os.path.join(root, "item_categories.json"), "w") as fh:
        json.dump({str(i): {"feature_index": f} for i, f in enumerate(feats)}, fh)
    with open(os.path.join(root, "photo_mean_duration.json"), "w") as fh:
        json.dump({str(i): float(d) for i, d in enumerate(z["durations"])}, fh)


@pytest.mark.parametrize("tau", [0.0, 800.0])
def test_load_dataset_matches_reference(golden_dir, tmp_path, tau):
    from core.user_data import load_dataset_kuaishou
    z = np.load(os.path.join(golden_dir, "userdata.npz"))
    root = str(tmp_path / "data")
    write_files(root, z)
    save = str(tmp_path / "saved_models" / "env" / "model")
    os.makedirs(save)
    dataset, x_columns, y_columns, ab_columns = load_dataset_kuaishou(tau, 8, 8, save, datapath=root)
    tag = f"tau{int(tau)}"
    x = np.asarray(dataset.x_numpy, np.float64)
    want = z[f"x_{tag}"]
    assert x.shape == want.shape
    # ids, features and the sampled negatives are integers: exact; durations are floats read back from the files
    np.testing.assert_array_equal(x[:, [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12]], want[:, [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12]])
    np.testing.assert_allclose(x[:, [6, 13]], want[:, [6, 13]], rtol=1e-12)
    np.testing.assert_allclose(np.asarray(dataset.y_numpy, np.float64), z[f"y_{tag}"], rtol=1e-12)
    np.testing.assert_allclose(np.asarray(dataset.score, np.float64), z[f"score_{tag}"], rtol=1e-9, atol=1e-12)
    assert [c.name for c in x_columns] == [str(n) for n in z["x_col_names"]]
    assert [int(getattr(c, "vocabulary_size", 0)) for c in x_columns] == z["x_col_vocab"].tolist()
    assert [int(getattr(c, "embedding_dim", getattr(c, "dimension", 0))) for c in x_columns] == z["x_col_dim"].tolist()
    assert [int(c.vocabulary_size) for c in ab_columns] == z["ab_col_vocab"].tolist()
    if tau > 0:   # the exposure effect is cached where the reference caches it and read back on the next call
        cache = os.path.join(save, "..", "saved_exposure", "exposure_pos_{:.1f}.csv".format(tau))
        assert os.path.isfile(cache)
        again = load_dataset_kuaishou(tau, 8, 8, save, datapath=root)[0]
        np.testing.assert_allclose(np.asarray(again.score, np.float64), z[f"score_{tag}"], rtol=1e-9, atol=1e-12)
