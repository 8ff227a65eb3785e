"""CPU: the validation-set loader of the user model (core.user_data.load_static_validate_data_kuaishou + StaticDataset.set_env_items)
vs the arrays the reference's loader produced from the same tiny files (tests/golden/userval.npz, oracle/gen_golden.py:gen_userval).
Pure pandas: no device code involved."""
import json
import os

import numpy as np
import pandas as pd


def test_validation_loader_matches_reference(golden_dir, tmp_path):
    from core.user_data import load_static_validate_data_kuaishou
    z = np.load(os.path.join(golden_dir, "userval.npz"))
    root = str(tmp_path)
    pd.DataFrame({"user_id": z["log_user"], "photo_id": z["log_photo"], "play_duration": 1, "watch_ratio": z["log_ratio"],
                  "photo_duration": z["log_dur"]}).to_csv(os.path.join(root, "small_matrix.csv"), index=False)
    feats = [[int(c) for c in row if c >= 0] for row in z["list_feat"]]
    with open(os.path.join(root, "item_categories.json"), "w") as fh:
        json.dump({str(i): {"feature_index": f} for i, f in enumerate(feats)}, fh)
    with open(os.path.join(root, "photo_mean_duration.json"), "w") as fh:
        json.dump({str(i): float(d) for i, d in enumerate(z["durations"])}, fh)
    ds = load_static_validate_data_kuaishou(8, 8, root)
    np.testing.assert_allclose(np.asarray(ds.x_numpy, np.float64), z["x"], rtol=1e-12)
    np.testing.assert_allclose(np.asarray(ds.y_numpy, np.float64), z["y"], rtol=1e-12)
    assert ds.df_photo_env.index.to_numpy().tolist() == z["env_index"].tolist()
    assert list(ds.df_photo_env.columns) == [str(c) for c in z["env_columns"]]
    np.testing.assert_allclose(ds.df_photo_env.to_numpy(dtype=np.float64), z["env_values"], rtol=1e-12)
    assert [int(getattr(c, "vocabulary_size", 0)) for c in ds.x_columns] == z["x_col_vocab"].tolist()
