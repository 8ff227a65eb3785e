"""Assembly of the user-model training set from the KuaiRec files (reference CIRS-UserModel-kuaishou.py:86-148,
`load_dataset_kuaishou`): positives from big_matrix.csv joined with the item categories, one sampled negative per row, the
exposure effect of every interaction; the two O(big) loops run on the device (core.util.negative_sampling,
core.util.compute_exposure_effect_kuaishouRec)."""
import json
import os

import numpy as np
import pandas as pd

from core.inputs import SparseFeatP
from core.static_dataset import StaticDataset
from core.util import compute_exposure_effect_kuaishouRec, negative_sampling
from deepctr_torch.inputs import DenseFeat

DATAPATH = "environments/KuaishouRec/data"
USER_COLS = ["user_id"]
ITEM_COLS = ["photo_id", "feat0", "feat1", "feat2", "feat3", "photo_duration"]


def item_feature_table(datapath):
    """(list_feat, df_feat): category lists per photo id and the 4-column table shifted by one (0 = padding)."""
    with open(os.path.join(datapath, "item_categories.json")) as fh:
        raw = json.load(fh)
    list_feat = [raw[str(i)]["feature_index"] for i in range(len(raw))]
    df_feat = pd.DataFrame(list_feat, columns=["feat0", "feat1", "feat2", "feat3"])
    df_feat.index.name = "photo_id"
    df_feat = (df_feat.fillna(-1) + 1).astype(int)
    return list_feat, df_feat


def feature_columns(n_user, n_photo, n_feat, entity_dim, feature_dim):
    """The 7 input columns of the DeepFM user model: user, photo, four category slots sharing one table (0 = padding), duration."""
    cols = [SparseFeatP("user_id", n_user, embedding_dim=entity_dim), SparseFeatP("photo_id", n_photo, embedding_dim=entity_dim)]
    cols += [SparseFeatP(f"feat{i}", n_feat, embedding_dim=feature_dim, embedding_name="feat", padding_idx=0) for i in range(4)]
    return cols + [DenseFeat("photo_duration", 1)]


def load_static_validate_data_kuaishou(entity_dim, feature_dim, datapath=None):
    """Validation set = the fully observed small matrix (reference core/util.py:81-133) + the evaluation env's item table."""
    datapath = DATAPATH if datapath is None else datapath
    small = pd.read_csv(os.path.join(datapath, "small_matrix.csv"), usecols=["user_id", "photo_id", "watch_ratio", "photo_duration"])
    small["photo_duration"] /= 1000
    _, df_feat = item_feature_table(datapath)
    small = small.join(df_feat, on=["photo_id"], how="left")
    small.loc[small["watch_ratio"] > 5, "watch_ratio"] = 5
    x_columns = feature_columns(small["user_id"].max() + 1, small["photo_id"].max() + 1, df_feat.max().max() + 1, entity_dim, feature_dim)
    with open(os.path.join(datapath, "photo_mean_duration.json")) as fh:
        mean_dur = {int(k): v for k, v in json.load(fh).items()}
    dataset_val = StaticDataset(x_columns, [DenseFeat("y", 1)], num_workers=4)
    dataset_val.compile_dataset(small[USER_COLS + ITEM_COLS], small[["watch_ratio"]])
    dataset_val.set_env_items(small, df_feat, mean_dur)
    return dataset_val


def load_dataset_kuaishou(tau, entity_dim, feature_dim, MODEL_SAVE_PATH, datapath=None):
    """-> (StaticDataset, x_columns, y_columns, ab_columns) exactly as the reference assembles them."""
    datapath = DATAPATH if datapath is None else datapath
    big = pd.read_csv(os.path.join(datapath, "big_matrix.csv"), usecols=["user_id", "photo_id", "timestamp", "watch_ratio", "photo_duration"])
    big["photo_duration"] /= 1000
    list_feat, df_feat = item_feature_table(datapath)
    big = big.join(df_feat, on=["photo_id"], how="left")
    big.loc[big["watch_ratio"] > 5, "watch_ratio"] = 5

    n_user, n_photo = big["user_id"].max() + 1, big["photo_id"].max() + 1
    x_columns = feature_columns(n_user, n_photo, df_feat.max().max() + 1, entity_dim, feature_dim)
    ab_columns = [SparseFeatP("alpha_u", n_user, embedding_dim=1), SparseFeatP("beta_i", n_photo, embedding_dim=1)]
    y_columns = [DenseFeat("y", 1)]

    pos_x, pos_y = big[USER_COLS + ITEM_COLS], big[["watch_ratio"]]
    neg = negative_sampling(big, df_feat, datapath)
    neg_x = neg[USER_COLS + ITEM_COLS].rename(columns=lambda c: c + "_neg")
    x_all = pd.concat([pos_x, neg_x], axis=1)

    if tau == 0:
        exposure = np.zeros([len(x_all), 1])
    else:
        exposure = compute_exposure_effect_kuaishouRec(pos_x, big["timestamp"], list_feat, tau, MODEL_SAVE_PATH, datapath)

    dataset = StaticDataset(x_columns, y_columns, num_workers=4)
    dataset.compile_dataset(x_all, pos_y, exposure)
    return dataset, x_columns, y_columns, ab_columns
