"""Feature-domination statistics of the training log (reference environments/KuaishouRec/env/data_handler.py:98-122).

get_sorted_domination_features: share of every category among the category slots of the well-liked interactions
(`yname >= threshold`), sorted descending -- the list Callback_Coverage_Count / get_feat_dominate_dict consume.
Setup-time host code (collections.Counter over the log), not on the rollout path.

Also the loaders CIRS-RL-kuaishou.py calls before training (reference data_handler.py:18-93,124-138): load_category, get_lbe,
load_item_feat, get_df_kuairec, get_training_item_domination over the KuaiRec files under DATAPATH
(`item_categories.json`, `small_matrix.csv`, `big_matrix.csv`; the reference ships none of them -- tests and the example write
synthetic files of the same format).  DATAPATH defaults to environments/KuaishouRec/data like the reference and follows the
CIRS_DATAPATH environment variable when that is set."""
import collections
import json
import os
import pickle

import numpy as np

CODEPATH = os.path.dirname(os.path.abspath(__file__))
ROOTPATH = os.path.dirname(CODEPATH)
DATAPATH = os.environ.get("CIRS_DATAPATH") or os.path.join(ROOTPATH, "data")


def _datapath():
    return os.environ.get("CIRS_DATAPATH") or DATAPATH


def load_category():
    """item_categories.json -> (list of category lists per photo id, DataFrame feat0..feat3 with ids shifted by one, 0 = no category)."""
    import pandas as pd
    with open(os.path.join(_datapath(), "item_categories.json")) as fh:
        data_feat = json.load(fh)
    list_feat = [data_feat[str(i)]["feature_index"] for i in range(len(data_feat))]
    feat = np.zeros((len(list_feat), 4), dtype=np.int64)
    for i, cats in enumerate(list_feat):
        feat[i, :len(cats)] = np.asarray(cats, dtype=np.int64) + 1
    df_feat = pd.DataFrame(feat, columns=["feat0", "feat1", "feat2", "feat3"])
    df_feat.index.name = "photo_id"
    return list_feat, df_feat


def get_lbe():
    """LabelEncoders over the users / photos of small_matrix.csv (cached as user_id_small.csv / item_id_small.csv, like the reference)."""
    import pandas as pd
    from sklearn.preprocessing import LabelEncoder
    root = _datapath()
    up, ip = os.path.join(root, "user_id_small.csv"), os.path.join(root, "item_id_small.csv")
    if os.path.isfile(up) and os.path.isfile(ip):
        users, items = pd.read_csv(up)["user_id_small"], pd.read_csv(ip)["item_id_small"]
    else:
        small = pd.read_csv(os.path.join(root, "small_matrix.csv"), header=0, usecols=["user_id", "photo_id"])
        users = pd.Series(small["user_id"].unique(), name="user_id_small")
        items = pd.Series(small["photo_id"].unique(), name="item_id_small")
        users.to_frame().to_csv(up, index=False)
        items.to_frame().to_csv(ip, index=False)
    return LabelEncoder().fit(users), LabelEncoder().fit(items)


def load_item_feat(only_small=False):
    """feat0..feat3 per photo; only_small: restricted to (and ordered like) the env's items."""
    _, df_item = load_category()
    if only_small:
        _, lbe_item = get_lbe()
        return df_item.loc[lbe_item.classes_]
    return df_item


def get_df_kuairec(name="big_matrix.csv"):
    """(log joined with the item features, item feature frame, category lists) of one KuaiRec interaction file."""
    import pandas as pd
    df_data = pd.read_csv(os.path.join(_datapath(), name), usecols=["user_id", "photo_id", "watch_ratio"])
    list_feat, df_feat = load_category()
    df_item = load_item_feat(only_small=(name != "big_matrix_processed.csv"))
    df_data = df_data.join(df_feat, on=["photo_id"], how="left")
    return df_data, df_item, list_feat


def get_training_item_domination():
    """Category shares among the well-liked (top 20 % watch_ratio) interactions of the training log, cached next to the data."""
    cache = os.path.join(_datapath(), "feature_domination.pickle")
    if os.path.isfile(cache):
        with open(cache, "rb") as fh:
            return pickle.load(fh)
    df_data, df_item, _ = get_df_kuairec("big_matrix.csv")
    dom = get_sorted_domination_features(df_data, df_item, is_multi_hot=True, yname="watch_ratio",
                                         threshold=np.percentile(df_data["watch_ratio"], 80))
    with open(cache, "wb") as fh:
        pickle.dump(dom, fh)
    return dom


def get_sorted_domination_features(df_data, df_item, is_multi_hot, yname=None, threshold=None):
    item_feat_domination = dict()
    if not is_multi_hot:  # for coat
        for x in df_item.columns.to_list():
            sorted_count = collections.Counter(df_data[x])
            sorted_percentile = {k: v / len(df_data) for k, v in dict(sorted_count).items()}
            item_feat_domination[x] = sorted(sorted_percentile.items(), key=lambda kv: kv[1], reverse=True)
    else:  # for kuairec and kuairand
        cols = [c for c in df_item.columns if str(c).startswith("feat")]
        feat_train = df_data.loc[df_data[yname] >= threshold, cols]
        cats_train = feat_train.to_numpy().reshape(-1)
        pos_cat_train = cats_train[cats_train > 0]
        sorted_count = collections.Counter(pos_cat_train)
        total = sum(sorted_count.values())
        sorted_percentile = {k: v / total for k, v in dict(sorted_count).items()}
        item_feat_domination["feat"] = sorted(sorted_percentile.items(), key=lambda kv: kv[1], reverse=True)
    return item_feat_domination
