"""Feature-domination statistics of the training log (reference environments/KuaishouRec/env/data_handler.py:98-122).

get_sorted_domination_features: share of every category among the category slots of the well-liked interactions
(`yname >= threshold`), sorted descending -- the list Callback_Coverage_Count / get_feat_dominate_dict consume.
Setup-time host code (collections.Counter over the log), not on the rollout path."""
import collections


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
