"""Evaluation callbacks of the CIRS protocol on device trajectories (reference evaluation.py).

  get_feat_dominate_dict   evaluation.py:10-77  ("ifeat_feat": share of recommendations carrying a dominating category)
  Callback_Coverage_Count  evaluation.py:286-371 (CV, CV_turn and ifeat_* for the FB / NX_0 / NX_k test collectors)

The reference walks the replay buffers on the host (buffer.prev / buffer.next); here each collector's buffer carries the
device trajectory of its fused rollout and the counts come from cirs_eval_coverage (integer kernel, bit-exact)."""
import numpy as np
import torch

from cirs_hip.evalmetrics import CoverageCounter, dominated_values, item_flags


def _feat_columns(df_item_val):
    return [c for c in df_item_val.columns if str(c).startswith("feat")]  # .filter(regex="^feat", axis=1)


def get_feat_dominate_dict(df_item_val, all_acts_origin, item_feat_domination, top_rate=0.6):
    """Same contract as the reference: df_item_val is indexed by the ORIGINAL item ids, all_acts_origin are original ids."""
    if item_feat_domination is None:  # for yahoo
        return dict()
    out = {}
    pos = df_item_val.index.get_indexer(np.asarray(all_acts_origin))
    assert (pos >= 0).all(), "recommended item missing from df_item_val"
    counter = CoverageCounter(len(df_item_val))
    if "feat" in item_feat_domination:  # kuairec / kuairand: multi-hot categories in the feat* columns
        flags = item_flags(df_item_val[_feat_columns(df_item_val)].to_numpy(), dominated_values(item_feat_domination["feat"], top_rate))
        _, n, fl = counter.count(torch.as_tensor(pos), torch.as_tensor(flags))
        out["ifeat_feat"] = fl / n
    else:  # coat: one column per feature
        for feat_name, sorted_items in item_feat_domination.items():
            flags = item_flags(df_item_val[[feat_name]].to_numpy(), dominated_values(sorted_items, top_rate))
            _, n, fl = counter.count(torch.as_tensor(pos), torch.as_tensor(flags))
            out["ifeat_" + feat_name] = fl / n
    return out


class Callback_Coverage_Count:
    def __init__(self, test_collector_set, df_item_val, need_transform, item_feat_domination, lbe_photo, top_rate):
        self.collector_dict = test_collector_set.collector_dict
        self.num_items = test_collector_set.env.mat[0].shape[1]
        self.df_item_val = df_item_val
        self.need_transform = need_transform
        self.item_feat_domination = item_feat_domination
        self.lbe_photo = lbe_photo
        self.top_rate = top_rate
        self._counter = None
        self._flags = {}
        if item_feat_domination is not None:
            # flags per ENV item id: env id -> original id (lbe_photo.inverse_transform == classes_[id]) -> df_item_val row
            ids = np.arange(self.num_items)
            # LabelEncoder.inverse_transform(ids) == classes_[ids]
            origin = (self.lbe_photo.inverse_transform(ids) if hasattr(self.lbe_photo, "inverse_transform")
                      else np.asarray(self.lbe_photo.classes_)[ids]) if need_transform else ids
            rows = df_item_val.loc[origin]
            if "feat" in item_feat_domination:
                self._flags["ifeat_feat"] = item_flags(rows[_feat_columns(rows)].to_numpy(), dominated_values(item_feat_domination["feat"], top_rate))
            else:
                for feat_name, sorted_items in item_feat_domination.items():
                    self._flags["ifeat_" + feat_name] = item_flags(rows[[feat_name]].to_numpy(), dominated_values(sorted_items, top_rate))

    def on_epoch_begin(self, epoch):
        pass

    def on_train_begin(self):
        pass

    def on_train_end(self):
        pass

    def on_epoch_end(self, epoch, results=None, **kwargs):
        results_all = {}
        for name, collector in self.collector_dict.items():
            ro = getattr(collector.buffer, "_rollout", None)
            assert ro is not None, "collector has not collected yet"
            act = ro.traj.act  # [T, B] int64 on the device, -1 once an env has finished
            if self._counter is None:
                self._counter = CoverageCounter(self.num_items, device=act.device)
                self._flags = {k: torch.as_tensor(v).to(act.device) for k, v in self._flags.items()}
            res = {}
            hit = n = None
            for key, fl in (self._flags or {None: None}).items():
                hit, n, n_fl = self._counter.count(act, fl)
                if key is not None:
                    res[key] = n_fl / n
            res["CV"] = hit / self.num_items
            res["CV_turn"] = hit / n
            results_all.update({name + "_" + k: v for k, v in res.items()} if name != "FB" else res)
        results.update(results_all)
        return results
