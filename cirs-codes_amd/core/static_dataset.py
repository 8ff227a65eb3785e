"""StaticDataset: container of the user-model training log, API of reference core/static_dataset.py:9-54.

x holds, per logged interaction, the positive pair's columns followed by the sampled negative pair's (14 columns for
KuaiRec: user, photo, feat0..3, duration, twice), y the label (watch ratio) and score the pre-computed exposure effect.
`fit_data` of the device build slices these arrays directly; `get_dataset_train` exists for code written against the
reference's DataLoader route."""
import numpy as np
import torch


def _as_array(table):
    return table.to_numpy() if hasattr(table, "to_numpy") else np.asarray(table)


class StaticDataset:
    def __init__(self, x_columns, y_columns, num_workers=4):
        self.x_columns, self.y_columns = x_columns, y_columns
        self.num_workers = num_workers
        self.neg_items_info = None
        self.x_numpy = self.y_numpy = self.score = None

    def set_env_items(self, df_small, df_feat, photo_mean_duration):
        """Per-item table of the evaluation environment (reference static_dataset.py:19-26): the items that occur in the small
        matrix, their 4 (shifted) category columns and mean duration, ordered by photo id."""
        items = np.sort(df_small["photo_id"].unique())
        table = df_feat.loc[items].copy()
        table["photo_duration"] = [photo_mean_duration[int(i)] for i in items]
        self.df_photo_env = table

    def compile_dataset(self, df_x, df_y, score=None):
        self.x_numpy, self.y_numpy = _as_array(df_x), _as_array(df_y)
        n = len(self.x_numpy)
        assert len(self.y_numpy) == n, "x and y must have one row per interaction"
        self.score = np.zeros((n, 1)) if score is None else _as_array(score)

    @property
    def len(self):
        return 0 if self.x_numpy is None else len(self.x_numpy)

    def __len__(self):
        return self.len

    def get_y(self):
        return self.y_numpy

    def get_dataset_train(self):
        cols = (self.x_numpy, self.y_numpy, self.score)
        return torch.utils.data.TensorDataset(*(torch.from_numpy(np.ascontiguousarray(c)) for c in cols))
