"""StaticDataset (reference core/static_dataset.py:9-54): the (x, y, score) arrays of the user-model training log."""
import numpy as np
import torch


class StaticDataset:
    def __init__(self, x_columns, y_columns, num_workers=4):
        self.x_columns, self.y_columns, self.num_workers = x_columns, y_columns, num_workers
        self.len = 0
        self.neg_items_info = None

    def compile_dataset(self, df_x, df_y, score=None):
        self.x_numpy = df_x.to_numpy() if hasattr(df_x, "to_numpy") else np.asarray(df_x)
        self.y_numpy = df_y.to_numpy() if hasattr(df_y, "to_numpy") else np.asarray(df_y)
        self.score = np.zeros([len(self.x_numpy), 1]) if score is None else score
        self.len = len(self.x_numpy)

    def get_dataset_train(self):
        return torch.utils.data.TensorDataset(torch.from_numpy(self.x_numpy), torch.from_numpy(self.y_numpy), torch.from_numpy(self.score))

    def get_y(self):
        return self.y_numpy

    def __len__(self):
        return self.len
