"""UserModel_Pairwise (reference core/user_model_pairwise.py:14-154): the DeepFM user model, forward on the device.

Constructor keywords and state_dict names match the shipped `DeepFM_params_Pair11.pickle` / `DeepFM_Pair11.pt`
(SURVEY Appendix C).  compile / fit_data / recommend_k_item live in the base class core.user_model.UserModel, like in the reference."""
import numpy as np
import torch
from torch import nn

from core.inputs import SparseFeatP, compute_input_dim
from core.user_model import UserModel
from deepctr_torch.inputs import DenseFeat, build_input_features


def make_loss_kuaishou_pairwise(lambda_ab: float):
    """loss_kuaishou_pairwise of CIRS-UserModel-kuaishou.py:262-278 with its `args.lambda_ab` bound.  The returned callable
    is the torch formula (documentation / host use); fit_data recognises it by its `lambda_ab` attribute and runs the same
    loss inside cirs_deepfm_train_step."""
    def loss_kuaishou_pairwise(y, y_deepfm_pos, y_deepfm_neg, exposure, alpha_u=None, beta_i=None):
        if alpha_u is not None:
            exposure_new = exposure * alpha_u * beta_i
            loss_ab = ((alpha_u - 1) ** 2).mean() + ((beta_i - 1) ** 2).mean()
        else:
            exposure_new, loss_ab = exposure, 0
        y_exposure = 1 / (1 + exposure_new) * y_deepfm_pos
        return ((y_exposure - y) ** 2).mean() - torch.sigmoid(y_deepfm_pos - y_deepfm_neg).log().mean() + lambda_ab * loss_ab
    loss_kuaishou_pairwise.lambda_ab = float(lambda_ab)
    return loss_kuaishou_pairwise


class UserModel_Pairwise(UserModel):
    def __init__(self, feature_columns, y_columns, task, task_logit_dim, dnn_hidden_units=(128, 128), l2_reg_embedding=1e-5,
                 l2_reg_dnn=1e-1, init_std=0.0001, task_dnn_units=None, seed=2022, dnn_dropout=0, dnn_activation="relu",
                 dnn_use_bn=False, device="cpu", padding_idx=None, ab_columns=None, l2_reg_linear=1e-5):
        super().__init__()
        assert task == "regression" and task_logit_dim == 1 and tuple(dnn_hidden_units) == (64, 64) and not dnn_use_bn
        self.feature_columns, self.y_columns = feature_columns, y_columns
        self.feature_index = build_input_features(feature_columns)
        self.device = device
        g = torch.Generator().manual_seed(seed)
        sparse = [f for f in feature_columns if isinstance(f, SparseFeatP)]
        names = {}
        for f in sparse:
            names.setdefault(f.embedding_name, f)
        self.embedding_dict = nn.ModuleDict({k: nn.Embedding(int(f.vocabulary_size), int(f.embedding_dim)) for k, f in names.items()})
        self.linear = nn.Module()
        self.linear.embedding_dict = nn.ModuleDict({k: nn.Embedding(int(f.vocabulary_size), 1) for k, f in names.items()})
        self.linear.weight = nn.Parameter(torch.randn(sum(f.dimension for f in feature_columns if isinstance(f, DenseFeat)), 1, generator=g) * init_std)
        self.linear_model = nn.Module()  # unused duplicate kept for state_dict compatibility
        self.linear_model.embedding_dict = nn.ModuleDict({k: nn.Embedding(int(f.vocabulary_size), 1) for k, f in names.items()})
        self.linear_model.weight = nn.Parameter(torch.zeros_like(self.linear.weight))
        k_in = compute_input_dim(feature_columns)
        self.dnn = nn.Module()
        self.dnn.linears = nn.ModuleList([nn.Linear(k_in, 64), nn.Linear(64, 64)])
        self.last = nn.Linear(64, 1, bias=False)
        self.out = nn.Module()
        self.out.bias = nn.Parameter(torch.zeros(1, 1))
        self.ab_columns = ab_columns
        if ab_columns is not None:
            self.ab_embedding_dict = nn.ModuleDict({c.embedding_name: nn.Embedding(int(c.vocabulary_size), 1) for c in ab_columns})
        self._dev = None
        self._l2 = (float(l2_reg_embedding), float(l2_reg_linear), float(l2_reg_dnn))
        self._trainer = None
        self.optim = None

    def device_model(self):
        """DeviceDeepFM over the current weights (rebuilt after load_state_dict)."""
        if self._dev is None:
            from cirs_hip.deepfm import DeviceDeepFM
            sd = self.state_dict()
            sd = dict(sd)
            sd["embedding_dict.user_id.weight"] = sd["embedding_dict.user_id.weight"]
            self._dev = DeviceDeepFM(sd)
        return self._dev

    def load_state_dict(self, state_dict, strict=True):
        self._dev = None
        return super().load_state_dict(state_dict, strict=False)

    def forward(self, x):
        """x: float tensor (n, 7) = [user_id, photo_id, feat0..3, photo_duration] carrying raw ids."""
        x = torch.as_tensor(x)
        ids = x[:, :6].long()
        y = self.device_model().forward(ids[:, 0], ids[:, 1], ids[:, 2:6].int(), x[:, 6].float())
        return y.unsqueeze(1)
