"""UserModel_Pairwise (reference core/user_model_pairwise.py:14-154): the DeepFM user model, forward on the device.

Constructor keywords and state_dict names match the shipped `DeepFM_params_Pair11.pickle` / `DeepFM_Pair11.pt`
(SURVEY Appendix C).  Training (`fit_data`, losses) is offline and out of scope."""
import numpy as np
import torch
from torch import nn

from core.inputs import SparseFeatP, compute_input_dim
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


class UserModel_Pairwise(nn.Module):
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

    # ---- training (reference core/user_model.py:74-170) ---------------------------------------------------------------
    def compile(self, optimizer, loss_dict=None, metrics=None, metric_fun=None, loss_func=None):
        assert optimizer == "adam" or isinstance(optimizer, torch.optim.Adam), "the device step implements torch.optim.Adam"
        assert loss_func is not None and hasattr(loss_func, "lambda_ab"), \
            "pass core.user_model_pairwise.make_loss_kuaishou_pairwise(lambda_ab): the loss runs inside cirs_deepfm_train_step"
        self.metrics_names = ["loss"]
        self.loss_func, self.metric_fun, self.metrics = loss_func, metric_fun, metrics
        self.optim = "adam"
        self._lr = optimizer.param_groups[0]["lr"] if isinstance(optimizer, torch.optim.Adam) else 1e-3

    def fit_data(self, dataset_train, dataset_val=None, batch_size=256, epochs=1, verbose=1, initial_epoch=0, callbacks=None, shuffle=True):
        """One pass per epoch over (x, y, score) minibatches; every step is cirs_deepfm_train_step on the device."""
        from cirs_hip.deepfm_train import DeepFMTrainer
        assert self.optim is not None, "call compile() first"
        if self._trainer is None:
            self._trainer = DeepFMTrainer(self.state_dict(), use_ab=self.ab_columns is not None, lambda_ab=self.loss_func.lambda_ab,
                                          l2_embedding=self._l2[0], l2_linear=self._l2[1], l2_all=self._l2[2], lr=self._lr)
        tr = self._trainer
        x = torch.as_tensor(dataset_train.x_numpy).to(tr.device, torch.float32)
        y = torch.as_tensor(dataset_train.y_numpy).to(tr.device, torch.float32)
        score = torch.as_tensor(dataset_train.score).to(tr.device, torch.float32)
        n_all = x.shape[0]
        callbacks = callbacks or []
        for cb in callbacks:
            cb.on_train_begin()
        history = []
        for epoch in range(initial_epoch, epochs):
            for cb in callbacks:
                cb.on_epoch_begin(epoch)
            order = torch.randperm(n_all, device=tr.device) if shuffle else torch.arange(n_all, device=tr.device)
            loss_sum = torch.zeros((), device=tr.device)
            for s0 in range(0, n_all, batch_size):
                idx = order[s0:s0 + batch_size]
                lo = tr.step(x[idx], y[idx], score[idx])
                loss_sum += lo[0] + lo[4]
            logs = {"loss": float(loss_sum) / n_all}       # total_loss_epoch / sample_num (core/user_model.py:205)
            history.append(logs)
            for cb in callbacks:
                cb.on_epoch_end(epoch, logs)
        for cb in callbacks:
            cb.on_train_end()
        # publish the trained parameters under the module's state_dict names
        with torch.no_grad():
            mine = dict(self.named_parameters())
            for k, v in tr.state_dict().items():
                if k in mine:
                    mine[k].copy_(v.reshape(mine[k].shape).to(mine[k].device))
        self._dev = None
        return history

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

    # ---- static-baseline recommendation (reference core/user_model.py:250-348) ------------------------------------------
    def compile_UCB(self, n_arm):
        self.n_rec = n_arm
        self.n_each = np.ones(n_arm)

    def recommend_k_item(self, user, dataset_val, k=1, is_softmax=True, epsilon=0, is_ucb=False, recommended_ids=[], gumbel=None,
                         seed=None):
        """One catalogue sweep for `user` (original id) over dataset_val.df_photo_env, then the choice of ONE item on the
        device (cirs_select_items).  Returns (recommended_id_transform, recommended_id_raw, value_rec) like the reference:
        position in df_photo_env, original id, u_value of the pick.  k > 1 is not built (the scripts use k = 1)."""
        assert k == 1, "only k = 1 is built (interactive_evaluation / test_kuaishou call with k=1)"
        from cirs_hip.static_policy import select_items
        df_item_val = dataset_val.df_photo_env
        item_index = df_item_val.index.to_numpy()
        I = len(item_index)
        dm = self.device_model()
        feats = df_item_val[["feat0", "feat1", "feat2", "feat3"]].to_numpy()
        dur = df_item_val["photo_duration"].to_numpy()
        pred, _ = dm.sweep(np.asarray([user]), item_index, feats, dur)          # [1, I] on the device
        visited = None
        if len(recommended_ids):
            words = np.zeros((I + 31) // 32, dtype=np.uint32)
            ids = np.asarray(recommended_ids, dtype=np.int64)
            np.bitwise_or.at(words, ids >> 5, (np.uint32(1) << (ids & 31).astype(np.uint32)))
            visited = torch.as_tensor(words.view(np.int32)).reshape(1, -1)
        bonus = None
        if is_ucb and len(recommended_ids) == 0:
            if not hasattr(self, "n_rec"):
                self.compile_UCB(I)
            bonus = torch.as_tensor(((2 * np.log(self.n_rec) / self.n_each) ** 0.5).astype(np.float32))
        self._rec_calls = getattr(self, "_rec_calls", 0) + 1
        act, val = select_items(pred, softmax=is_softmax, bonus=bonus, visited=visited, epsilon=float(epsilon), gumbel=gumbel,
                                seed=self.seed_rec if seed is None else seed, rng_step=self._rec_calls)
        recommended_id_transform = act.cpu().numpy()
        if is_ucb:
            self.n_rec += k
            self.n_each[recommended_id_transform] += 1
        return recommended_id_transform, item_index[recommended_id_transform], val.cpu().numpy()

    seed_rec = 2022

    def forward(self, x):
        """x: float tensor (n, 7) = [user_id, photo_id, feat0..3, photo_duration] carrying raw ids."""
        x = torch.as_tensor(x)
        ids = x[:, :6].long()
        y = self.device_model().forward(ids[:, 0], ids[:, 1], ids[:, 2:6].int(), x[:, 6].float())
        return y.unsqueeze(1)
