"""UserModel (reference core/user_model.py:29-581): the base class of the CIRS user models — training loop surface
(`compile`, `fit_data`), static-baseline recommendation (`compile_UCB`, `recommend_k_item`) — delegating every computation to
the device model of the subclass (`device_model()` -> cirs_hip.deepfm.DeviceDeepFM, `cirs_deepfm_train_step`,
`cirs_select_items`).  Subclasses build the parameters under the reference's state_dict names (UserModel_Pairwise)."""
import numpy as np
import torch
from torch import nn

from core.inputs import compute_input_dim  # noqa: F401  (re-exported like the reference module does)
from deepctr_torch.inputs import build_input_features  # noqa: F401


class UserModel(nn.Module):
    seed_rec = 2022

    def device_model(self):
        raise NotImplementedError("subclasses bind their parameters to a device model")

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

    # ---- static-baseline recommendation (reference core/user_model.py:250-348) ------------------------------------------
    def compile_UCB(self, n_arm):
        self.n_rec = n_arm
        self.n_each = np.ones(n_arm)

    def recommend_k_item(self, user, dataset_val, k=1, is_softmax=True, epsilon=0, is_ucb=False, recommended_ids=[], gumbel=None,
                         seed=None):
        """One catalogue sweep for `user` (original id) over dataset_val.df_photo_env, then the choice of k items on the device
        (cirs_select_items).  Returns (recommended_id_transform, recommended_id_raw, value_rec) like the reference (core/user_model.py:254-346):
        positions in df_photo_env, original ids, u_value of the picks.
        k > 1: `torch.multinomial(softmax, k, replacement=False)` draws item after item from the renormalised rest and `torch.topk` is a
        repeated arg-max -- both are k selections with the already chosen items removed, which is how they run here (k launches of the
        selection kernel over the same scores; with harness noise `gumbel` the k picks are the Gumbel top-k of logit + noise, i.e. one
        sample without replacement).  The epsilon-greedy branch replaces all k picks by uniform draws (with repetition, like
        `torch.randint(0, n, (k,))`)."""
        from cirs_hip.static_policy import select_items
        df_item_val = dataset_val.df_photo_env
        item_index = df_item_val.index.to_numpy()
        I = len(item_index)
        assert 1 <= k <= I - len(recommended_ids), "k exceeds the number of items left"
        dm = self.device_model()
        feats = df_item_val[["feat0", "feat1", "feat2", "feat3"]].to_numpy()
        dur = df_item_val["photo_duration"].to_numpy()
        pred, _ = dm.sweep(np.asarray([user]), item_index, feats, dur)          # [1, I] on the device
        words = np.zeros((I + 31) // 32, dtype=np.uint32)
        if len(recommended_ids):
            ids = np.asarray(recommended_ids, dtype=np.int64)
            np.bitwise_or.at(words, ids >> 5, (np.uint32(1) << (ids & 31).astype(np.uint32)))
        bonus = None
        if is_ucb and len(recommended_ids) == 0:
            if not hasattr(self, "n_rec"):
                self.compile_UCB(I)
            bonus = torch.as_tensor(((2 * np.log(self.n_rec) / self.n_each) ** 0.5).astype(np.float32))
        explore = k > 1 and epsilon > 0 and np.random.random() < epsilon     # k = 1: the kernel's own epsilon draw (bit-exact vs the oracle)
        picks, vals = [], []
        for i in range(k):
            self._rec_calls = getattr(self, "_rec_calls", 0) + 1
            visited = torch.as_tensor(words.view(np.int32)).reshape(1, -1) if (len(recommended_ids) or i > 0) else None
            act, val = select_items(pred, softmax=is_softmax, bonus=bonus, visited=visited,
                                    epsilon=float(epsilon) if k == 1 else (1.0 if explore else 0.0), gumbel=gumbel,
                                    seed=self.seed_rec if seed is None else seed, rng_step=self._rec_calls)
            a = int(act.cpu()[0])
            picks.append(a); vals.append(float(val.cpu()[0]))
            if not explore:
                words[a >> 5] |= np.uint32(1) << np.uint32(a & 31)
        recommended_id_transform = np.asarray(picks, dtype=np.int64)
        if is_ucb:
            self.n_rec += k
            self.n_each[recommended_id_transform] += 1
        return recommended_id_transform, item_index[recommended_id_transform], np.asarray(vals, dtype=np.float32)

