"""StateTrackerTransformer (reference core/state_tracker.py:128-250) on the device engine.

Same constructor keywords and the same three `build_state` call forms the Collector uses (collector.py:127-134,261-269).
Parameters are exposed under the reference's state_dict names but live in ONE flat device buffer; the forward is the
KV-cached decode kernel (csrc/tracker.hip) and the gradient arrives through cirs_tracker_backward (no autograd graph)."""
import numpy as np
import torch
from torch import nn

from cirs_hip.engine import init_tracker_params
from cirs_hip.tracker import DeviceTracker, flat_tracker_params, tracker_param_shapes


class StateTrackerTransformer(nn.Module):
    def __new__(cls, user_columns=None, action_columns=None, feedback_columns=None, *args, **kwargs):
        """VirtualTB-v0 (BASELINE configs[0]: CPU plumbing, dense features) is served by the host tracker of core.host_rl, exactly
        as the reference runs it on the CPU; KuaishouEnv-v0 by the device engine below."""
        dataset = kwargs.get("dataset", args[4] if len(args) > 4 else None)
        if dataset is None:      # not given: sparse id columns mean the KuaishouEnv tracker, all-dense columns the VirtualTB one
            from deepctr_torch.inputs import SparseFeat
            dataset = "KuaishouEnv-v0" if any(isinstance(c, SparseFeat) for c in list(user_columns or []) + list(action_columns or [])) else "VirtualTB-v0"
        if cls is StateTrackerTransformer and dataset == "VirtualTB-v0":
            from core.host_rl import HostStateTracker
            return HostStateTracker(user_columns, action_columns, feedback_columns, *args, **kwargs)
        return super().__new__(cls)

    def __init__(self, user_columns, action_columns, feedback_columns, dim_model, dim_state, dim_max_batch, dropout=0.1,
                 dataset="KuaishouEnv-v0", has_user_embedding=True, has_action_embedding=True, has_feedback_embedding=False,
                 nhead=8, d_hid=128, nlayers=2, device="cpu", seed=2021, init_std=0.0001, padding_idx=None, MAX_TURN=100):
        super().__init__()
        if dataset != "KuaishouEnv-v0":
            raise NotImplementedError("the device tracker serves KuaishouEnv-v0; VirtualTB-v0 dispatches to core.host_rl.HostStateTracker")
        self.dataset, self.device = dataset, torch.device(device if str(device) != "cpu" else "cuda")
        self.dim_model, self.dim_state, self.nhead, self.d_hid, self.nlayers = dim_model, dim_state, nhead, d_hid, nlayers
        self.MAX_TURN = MAX_TURN + 1
        self.n_users, self.n_items = user_columns[0].vocabulary_size, action_columns[0].vocabulary_size
        # nn.Dropout(p) of PositionalEncoding and of both encoder layers (core/state_tracker.py:155-156): live while the module is in
        # training mode -- which in the reference's scripts is ALWAYS (the tracker is not a sub-module of the policy, so
        # policy.eval() never reaches it).  state_tracker.eval() switches it off (the mode of the parity fixtures).
        self.dropout_p = float(dropout)
        init = init_tracker_params(self.n_users, self.n_items, MAX_TURN, seed=seed, dim_model=dim_model, dim_state=dim_state,
                                   nhead=nhead, d_hid=d_hid, nlayers=nlayers, init_std=init_std)
        shapes = tracker_param_shapes(self.n_users, self.n_items, dim_model, dim_state, d_hid, nlayers)
        self.flat, views = flat_tracker_params(shapes, device=self.device, init=init)
        self._views = views
        for name, v in views.items():  # register under the reference's names (dots -> nested attribute path)
            self._register(name, nn.Parameter(v, requires_grad=False))
        self.register_buffer("pe", init["pos_encoder.pe"].to(self.device))
        self._engines = {}   # n_env -> DeviceTracker (all share the flat parameter buffer)
        self._n_env = None
        self._train_state = None  # (flat_grad, grad_views, adam_m, adam_v) shared by whichever engine runs the backward
        self.adam_steps = 0
        # key of the dropout masks of the per-step protocol (build_state without a DeviceRollout): (seed, reset counter) -- every
        # reset starts a rollout with fresh masks, like the reference's fresh nn.Dropout noise at every call
        self._drop_seed, self._drop_resets = int(seed), 0

    def _register(self, dotted, param):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], param)

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        prefix = kwargs.get("prefix", args[1] if len(args) > 1 else "")
        if prefix + "pe" in sd:
            sd[prefix + "pos_encoder.pe"] = sd.pop(prefix + "pe")
        return sd

    def load_state_dict(self, sd, strict=True):
        with torch.no_grad():
            for k, v in self._views.items():
                v.copy_(torch.as_tensor(sd[k]).to(v.device, v.dtype).reshape(v.shape))

    def engine(self, n_env=None, owner=None) -> DeviceTracker:
        """Device engine for a vector env of n_env envs; the parameters and the Adam state are shared, the history slots
        (the reference's self.data) and K/V caches are per engine.  `owner`: every Collector passes itself, so that two
        collectors with the same env count (training_num == test_num == 100 in CIRS-RL-kuaishou.py) never share slots — the
        tracker backward of policy.update(train_buffer) re-reads the slots of the TRAIN rollout even if a test rollout ran
        in between (onpolicy.py:65-72 with stop_fn + test_in_train).  owner=None: the per-step build_state protocol."""
        n_env = n_env or self._n_env
        key = n_env if owner is None else (n_env, id(owner))
        if owner is not None:
            self._owners = getattr(self, "_owners", {})
            self._owners[id(owner)] = owner      # pin the id
        if key not in self._engines:
            params = dict(self._views)
            params["pos_encoder.pe"] = self.pe
            eng = DeviceTracker(params, self.n_users, self.n_items, n_env, self.MAX_TURN - 1, dim_model=self.dim_model,
                                dim_state=self.dim_state, nhead=self.nhead, d_hid=self.d_hid, nlayers=self.nlayers,
                                device=self.device)
            eng.set_dropout(self.dropout_p if self.training else 0.0)
            eng.enable_training(self.flat)
            if self._train_state is None:
                self._train_state = (eng.flat_grad, eng.grad_views, eng.g, eng.adam_m, eng.adam_v)
            else:
                eng.flat_grad, eng.grad_views, eng.g, eng.adam_m, eng.adam_v = self._train_state
            self._engines[key] = eng
        if owner is None:
            self._n_env = n_env
        return self._engines[key]

    def train(self, mode: bool = True):
        super().train(mode)
        for eng in getattr(self, "_engines", {}).values():
            eng.set_dropout(self.dropout_p if mode else 0.0)
        return self

    def build_state(self, obs=None, env_id=None, obs_next=None, rew=None, done=None, info=None, policy=None, dim_batch=None,
                    reset=False):
        if reset and dim_batch:
            eng = self.engine(dim_batch)
            eng.reset()
            # the per-step engine is keyed here (DeviceRollout.collect keys the fused one): without it every episode of the
            # stepwise protocol (Collector.collect(random=True), external preprocess_fn users) would reuse the masks of key 0
            self._drop_resets += 1
            eng.set_dropout_key(self._drop_seed, (1 << 40) + self._drop_resets, 0)
            return
        eng = self.engine(self._n_env)
        ids = None if env_id is None else torch.as_tensor(np.asarray(env_id).astype(np.int32)).to(self.device)
        if obs is not None:
            users = torch.as_tensor(np.asarray(obs).reshape(-1))
            return {"obs": eng.init(users, ids)}
        if obs_next is not None:
            items = torch.as_tensor(np.asarray(obs_next).reshape(-1))
            r = torch.as_tensor(np.asarray(rew, dtype=np.float64).reshape(-1))
            return {"obs_next": eng.step(items, r, ids)}
        return {}
