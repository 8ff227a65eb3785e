"""Device engine of the state tracker: weights + KV caches in HBM, one launch per step (csrc/tracker.hip)."""
import ctypes as C
import math
from typing import Dict, Optional

import torch

from . import abi

LAYER_FIELDS = [("in_proj_w", "self_attn.in_proj_weight"), ("in_proj_b", "self_attn.in_proj_bias"),
                ("out_proj_w", "self_attn.out_proj.weight"), ("out_proj_b", "self_attn.out_proj.bias"),
                ("lin1_w", "linear1.weight"), ("lin1_b", "linear1.bias"), ("lin2_w", "linear2.weight"),
                ("lin2_b", "linear2.bias"), ("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"),
                ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias")]
TOP_FIELDS = [("emb_user", "embedding_dict.feat_user.weight"), ("emb_item", "embedding_dict.feat_item.weight"),
              ("ffn_user_w", "ffn_user.weight"), ("ffn_user_b", "ffn_user.bias"), ("gate_w", "fnn_gate.weight"),
              ("gate_b", "fnn_gate.bias"), ("dec_w", "decoder.weight"), ("dec_b", "decoder.bias")]


def positional_encoding(max_len: int, d_model: int) -> torch.Tensor:
    """pe[max_len, d_model] (reference core/state_tracker.py:255-271; even d_model)."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)[:, : d_model // 2]
    return pe


def weights_struct(params: Dict[str, torch.Tensor], pe: torch.Tensor, nlayers: int):
    """Build cirs_tracker_weights from tensors keyed by the reference's state_dict names (fp32, contiguous, device)."""
    w = abi.TrackerWeights()
    for f, name in TOP_FIELDS:
        t = params[name]
        assert t.dtype == torch.float32 and t.is_contiguous(), name
        setattr(w, f, t.data_ptr())
    w.pe = pe.data_ptr()
    for l in range(nlayers):
        for f, name in LAYER_FIELDS:
            t = params[f"transformer_encoder.layers.{l}.{name}"]
            assert t.dtype == torch.float32 and t.is_contiguous(), name
            setattr(w.layer[l], f, t.data_ptr())
    return w


def tracker_param_shapes(n_users, n_items, dim_model=32, dim_state=20, d_hid=128, nlayers=2):
    """Ordered {reference state_dict name: shape} of the trainable tracker tensors."""
    D, S, H = dim_model, dim_state, d_hid
    sh = {"embedding_dict.feat_user.weight": (n_users, D), "embedding_dict.feat_item.weight": (n_items, D),
          "ffn_user.weight": (D, D), "ffn_user.bias": (D,), "fnn_gate.weight": (D, D + 1), "fnn_gate.bias": (D,)}
    for l in range(nlayers):
        pre = f"transformer_encoder.layers.{l}."
        sh.update({pre + "self_attn.in_proj_weight": (3 * D, D), pre + "self_attn.in_proj_bias": (3 * D,),
                   pre + "self_attn.out_proj.weight": (D, D), pre + "self_attn.out_proj.bias": (D,),
                   pre + "linear1.weight": (H, D), pre + "linear1.bias": (H,), pre + "linear2.weight": (D, H),
                   pre + "linear2.bias": (D,), pre + "norm1.weight": (D,), pre + "norm1.bias": (D,),
                   pre + "norm2.weight": (D,), pre + "norm2.bias": (D,)})
    sh.update({"decoder.weight": (S, D), "decoder.bias": (S,)})
    return sh


def flat_tracker_params(shapes, device="cuda", init=None):
    """One flat fp32 buffer + named views (so a single Adam launch covers every tracker tensor)."""
    import numpy as np
    total = sum(int(np.prod(v)) for v in shapes.values())
    flat = torch.zeros(total, dtype=torch.float32, device=device)
    views, off = {}, 0
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        views[k] = flat[off:off + n].view(shp)
        off += n
    if init is not None:
        for k, t in init.items():
            if k in views:
                views[k].copy_(t.to(device=device, dtype=torch.float32).reshape(views[k].shape))
    return flat, views


class DeviceTracker:
    """KV-cached tracker state for B envs.  `params` maps reference state_dict names to device tensors."""

    def __init__(self, params: Dict[str, torch.Tensor], n_users, n_items, n_env, max_turn, *, dim_model=32,
                 dim_state=20, nhead=4, d_hid=128, nlayers=2, device="cuda", dropout_p=0.0):
        self.device = torch.device(device)
        self.cfg = abi.TrackerCfg(n_users=n_users, n_items=n_items, dim_model=dim_model, dim_state=dim_state,
                                  nhead=nhead, d_hid=d_hid, nlayers=nlayers, max_len=max_turn + 1, n_env=n_env,
                                  dropout_p=float(dropout_p), drop_env_base=0, dropout_seed=0)
        self.params = params
        pe = params.get("pos_encoder.pe")
        if pe is None:
            pe = positional_encoding(max_turn + 1, dim_model)
        self.pe = pe.reshape(max_turn + 1, dim_model).to(self.device, torch.float32).contiguous()
        self.nlayers = nlayers
        self.w = weights_struct(params, self.pe, nlayers)
        L, B, D = max_turn + 1, n_env, dim_model
        dev = self.device
        self.x_hist = torch.zeros((B, L, D), dtype=torch.float32, device=dev)
        self.kcache = torch.zeros((nlayers, B, L, D), dtype=torch.float32, device=dev)
        self.vcache = torch.zeros((nlayers, B, L, D), dtype=torch.float32, device=dev)
        self.len = torch.zeros(B, dtype=torch.int32, device=dev)
        self._bws = None       # workspace of backward() / prefix_states(), allocated on first use
        self.st = abi.TrackerState(x_hist=self.x_hist.data_ptr(), kcache=self.kcache.data_ptr(),
                                   vcache=self.vcache.data_ptr(), len=self.len.data_ptr())
        self._lib = abi.lib()
        self.dim_state = dim_state

    def set_dropout(self, p: float):
        """Dropout probability of the five sites (0 = off: the mode of every reference-recorded fixture)."""
        assert 0.0 <= p < 1.0
        self.cfg.dropout_p = float(p)

    def set_dropout_key(self, seed: int, tag: int = 0, env_base: int = 0):
        """Mask key of the rollout about to start: masks are a pure function of (key, env_base + env, position, layer, site,
        element); the backward of THIS rollout's buffer regenerates them from the same key.  Every rank of a job uses the same
        (seed, tag) and its own env_base = rank * n_env, so masks are defined per global env."""
        mix = (int(seed) * 0x9E3779B97F4A7C15 + (int(tag) + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        mix ^= mix >> 29
        self.cfg.dropout_seed = mix
        self.cfg.drop_env_base = int(env_base)

    def refresh_weights(self):
        self.w = weights_struct(self.params, self.pe, self.nlayers)

    # ---- training side: gradient through the stored obs + Adam ------------------------------------------------
    def enable_training(self, flat_params: torch.Tensor, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        """`flat_params` must be the flat buffer self.params' tensors are views of (flat_tracker_params)."""
        shapes = {k: tuple(v.shape) for k, v in self.params.items() if k != "pos_encoder.pe"}
        self.flat = flat_params
        self.flat_grad, self.grad_views = flat_tracker_params(shapes, device=self.device)
        assert self.flat_grad.numel() == flat_params.numel()
        self.g = weights_struct({**self.grad_views}, self.pe, self.nlayers)  # same field layout, pe unused
        self.adam_m = torch.zeros_like(flat_params)
        self.adam_v = torch.zeros_like(flat_params)
        self.adam_steps = 0
        self.lr, self.betas, self.adam_eps = lr, betas, eps
        self._bws = None

    def backward(self, users, traj, row_env, row_t, offsets, lens, n_rows, dstate, x_hist=None, drop_env_base=0, last_rows_only=False):
        """d loss / d tracker params from d loss / d obs (dstate [T+1,B,S]); fills self.flat_grad.
        x_hist: stored input slots [B', max_len, D] when the rows come from a gathered (multi-rank) buffer whose
        env count B' = traj.B differs from this tracker's own n_env.
        last_rows_only: dstate is [B', S] = the gradient of every env's LAST row, all other rows carry none (cirs_tracker_backward_last)."""
        users = users.to(self.device, torch.int32).contiguous()
        cfg, st = self.cfg, self.st
        if x_hist is not None:
            cfg = abi.TrackerCfg.from_buffer_copy(self.cfg)
            cfg.n_env = x_hist.shape[0]
            cfg.drop_env_base = int(drop_env_base)      # rows of a gathered buffer carry GLOBAL env ids already (0); redraw.py: the rollout's base
            st = abi.TrackerState(x_hist=x_hist.data_ptr(), kcache=self.kcache.data_ptr(), vcache=self.vcache.data_ptr(),
                                  len=self.len.data_ptr())
        need = self._lib.cirs_tracker_backward_workspace_bytes(C.byref(cfg), max(n_rows, cfg.n_env) if last_rows_only else n_rows)
        if self._bws is None or self._bws.numel() < need:
            self._bws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if last_rows_only:
            assert tuple(dstate.shape) == (cfg.n_env, self.dim_state) and dstate.is_contiguous()
            abi.check(self._lib.cirs_tracker_backward_last(
                C.byref(cfg), C.byref(self.w), C.byref(st), users.data_ptr(), traj.act.data_ptr(),
                traj.rew.data_ptr(), row_env.data_ptr(), row_t.data_ptr(), offsets.data_ptr(), lens.data_ptr(), n_rows,
                dstate.data_ptr(), C.byref(self.g), self._bws.data_ptr(), self._bws.numel(), self._stream()),
                "cirs_tracker_backward_last")
            return
        abi.check(self._lib.cirs_tracker_backward(
            C.byref(cfg), C.byref(self.w), C.byref(st), users.data_ptr(), traj.act.data_ptr(),
            traj.rew.data_ptr(), row_env.data_ptr(), row_t.data_ptr(), offsets.data_ptr(), lens.data_ptr(), n_rows,
            dstate.data_ptr(), C.byref(self.g), self._bws.data_ptr(), self._bws.numel(), self._stream()),
            "cirs_tracker_backward")

    def prefix_states(self, row_env, row_t, offsets, lens, n_rows, out, out_stride=None):
        """One causal pass over the rows (env b, positions 0 .. lens[b]-1) from the stored input slots under the CURRENT dropout key; the state
        of every env's last row -> out [B, S] (cirs_tracker_prefix_states: the forward half of backward())."""
        need = self._lib.cirs_tracker_backward_workspace_bytes(C.byref(self.cfg), n_rows)
        if self._bws is None or self._bws.numel() < need:
            self._bws = torch.empty(need, dtype=torch.uint8, device=self.device)
        abi.check(self._lib.cirs_tracker_prefix_states(
            C.byref(self.cfg), C.byref(self.w), C.byref(self.st), row_env.data_ptr(), row_t.data_ptr(), offsets.data_ptr(), lens.data_ptr(),
            n_rows, out.data_ptr(), self.dim_state if out_stride is None else out_stride, self._bws.data_ptr(), self._bws.numel(), self._stream()),
            "cirs_tracker_prefix_states")

    def reserve_backward(self, max_rows):
        need = self._lib.cirs_tracker_backward_workspace_bytes(C.byref(self.cfg), max_rows)
        if self._bws is None or self._bws.numel() < need:
            self._bws = torch.empty(need, dtype=torch.uint8, device=self.device)

    def adam_update(self):
        """optim_state.step(): one torch.optim.Adam step over every tracker tensor (ppo.py:235)."""
        abi.check(self._lib.cirs_adam_step(self.flat.data_ptr(), self.flat_grad.data_ptr(), self.adam_m.data_ptr(),
                                           self.adam_v.data_ptr(), self.flat.numel(), self.adam_steps, 1, self.lr,
                                           self.betas[0], self.betas[1], self.adam_eps, None, 0, self._stream()),
                  "cirs_adam_step")
        self.adam_steps += 1

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def reset(self):
        self.len.zero_()

    def init(self, users, env_ids=None, out=None, out_stride=None):
        users = users.to(self.device, torch.int32).contiguous()
        n = users.numel()
        ids = None if env_ids is None else env_ids.to(self.device, torch.int32).contiguous()
        if out is None:
            out = torch.empty((n, self.dim_state), dtype=torch.float32, device=self.device)
            out_stride = self.dim_state
        abi.check(self._lib.cirs_tracker_init(C.byref(self.cfg), C.byref(self.w), C.byref(self.st), users.data_ptr(),
                                              abi.ptr(ids), n, out.data_ptr(), out_stride, self._stream()),
                  "cirs_tracker_init")
        return out

    def step(self, items, rew, env_ids=None, skip=None, out=None, out_stride=None):
        items = items.to(self.device, torch.int64).contiguous()
        rew = rew.to(self.device, torch.float64).contiguous()
        n = items.numel()
        ids = None if env_ids is None else env_ids.to(self.device, torch.int32).contiguous()
        if out is None:
            out = torch.empty((n, self.dim_state), dtype=torch.float32, device=self.device)
            out_stride = self.dim_state
        abi.check(self._lib.cirs_tracker_step(C.byref(self.cfg), C.byref(self.w), C.byref(self.st), items.data_ptr(),
                                              rew.data_ptr(), abi.ptr(ids), abi.ptr(skip), n, out.data_ptr(),
                                              out_stride, self._stream()), "cirs_tracker_step")
        return out
