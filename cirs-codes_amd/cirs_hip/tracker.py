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


class DeviceTracker:
    """KV-cached tracker state for B envs.  `params` maps reference state_dict names to device tensors."""

    def __init__(self, params: Dict[str, torch.Tensor], n_users, n_items, n_env, max_turn, *, dim_model=32,
                 dim_state=20, nhead=4, d_hid=128, nlayers=2, device="cuda"):
        self.device = torch.device(device)
        self.cfg = abi.TrackerCfg(n_users=n_users, n_items=n_items, dim_model=dim_model, dim_state=dim_state,
                                  nhead=nhead, d_hid=d_hid, nlayers=nlayers, max_len=max_turn + 1, n_env=n_env)
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
        self.st = abi.TrackerState(x_hist=self.x_hist.data_ptr(), kcache=self.kcache.data_ptr(),
                                   vcache=self.vcache.data_ptr(), len=self.len.data_ptr())
        self._lib = abi.lib()
        self.dim_state = dim_state

    def refresh_weights(self):
        self.w = weights_struct(self.params, self.pe, self.nlayers)

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
