"""Device engine of the rollout-side policy forward (csrc/policy.hip): trunk + MFMA actor head + fused sampler."""
import ctypes as C
from typing import Dict, Optional

import torch

from . import abi

POLICY_FIELDS = [("w1", "actor.preprocess.model.model.0.weight"), ("b1", "actor.preprocess.model.model.0.bias"),
                 ("w2", "actor.preprocess.model.model.2.weight"), ("b2", "actor.preprocess.model.model.2.bias"),
                 ("wa", "actor.last.model.0.weight"), ("ba", "actor.last.model.0.bias"),
                 ("wc", "critic.last.model.0.weight"), ("bc", "critic.last.model.0.bias")]


def weights_struct(params: Dict[str, torch.Tensor]):
    w = abi.PolicyWeights()
    for f, name in POLICY_FIELDS:
        t = params[name]
        assert t.dtype == torch.float32 and t.is_contiguous(), name
        setattr(w, f, t.data_ptr())
    return w


class DevicePolicy:
    def __init__(self, params: Dict[str, torch.Tensor], n_items, *, dim_state=20, hidden=64, device="cuda"):
        self.device = torch.device(device)
        self.cfg = abi.PolicyCfg(n_items=n_items, dim_state=dim_state, hidden=hidden)
        self.params = params
        self.w = weights_struct(params)
        self._lib = abi.lib()
        self._ws = None
        self.n_items = n_items

    def refresh_weights(self):
        self.w = weights_struct(self.params)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def workspace(self, n):
        need = self._lib.cirs_policy_workspace_bytes(C.byref(self.cfg), n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def values(self, state, *, state_stride=None, n=None, value_out=None):
        """critic(state) for n rows with the current parameters (the value pass of A2CPolicy._compute_returns, a2c.py:80-86)."""
        n = state.shape[0] if n is None else n
        state_stride = state.stride(0) if state_stride is None else state_stride
        value = value_out if value_out is not None else torch.empty(n, dtype=torch.float32, device=self.device)
        ws = self.workspace(n)
        abi.check(self._lib.cirs_critic_values(C.byref(self.cfg), C.byref(self.w), state.data_ptr(), state_stride, n, value.data_ptr(),
                                               ws.data_ptr(), ws.numel(), self._stream()), "cirs_critic_values")
        return value

    def sample(self, state, *, state_stride=None, n=None, gumbel=None, seed=0, rng_step=0, env_ids=None, visited=None,
               skip=None, act_out=None, logp_out=None, value_out=None):
        if n is None:
            n = state.shape[0]
        if state_stride is None:
            state_stride = state.stride(0)
        dev = self.device
        act = act_out if act_out is not None else torch.empty(n, dtype=torch.int64, device=dev)
        logp = logp_out if logp_out is not None else torch.empty(n, dtype=torch.float32, device=dev)
        value = value_out if value_out is not None else torch.empty(n, dtype=torch.float32, device=dev)
        ws = self.workspace(n)
        abi.check(self._lib.cirs_actor_sample(C.byref(self.cfg), C.byref(self.w), state.data_ptr(), state_stride, n,
                                              abi.ptr(gumbel), seed, rng_step, abi.ptr(env_ids), abi.ptr(visited),
                                              abi.ptr(skip), act.data_ptr(), logp.data_ptr(), value.data_ptr(),
                                              ws.data_ptr(), ws.numel(), self._stream()), "cirs_actor_sample")
        return act, logp, value
