"""Device-resident batched SimulatedEnv(KuaishouEnv): tables + per-env state live in HBM, one launch per vector step.

Host-side counterpart of
  core/env/simulatedEnv/simulated_env.py (SimulatedEnv)   environments/KuaishouRec/env/kuaishouEnv.py (KuaishouEnv)
  tianshou/env/venvs.py (DummyVectorEnv's serial loop)
in the reference.  All arithmetic happens in csrc/env.hip through the C ABI (cirs_env_reset / cirs_env_step).
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import abi
from .synthetic import pack_item_cats


def _dev(x, dtype, device):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(device).contiguous()


class DeviceEnvTables:
    """Read-only tables in HBM (layout: include/cirs_hip.h cirs_env_tables)."""

    def __init__(self, mat, normed_mat, item_cats, *, dist=None, alpha_env=None, beta_env=None, device="cuda",
                 build_dist_on_device=False, stream=None, n_users=None, n_items=None):
        """mat may be None for a simulated env whose catalogue is too large for a U x I table (then pass n_users /
        n_items and leave normed_mat None: the rollout scores rewards online, cirs_online_reward)."""
        self.device = torch.device(device)
        if mat is None:
            assert n_users is not None and n_items is not None
            self.n_users, self.n_items = int(n_users), int(n_items)
            self.mat = None
        else:
            self.n_users, self.n_items = mat.shape
            self.mat = _dev(mat, torch.float64, self.device)
        self.normed_mat = None if normed_mat is None else _dev(normed_mat, torch.float64, self.device)
        packed = item_cats if np.asarray(item_cats).ndim == 1 else pack_item_cats(item_cats)
        self.item_cats = torch.as_tensor(np.ascontiguousarray(packed).view(np.int32)).to(self.device)
        self.dist = None
        if dist is not None:
            self.dist = _dev(dist, torch.float64, self.device)
        elif build_dist_on_device:
            self.dist = torch.empty((self.n_items, self.n_items), dtype=torch.float64, device=self.device)
            abi.check(abi.lib().cirs_dist_jaccard(self.item_cats.data_ptr(), self.n_items, self.dist.data_ptr(),
                                                  stream), "cirs_dist_jaccard")
        self.has_ab = alpha_env is not None
        self.alpha_env = _dev(alpha_env if self.has_ab else np.ones(self.n_users), torch.float64, self.device)
        self.beta_env = _dev(beta_env if self.has_ab else np.ones(self.n_items), torch.float64, self.device)

    def struct(self):
        return abi.EnvTables(mat=None if self.mat is None else self.mat.data_ptr(),
                             normed_mat=None if self.normed_mat is None else self.normed_mat.data_ptr(),
                             dist=None if self.dist is None else self.dist.data_ptr(),
                             item_cats=self.item_cats.data_ptr(), alpha_env=self.alpha_env.data_ptr(),
                             beta_env=self.beta_env.data_ptr())


class DeviceEnv:
    """B environments stepped by one kernel launch.  Mutable state is SoA in HBM (cirs_env_state)."""

    def __init__(self, tables: DeviceEnvTables, n_env: int, *, num_leave_compute=5, leave_threshold=1, max_turn=100,
                 tau=1.0, gamma_exposure=1.0, version="v1", r_decay=1.0, use_exposure_intervention=True,
                 simulated=True, dist_mode: Optional[int] = None):
        self.tables = tables
        self.n_env = int(n_env)
        dev = tables.device
        if dist_mode is None:
            dist_mode = 0 if tables.dist is not None else 1
        ver = {"v1": 1, "v2": 2}.get(version, version)
        self.cfg = abi.EnvCfg(n_users=tables.n_users, n_items=tables.n_items, max_turn=int(max_turn),
                              num_leave_compute=int(num_leave_compute), leave_threshold=int(leave_threshold),
                              version=int(ver), use_exposure=int(bool(use_exposure_intervention)),
                              has_ab=int(tables.has_ab), dist_mode=int(dist_mode), simulated=int(bool(simulated)),
                              tau=float(tau), gamma_exposure=float(gamma_exposure), r_decay=float(r_decay))
        self.max_turn = int(max_turn)
        B, T = self.n_env, self.max_turn
        self.user = torch.zeros(B, dtype=torch.int32, device=dev)
        self.turn = torch.zeros(B, dtype=torch.int32, device=dev)
        self.done = torch.ones(B, dtype=torch.uint8, device=dev)  # not reset yet -> inert
        self.hist_action = torch.zeros((B, T), dtype=torch.int32, device=dev)
        self.cum_reward = torch.zeros(B, dtype=torch.float64, device=dev)
        self._tab = tables.struct()
        self._st = abi.EnvState(user=self.user.data_ptr(), turn=self.turn.data_ptr(), done=self.done.data_ptr(),
                                hist_action=self.hist_action.data_ptr(), cum_reward=self.cum_reward.data_ptr())
        self._lib = abi.lib()

    @property
    def device(self):
        return self.tables.device

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def reset(self, users: torch.Tensor, env_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
        users = users.to(device=self.device, dtype=torch.int32).contiguous()
        n = users.numel()
        ids = None if env_ids is None else env_ids.to(device=self.device, dtype=torch.int32).contiguous()
        obs = torch.empty(n, dtype=torch.int64, device=self.device)
        abi.check(self._lib.cirs_env_reset(C.byref(self.cfg), C.byref(self._st), users.data_ptr(), abi.ptr(ids), n,
                                           obs.data_ptr(), self._stream()), "cirs_env_reset")
        return obs

    def step(self, actions: torch.Tensor, env_ids: Optional[torch.Tensor] = None, want_exposure=False):
        actions = actions.to(device=self.device, dtype=torch.int64).contiguous()
        n = actions.numel()
        ids = None if env_ids is None else env_ids.to(device=self.device, dtype=torch.int32).contiguous()
        dev = self.device
        obs = torch.empty(n, dtype=torch.int64, device=dev)
        rew = torch.empty(n, dtype=torch.float64, device=dev)
        done = torch.empty(n, dtype=torch.uint8, device=dev)
        ctr = torch.empty(n, dtype=torch.float64, device=dev)
        expo = torch.empty(n, dtype=torch.float64, device=dev) if want_exposure else None
        abi.check(self._lib.cirs_env_step(C.byref(self.cfg), C.byref(self._tab), C.byref(self._st),
                                          actions.data_ptr(), abi.ptr(ids), n, obs.data_ptr(), rew.data_ptr(),
                                          done.data_ptr(), ctr.data_ptr(), abi.ptr(expo), self._stream()),
                  "cirs_env_step")
        return obs, rew, done, ctr, expo
