"""The user model as a static recommendation policy (csrc/static_policy.hip): item selection from catalogue scores and the
lock-step evaluation rollout.  Host-side counterpart of reference core/user_model.py:254-348 (recommend_k_item) and
evaluation.py:79-151 (interactive_evaluation)."""
import ctypes as C
from typing import Optional

import torch

from . import abi
from .env import DeviceEnv


def select_items(scores: torch.Tensor, *, softmax: bool, bonus: Optional[torch.Tensor] = None, visited: Optional[torch.Tensor] = None,
                 skip: Optional[torch.Tensor] = None, epsilon: float = 0.0, gumbel: Optional[torch.Tensor] = None, seed: int = 0,
                 rng_step: int = 0):
    """scores [n, I] float32 on the device -> (act [n] int64, value [n] float32)."""
    assert scores.is_cuda and scores.dtype == torch.float32 and scores.dim() == 2 and scores.stride(1) == 1
    n, I = scores.shape
    dev = scores.device
    act = torch.empty(n, dtype=torch.int64, device=dev)
    val = torch.empty(n, dtype=torch.float32, device=dev)
    f32 = lambda t: None if t is None else t.to(dev, torch.float32).contiguous()
    bonus, gumbel = f32(bonus), f32(gumbel)
    if visited is not None:
        visited = visited.to(dev).contiguous()
        assert visited.dtype == torch.int32 and visited.shape == (n, (I + 31) // 32)
    if skip is not None:
        skip = skip.to(dev, torch.uint8).contiguous()
    abi.check(abi.lib().cirs_select_items(scores.data_ptr(), scores.stride(0), n, I, int(bool(softmax)), abi.ptr(bonus), abi.ptr(visited),
                                          abi.ptr(skip), float(epsilon), abi.ptr(gumbel), int(seed), int(rng_step) & 0xFFFFFFFF,
                                          act.data_ptr(), val.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "cirs_select_items")
    return act, val


class StaticRollout:
    """n_env trajectories of `interactive_evaluation` in lock-step on a (non-simulated) DeviceEnv."""

    def __init__(self, env: DeviceEnv):
        self.env = env
        dev, B, T = env.device, env.n_env, env.max_turn
        self.act = torch.full((T, B), -1, dtype=torch.int64, device=dev)
        self.rew = torch.zeros((T, B), dtype=torch.float64, device=dev)
        self.done = torch.zeros((T, B), dtype=torch.uint8, device=dev)
        self.value = torch.zeros((T, B), dtype=torch.float32, device=dev)   # reward_pred of the chosen item
        self.ctr = torch.zeros((T, B), dtype=torch.float64, device=dev)
        self._traj = abi.Traj(obs=None, act=self.act.data_ptr(), rew=self.rew.data_ptr(), done=self.done.data_ptr(), logp=None,
                              value=self.value.data_ptr(), ctr=self.ctr.data_ptr())
        self._scratch = torch.zeros(B, dtype=torch.int64, device=dev)
        self._lib = abi.lib()

    def run(self, users: torch.Tensor, scores: torch.Tensor, *, softmax=False, epsilon=0.0, seed=0, rng_base=0, remove_recommended=False,
            force_length=0, bonus: Optional[torch.Tensor] = None, n_steps: Optional[int] = None):
        env = self.env
        B, I = env.n_env, env.tables.n_items
        assert scores.shape == (B, I) and scores.dtype == torch.float32 and scores.is_cuda and scores.stride(1) == 1
        self.act.fill_(-1); self.done.zero_(); self.rew.zero_(); self.value.zero_()
        env.reset(users)
        visited = torch.zeros((B, (I + 31) // 32), dtype=torch.int32, device=env.device) if remove_recommended else None
        bonus = None if bonus is None else bonus.to(env.device, torch.float32).contiguous()
        T = env.max_turn if n_steps is None else n_steps
        abi.check(self._lib.cirs_rollout_static(C.byref(env.cfg), C.byref(env._tab), C.byref(env._st), scores.data_ptr(), scores.stride(0),
                                                abi.ptr(bonus), C.byref(self._traj), B, 0, T, int(bool(softmax)), float(epsilon), int(seed),
                                                int(rng_base) & 0xFFFFFFFF, abi.ptr(visited), int(force_length), self._scratch.data_ptr(),
                                                torch.cuda.current_stream(env.device).cuda_stream), "cirs_rollout_static")
        return env.turn.clone()

    # ---- one vector step at a time (UCB: the exploration bonus changes after every recommendation) -----------------------------
    def begin(self, users: torch.Tensor, remove_recommended=False):
        env = self.env
        B, I = env.n_env, env.tables.n_items
        self.act.fill_(-1); self.done.zero_(); self.rew.zero_(); self.value.zero_()
        env.reset(users)
        self._visited = torch.zeros((B, (I + 31) // 32), dtype=torch.int32, device=env.device) if remove_recommended else None

    def step(self, t: int, scores: torch.Tensor, *, softmax=False, epsilon=0.0, seed=0, rng_base=0, force_length=0,
             bonus: Optional[torch.Tensor] = None):
        """Vector step t of the trajectories opened by begin(): select (scores + bonus) -> mark visited -> env step."""
        env = self.env
        B, I = env.n_env, env.tables.n_items
        assert scores.shape == (B, I) and scores.dtype == torch.float32 and scores.is_cuda and scores.stride(1) == 1
        bonus = None if bonus is None else bonus.to(env.device, torch.float32).contiguous()
        abi.check(self._lib.cirs_rollout_static(C.byref(env.cfg), C.byref(env._tab), C.byref(env._st), scores.data_ptr(), scores.stride(0),
                                                abi.ptr(bonus), C.byref(self._traj), B, int(t), int(t) + 1, int(bool(softmax)), float(epsilon),
                                                int(seed), int(rng_base) & 0xFFFFFFFF, abi.ptr(self._visited), int(force_length),
                                                self._scratch.data_ptr(), torch.cuda.current_stream(env.device).cuda_stream),
                  "cirs_rollout_static")

