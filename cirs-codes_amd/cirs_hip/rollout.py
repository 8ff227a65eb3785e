"""Device-resident Collector loop (csrc/rollout.hip): whole episodes are collected without a host round trip per step.

Host-side counterpart of reference core/collector.py:147-367 for the case the reference scripts use
(n_episode == env_num, finished envs are dropped).  The time-major trajectory tensors double as the
replay buffer (VectorReplayBuffer order = env-major concatenation of the per-env episodes).
"""
import ctypes as C
import os
from typing import Optional

import torch

from . import abi
from .env import DeviceEnv
from .policy import DevicePolicy
from .tracker import DeviceTracker


class Trajectory:
    def __init__(self, n_env, max_turn, dim_state, device):
        T, B, S = max_turn, n_env, dim_state
        self.T, self.B, self.S = T, B, S
        self.obs = torch.zeros((T + 1, B, S), dtype=torch.float32, device=device)
        self.act = torch.full((T, B), -1, dtype=torch.int64, device=device)
        self.rew = torch.zeros((T, B), dtype=torch.float64, device=device)
        self.done = torch.zeros((T, B), dtype=torch.uint8, device=device)
        self.logp = torch.zeros((T, B), dtype=torch.float32, device=device)
        self.value = torch.zeros((T, B), dtype=torch.float32, device=device)
        self.ctr = torch.zeros((T, B), dtype=torch.float64, device=device)
        self.struct = abi.Traj(obs=self.obs.data_ptr(), act=self.act.data_ptr(), rew=self.rew.data_ptr(),
                               done=self.done.data_ptr(), logp=self.logp.data_ptr(), value=self.value.data_ptr(),
                               ctr=self.ctr.data_ptr())

    def clear(self):
        self.act.fill_(-1)
        self.done.zero_()


class OnlineReward:
    """Scores the chosen (user, item) pairs with the DeepFM user model inside the rollout loop instead of reading a
    precomputed U x I table (reference simulated_env.py:88-98 keeps this variant commented out; BASELINE configs[4]
    needs it: a 10^6 x 10^6 float64 table does not exist).

    user_model: cirs_hip.deepfm.DeviceDeepFM; raw_uid [U] / raw_pid [I] int64 (lbe_*.classes_); item_feats [I,4] int32
    (ids shifted by +1, 0 = padding); item_dur [I] float32; minmax: (min, max) of the raw scores used by
    compute_normed_reward (kuaishouEnv.py:139-143)."""

    def __init__(self, user_model, raw_uid, raw_pid, item_feats, item_dur, minmax, n_env, device="cuda"):
        dev = torch.device(device)
        self.user_model = user_model
        self.raw_uid = torch.as_tensor(raw_uid, dtype=torch.int64).to(dev).contiguous()
        self.raw_pid = torch.as_tensor(raw_pid, dtype=torch.int64).to(dev).contiguous()
        self.item_feats = torch.as_tensor(item_feats, dtype=torch.int32).to(dev).contiguous()
        self.item_dur = torch.as_tensor(item_dur, dtype=torch.float32).to(dev).contiguous()
        assert self.item_feats.shape == (self.raw_pid.numel(), 4)
        self.minmax = torch.as_tensor(minmax, dtype=torch.float32).to(dev).contiguous()
        B = int(n_env)
        self.uid_buf = torch.zeros(B, dtype=torch.int64, device=dev)
        self.pid_buf = torch.zeros(B, dtype=torch.int64, device=dev)
        self.feat_buf = torch.zeros((B, 4), dtype=torch.int32, device=dev)
        self.dur_buf = torch.zeros(B, dtype=torch.float32, device=dev)
        self.pred_buf = torch.zeros(B, dtype=torch.float32, device=dev)
        self.struct = abi.OnlineReward(
            cfg=C.pointer(user_model.cfg), w=C.pointer(user_model.w), raw_uid=self.raw_uid.data_ptr(),
            raw_pid=self.raw_pid.data_ptr(), item_feats=self.item_feats.data_ptr(), item_dur=self.item_dur.data_ptr(),
            pred_minmax=self.minmax.data_ptr(), uid_buf=self.uid_buf.data_ptr(), pid_buf=self.pid_buf.data_ptr(),
            feat_buf=self.feat_buf.data_ptr(), dur_buf=self.dur_buf.data_ptr(), pred_buf=self.pred_buf.data_ptr())


class DeviceRollout:
    def __init__(self, env: DeviceEnv, tracker: DeviceTracker, policy: DevicePolicy, *, remove_recommended_ids=False,
                 force_length=0, online: Optional[OnlineReward] = None):
        assert env.n_env == tracker.cfg.n_env
        self.env, self.tracker, self.policy = env, tracker, policy
        self.device = env.device
        self.traj = Trajectory(env.n_env, env.max_turn, tracker.dim_state, self.device)
        self.remove_recommended_ids = remove_recommended_ids
        self.force_length = int(force_length)
        self.online = online
        if online is None and env.cfg.simulated:
            assert env.tables.normed_mat is not None, "simulated env needs normed_mat or an OnlineReward scorer"
        words = (policy.n_items + 31) // 32
        self.visited = torch.zeros((env.n_env, words), dtype=torch.int32, device=self.device) if remove_recommended_ids else None
        self._lib = abi.lib()
        # dropout masks are defined per GLOBAL env of a multi-rank job: key shared by the ranks, env ids offset per rank
        self.dropout_env_base = 0
        self.dropout_key_from_high_bits = False   # CirsEngine packs (seed << 8) + rank into the sampler seed

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def reset(self, users: torch.Tensor):
        """Collector.reset_env (collector.py:123-134): tracker reset, env.reset, preprocess_fn(obs=...)."""
        self.traj.clear()
        self.tracker.reset()
        if self.visited is not None:
            self.visited.zero_()
        self.env.reset(users)
        B, S = self.env.n_env, self.tracker.dim_state
        self.tracker.init(users, out=self.traj.obs[0], out_stride=S)

    def run_steps(self, t_begin, t_end, seed, rng_base, gumbel=None):
        ws = self.policy.workspace(self.env.n_env)
        if gumbel is not None:      # harness-supplied sampler noise (parity fixtures recorded from the reference)
            assert self.online is None and gumbel.dtype == torch.float32 and gumbel.is_contiguous()
            assert tuple(gumbel.shape) == (self.env.max_turn, self.env.n_env, self.policy.n_items)
            abi.check(self._lib.cirs_rollout_steps_noise(
                C.byref(self.env.cfg), C.byref(self.env._tab), C.byref(self.env._st), C.byref(self.tracker.cfg),
                C.byref(self.tracker.w), C.byref(self.tracker.st), C.byref(self.policy.cfg), C.byref(self.policy.w),
                C.byref(self.traj.struct), self.env.n_env, t_begin, t_end, gumbel.data_ptr(), abi.ptr(self.visited),
                self.force_length, ws.data_ptr(), ws.numel(), self._stream()), "cirs_rollout_steps_noise")
            return
        if self.online is not None:
            abi.check(self._lib.cirs_rollout_steps_online(
                C.byref(self.env.cfg), C.byref(self.env._tab), C.byref(self.env._st), C.byref(self.tracker.cfg),
                C.byref(self.tracker.w), C.byref(self.tracker.st), C.byref(self.policy.cfg), C.byref(self.policy.w),
                C.byref(self.traj.struct), self.env.n_env, t_begin, t_end, seed, rng_base, abi.ptr(self.visited),
                self.force_length, C.byref(self.online.struct), ws.data_ptr(), ws.numel(), self._stream()),
                "cirs_rollout_steps_online")
            return
        abi.check(self._lib.cirs_rollout_steps(
            C.byref(self.env.cfg), C.byref(self.env._tab), C.byref(self.env._st), C.byref(self.tracker.cfg),
            C.byref(self.tracker.w), C.byref(self.tracker.st), C.byref(self.policy.cfg), C.byref(self.policy.w),
            C.byref(self.traj.struct), self.env.n_env, t_begin, t_end, seed, rng_base, abi.ptr(self.visited),
            self.force_length, ws.data_ptr(), ws.numel(), self._stream()), "cirs_rollout_steps")

    def collect(self, users: torch.Tensor, *, seed=0, rng_base=0, sync_every: Optional[int] = None, gumbel=None):
        """One `collect(n_episode = n_env)`: all envs run to the end of their episode.  Returns (n_steps, lengths).
        sync_every: poll the live-env count every that many steps to stop early (None: run all max_turn steps
        without any host sync; idle steps of finished envs are no-ops)."""
        if self.tracker.cfg.dropout_p > 0:   # fresh masks per collect (the reference draws fresh dropout noise at every call)
            self.tracker.set_dropout_key(seed >> 8 if self.dropout_key_from_high_bits else seed, rng_base, self.dropout_env_base)
        T = self.env.max_turn
        if gumbel is None and sync_every is None and self.online is None and not os.environ.get("CIRS_ROLLOUT_STEPWISE_RESET"):
            # reset_env + the whole collect from one call (cirs_rollout_collect): no clears, the tracker's first position from the packed weight image
            # with the first trunk in its launch -- the same bits as reset() + run_steps(0, T) (tests/test_gpu_rollout.py)
            users_d = users.to(self.device, torch.int32).contiguous()
            if self.visited is not None:
                self.visited.zero_()
            ws = self.policy.workspace(self.env.n_env)
            abi.check(self._lib.cirs_rollout_collect(
                C.byref(self.env.cfg), C.byref(self.env._tab), C.byref(self.env._st), C.byref(self.tracker.cfg),
                C.byref(self.tracker.w), C.byref(self.tracker.st), C.byref(self.policy.cfg), C.byref(self.policy.w),
                C.byref(self.traj.struct), self.env.n_env, users_d.data_ptr(), seed, rng_base, abi.ptr(self.visited),
                self.force_length, ws.data_ptr(), ws.numel(), self._stream()), "cirs_rollout_collect")
            self._users_keepalive = users_d
            return self.env.turn.clone()
        self.reset(users)
        if gumbel is not None:
            gumbel = torch.as_tensor(gumbel).to(self.device, torch.float32).contiguous()
            self.run_steps(0, T, seed, rng_base, gumbel=gumbel)
            self._gumbel_keep = gumbel
        elif sync_every is None:
            self.run_steps(0, T, seed, rng_base)
        else:
            t = 0
            while t < T:
                t2 = min(T, t + sync_every)
                self.run_steps(t, t2, seed, rng_base)
                t = t2
                if t < T and bool(self.env.done.all()):
                    break
        lengths = self.env.turn.clone()
        return lengths
