"""Collector (reference core/collector.py:25-367) on the fused device rollout.

Keeps the constructor and `collect(n_episode=...)` contract incl. the result dict (collector.py:343-362), the fresh
buffer per collect (:113-121), `remove_recommended_ids` and `force_length` (:253-258).  The per-step Python loop
(policy -> env.step -> preprocess_fn -> buffer.add) is one call to cirs_rollout_steps; the replay buffer is filled
from the device trajectory afterwards.  Like the reference's scripts it requires n_episode == env_num (finished envs
are dropped, never reset: SURVEY Q4)."""
import time
from typing import Any, Callable, Dict, Optional

import numpy as np
import torch

from cirs_hip.rollout import DeviceRollout
from tianshou.data import Batch, VectorReplayBuffer


def result_from_trajectory(traj, lengths: np.ndarray, offsets: np.ndarray) -> Dict[str, Any]:
    """Collector.collect's result dict (reference core/collector.py:343-362) from a finished time-major trajectory.
    `rews / lens / idxs` are in the reference's order — episodes as they complete: by final step, env id ascending inside a
    step (:280-288) — and every episode reward is the running float64 sum in step order (base.py:171 `self._ep_rew += rew`),
    so the dict is bit-identical to the reference's for the same transitions."""
    lengths = np.asarray(lengths, dtype=int)
    rew = traj.rew.cpu().numpy()
    ep_rew = np.zeros(len(lengths), dtype=np.float64)
    for t in range(int(lengths.max(initial=0))):
        live = lengths > t
        ep_rew[live] += rew[t, :len(lengths)][live]
    order = np.argsort(lengths, kind="stable")
    rews, lens, idxs = ep_rew[order], lengths[order], np.asarray(offsets)[order]
    n_ep = len(lengths)
    if n_ep > 0:
        rew_mean, rew_std, len_mean, len_std = rews.mean(), rews.std(), lens.mean(), lens.std()
    else:
        rew_mean = rew_std = len_mean = len_std = 0
    return {"n/ep": n_ep, "n/st": int(lengths.sum()), "rews": rews, "lens": lens, "idxs": idxs, "rew": rew_mean, "len": len_mean,
            "rew_std": rew_std, "len_std": len_std}


class Collector:
    def __new__(cls, policy=None, env=None, *args, **kwargs):
        """Host vector envs (VirtualTB-v0: BASELINE configs[0], CPU plumbing) are collected by the per-step loop of core.host_rl."""
        if cls is Collector and getattr(env, "host_mode", False):
            from core.host_rl import HostCollector
            return HostCollector(policy, env, *args, **kwargs)
        return super().__new__(cls)

    def __init__(self, policy, env, buffer: Optional[VectorReplayBuffer] = None, preprocess_fn: Optional[Callable[..., Any]] = None,
                 exploration_noise: bool = False, remove_recommended_ids=False, force_length=0):
        assert hasattr(env, "__len__"), "pass a tianshou.env.DummyVectorEnv"
        assert preprocess_fn is not None and hasattr(preprocess_fn, "__self__"), \
            "preprocess_fn must be StateTrackerTransformer.build_state (CIRS-RL-kuaishou.py:291)"
        assert not exploration_noise
        self.env, self.env_num = env, len(env)
        self.policy = policy
        self.preprocess_fn = preprocess_fn
        self.tracker = preprocess_fn.__self__
        self.remove_recommended_ids = remove_recommended_ids
        self.force_length = force_length
        self._action_space = env.action_space
        self.buffer = buffer if buffer is not None else VectorReplayBuffer(self.env_num, self.env_num)
        assert self.buffer.buffer_num >= self.env_num
        self._rollout: Optional[DeviceRollout] = None
        self._collect_count = 0
        # sampler key of this collector: (policy seed, collector stream).  Train / FB / NX_0 / NX_k collectors draw independent
        # noise (with one shared key they would all see the same Gumbel draw per (env, step, item) in every epoch)
        self.stream_id = getattr(policy, "_n_collectors", 0)     # the policy's first collector (train) keeps the bare seed
        policy.__dict__["_n_collectors"] = self.stream_id + 1
        self.data = Batch()
        if hasattr(policy, "_tracker") and env.workers[0].simulated:
            # the gradient through the stored obs of the TRAINING buffer goes to this tracker (ppo.py:215)
            policy.__dict__["_tracker"] = self.tracker
            policy.__dict__["_train_n_env"] = self.env_num
        self.reset_stat()

    def reset_stat(self):
        self.collect_step, self.collect_episode, self.collect_time = 0, 0, 0.0

    def reset_buffer(self, keep_statistics: bool = False):
        self.buffer = VectorReplayBuffer(self.buffer.maxsize, self.buffer.buffer_num)  # a brand-new buffer per collect

    def reset_env(self):
        pass  # env reset + tracker init happen at the start of the fused rollout

    def reset(self):
        self.reset_env()
        self.reset_buffer()
        self.reset_stat()

    def _assign_buffer(self, buffer):
        self.buffer = buffer

    def _reset_state(self, id):
        pass

    def _get_rollout(self) -> DeviceRollout:
        if self._rollout is None:
            dev_env = self.env.device_env()
            trk = self.tracker.engine(self.env_num, owner=self)
            self._rollout = DeviceRollout(dev_env, trk, self.policy.device_policy(), remove_recommended_ids=self.remove_recommended_ids,
                                          force_length=self.force_length)
        return self._rollout

    def sampler_seed(self):
        return ((int(self.policy.seed) * 0x9E3779B1) ^ (self.stream_id * 0x85EBCA6B)) & 0x7FFFFFFF if self.stream_id else int(self.policy.seed)

    def collect(self, n_step: Optional[int] = None, n_episode: Optional[int] = None, random: bool = False, render=None,
                no_grad: bool = True, users=None, gumbel=None) -> Dict[str, Any]:
        """users / gumbel: teacher forcing for parity tests -- the users the envs are reset to and the sampler noise
        [max_turn, env_num, n_items] (g = -log q of the reference's Categorical.sample race)."""
        assert n_step is None and n_episode is not None, "the CIRS scripts collect whole episodes (n_episode)"
        assert n_episode == self.env_num, "n_episode must equal the number of envs (finished envs are not reset, SURVEY Q4)"
        if random:
            return self._collect_stepwise(n_episode, users)
        start = time.time()
        ro = self._get_rollout()
        self.reset_buffer()
        if users is None:
            users = self.env.draw_users(self.env_num)
        users_t = torch.as_tensor(np.asarray(users))
        T = ro.env.max_turn
        lengths = ro.collect(users_t, seed=self.sampler_seed(), rng_base=(self._collect_count * T) & 0xFFFFFFFF, gumbel=gumbel).cpu().numpy()
        self._collect_count += 1
        self.buffer.fill_from_trajectory(ro.traj, lengths)
        self.buffer._rollout, self.buffer._users = ro, users_t  # policy.update() consumes them with the buffer
        res = result_from_trajectory(ro.traj, lengths, self.buffer._offset[:self.env_num])
        self.collect_step += res["n/st"]
        self.collect_episode += res["n/ep"]
        self.collect_time += max(time.time() - start, 1e-9)
        return res

    def _collect_stepwise(self, n_episode, users=None) -> Dict[str, Any]:
        """`collect(random=True)` (reference core/collector.py:225-227): actions are drawn from the envs' action spaces instead of
        the policy, everything else is the reference's loop one vector step at a time through the per-step protocol -- env.step
        (one cirs_env_step launch), preprocess_fn = StateTracker.build_state (one cirs_tracker_step launch), buffer.add -- with
        finished envs dropped from the ready set and the result dict assembled from add()'s episode accounting (:272-362)."""
        start = time.time()
        self.reset_buffer()
        n = self.env_num
        self.preprocess_fn(dim_batch=n, reset=True)
        obs = self.env.reset(users=users)
        state = self.preprocess_fn(obs=obs, env_id=np.arange(n))["obs"]
        ready = np.arange(n)
        step_count, episode_count, cnt_loop = 0, 0, 0
        ep_rews, ep_lens, ep_idxs = [], [], []
        while True:
            act = np.array([int(np.asarray(self._action_space[i].sample()).reshape(-1)[0]) for i in ready], dtype=np.int64)
            obs_next, rew, done, info = self.env.step(self.policy.map_action(act), ready)
            cnt_loop += 1
            if self.force_length > 0:
                done = np.full_like(done, cnt_loop >= self.force_length)
            nxt = self.preprocess_fn(obs_next=obs_next, rew=rew, done=done, info=info, policy=None, env_id=ready)["obs_next"]
            ptr, e_rew, e_len, e_idx = self.buffer.add(Batch(obs=state, act=act, rew=rew, done=done, obs_next=nxt,
                                                             info=Batch(env_id=ready.copy())), buffer_ids=ready)
            step_count += len(ready)
            if np.any(done):
                fin = np.where(done)[0]
                episode_count += len(fin)
                ep_lens.append(e_len[fin]); ep_rews.append(e_rew[fin]); ep_idxs.append(e_idx[fin])
                keep = ~done
                ready, nxt = ready[keep], nxt[torch.as_tensor(keep, device=nxt.device)]
            state = nxt
            if episode_count >= n_episode or len(ready) == 0:
                break
        rews, lens, idxs = np.concatenate(ep_rews), np.concatenate(ep_lens), np.concatenate(ep_idxs)
        self.collect_step += step_count
        self.collect_episode += episode_count
        self.collect_time += max(time.time() - start, 1e-9)
        return {"n/ep": episode_count, "n/st": step_count, "rews": rews, "lens": lens, "idxs": idxs, "rew": rews.mean(), "len": lens.mean(),
                "rew_std": rews.std(), "len_std": lens.std()}
