"""Exact-redraw dropout mode of the state tracker (reference core/state_tracker.py:170-186,243-246).

The reference never switches the tracker to eval(): every `build_state` call re-runs the causal transformer over the WHOLE prefix
`data[:len]` in training mode, i.e. with FRESH dropout masks at every position, layer and site -- the state s_t the policy sees (and
the graph PPO back-propagates through) belongs to the masks of call t alone.  The production mode of this build (csrc/rng.h: masks
keyed by position, K/V-cached decode) keeps a position's masks for the rest of the episode instead: same marginal distribution of
every state, different joint distribution over an episode.  This module is the reference's procedure as a tested OPTION, so that the
two can be compared (tools/compare_dropout_modes.py):

  RedrawRollout.collect     per vector step t: the mask key becomes (seed, collect tag, CALL t) and ONE batched causal pass over positions 0..t of
                            every env (cirs_tracker_prefix_states: the forward half of the BPTT, 6 launches) gives s_t; the cached decode only
                            keeps writing the input slots.  O(T) launches and O(T^2) row-passes per collect -- the procedure's own arithmetic --,
                            then cirs_actor_sample and cirs_env_step as separate launches
  redraw_tracker_backward   d loss / d s_t flows through call t only: ONE cirs_tracker_backward over pseudo-envs (call c, env e) = episodes of c + 1
                            rows with the d-state at their last row; a call's masks are keyed by its pseudo-env id, so the pass regenerates them."""
import ctypes as C
from typing import Optional

import torch

from . import abi
from .rollout import DeviceRollout


def call_tag(rng_base: int, call: int = 0) -> int:
    """Mask-key tag of the collect whose sampler counters start at rng_base.  The build_state calls of one collect share the tag and differ in
    the env field of the mask counter: call t of env e draws the masks of pseudo-env t * B + e (RedrawRollout._call_state).  (`call` is kept for
    callers that want one tag per call; the rollout passes 0.)"""
    assert 0 <= int(call) < 0xFFFF
    return ((int(rng_base) & 0xFFFFFFFFFF) << 16) + int(call) + 1


class RedrawRollout(DeviceRollout):
    def _prefix_rows(self, t, B):
        """Row description of call t: every env, positions 0..t (device tensors, built once per (t, B))."""
        cache = self.__dict__.setdefault("_rows_cache", {})
        if (t, B) not in cache:
            dev = self.device
            ar = torch.arange(B, dtype=torch.int32, device=dev)
            cache[(t, B)] = (ar.repeat_interleave(t + 1).contiguous(), torch.arange(t + 1, dtype=torch.int32, device=dev).repeat(B).contiguous(),
                             (ar * (t + 1)).contiguous(), torch.full((B,), t + 1, dtype=torch.int32, device=dev))
        return cache[(t, B)]

    def _call_state(self, t, key_seed, rng_base, out):
        """s_t of build_state call t: ONE batched causal pass over positions 0..t of every env from the stored input slots with call t's masks
        (cirs_tracker_prefix_states).  Envs that finished before call t get a state nobody reads (the sampler skips them)."""
        trk = self.tracker
        B = self.env.n_env
        row_env, row_t, offsets, lens = self._prefix_rows(t, B)
        # call t's masks: the collect's key with the pseudo-env id t * B + e in place of the env id (a fresh set per call; the batched backward
        # regenerates them from the same ids)
        trk.set_dropout_key(key_seed, call_tag(rng_base, 0), self.dropout_env_base + t * getattr(self, "B_total", B))     # (B_total: envs of ALL ranks)
        trk.prefix_states(row_env, row_t, offsets, lens, B * (t + 1), out)

    def _all_call_rows(self, T, B):
        """Row lists of build_state calls 0 .. T, concatenated (call c: every env, positions 0 .. c; n_env * (c + 1) rows starting at n_env c (c + 1) / 2)
        + offsets / lens [T + 1][B]: device tensors, built once per (T, B)."""
        cache = self.__dict__.setdefault("_all_rows_cache", {})
        if (T, B) not in cache:
            parts = [self._prefix_rows(c, B) for c in range(T + 1)]
            cache[(T, B)] = tuple(torch.cat([p[k] for p in parts]).contiguous() for k in range(4))
        return cache[(T, B)]

    def collect(self, users: torch.Tensor, *, seed=0, rng_base=0, sync_every: Optional[int] = None, gumbel=None):
        """The whole collect from one call (cirs_rollout_steps_redraw, round 5): per vector step the batched prefix pass of call t, the trunk, the
        sampler's chunk masses and the fused step kernel -- the loop of collect_stepwise() without its ~18 launches and torch ops per step from Python."""
        assert gumbel is None and self.online is None and self.visited is None and self.force_length == 0, \
            "the exact-redraw option covers the plain training rollout"
        env, tr, trk = self.env, self.traj, self.tracker
        B, T, S = env.n_env, env.max_turn, trk.dim_state
        assert T < trk.cfg.max_len
        users = users.to(self.device, torch.int32)
        key_seed = seed >> 8 if self.dropout_key_from_high_bits else seed
        self._users, self._key = users, (key_seed, rng_base)
        self.reset(users)                       # (traj cleared, tracker reset + user slot, env reset; obs[0] is overwritten by call 0's pass)
        row_env, row_t, offsets, lens = self._all_call_rows(T, B)
        trk.set_dropout_key(key_seed, call_tag(rng_base, 0), self.dropout_env_base)          # -> cfg.dropout_seed: the collect's key
        trk.reserve_backward(B * (T + 1))
        rd = abi.Redraw(row_env=row_env.data_ptr(), row_t=row_t.data_ptr(), offsets=offsets.data_ptr(), lens=lens.data_ptr(),
                        dropout_seed=trk.cfg.dropout_seed, env_base0=self.dropout_env_base, env_stride=getattr(self, "B_total", B),
                        workspace=trk._bws.data_ptr(), workspace_bytes=trk._bws.numel())
        ws = self.policy.workspace(B)
        abi.check(self._lib.cirs_rollout_steps_redraw(
            C.byref(env.cfg), C.byref(env._tab), C.byref(env._st), C.byref(trk.cfg), C.byref(trk.w), C.byref(trk.st), C.byref(self.policy.cfg),
            C.byref(self.policy.w), C.byref(tr.struct), B, 0, T, seed, rng_base, C.byref(rd), ws.data_ptr(), ws.numel(), self._stream()),
            "cirs_rollout_steps_redraw")
        return env.turn.clone()

    def collect_stepwise(self, users: torch.Tensor, *, seed=0, rng_base=0):
        """The same collect one library call per stage and step (round 4's form; kept as the test's reference for collect())."""
        env, tr, trk = self.env, self.traj, self.tracker
        B, T, S = env.n_env, env.max_turn, trk.dim_state
        users = users.to(self.device, torch.int32)
        key_seed = seed >> 8 if self.dropout_key_from_high_bits else seed
        self._scratch = torch.empty((B, S), dtype=torch.float32, device=self.device)
        tr.clear()
        env.reset(users)
        self._users, self._key = users, (key_seed, rng_base)
        # the cached decode keeps writing the INPUT slots (user slot, then one slot per action: they do not depend on any mask); its own states,
        # which belong to the position-keyed masks of the production mode, are discarded
        trk.reset()
        trk.init(users, out=self._scratch, out_stride=S)
        for t in range(T):
            self._call_state(t, key_seed, rng_base, tr.obs[t])
            done = env.done.clone()
            self.policy.sample(tr.obs[t], seed=seed, rng_step=(rng_base + t) & 0xFFFFFFFF, skip=done, act_out=tr.act[t], logp_out=tr.logp[t],
                               value_out=tr.value[t])
            _, rew, dn, ctr, _ = env.step(tr.act[t].clamp(min=0) * (done == 0))
            live = done == 0
            tr.rew[t].copy_(torch.where(live, rew, torch.zeros_like(rew)))
            tr.done[t].copy_(torch.where(live, dn, torch.zeros_like(dn)))
            tr.ctr[t].copy_(ctr)
            if t + 1 < trk.cfg.max_len:
                trk.step(tr.act[t].clamp(min=0), tr.rew[t], skip=(tr.act[t] < 0).to(torch.uint8), out=self._scratch, out_stride=S)
        if T < trk.cfg.max_len:
            self._call_state(T, key_seed, rng_base, tr.obs[T])
        return env.turn.clone()


class _CallBatch:
    """act / rew of the rollout's trajectory seen by T_calls x B pseudo-envs (pseudo-env c * B + e = env e in the graph of call c)."""
    def __init__(self, tr, n_calls):
        self.act = tr.act.repeat(1, n_calls).contiguous()
        self.rew = tr.rew.repeat(1, n_calls).contiguous()


def redraw_tracker_backward(rollout: RedrawRollout, row_env, row_t, offsets, lens, n_rows, dstate, lens_host=None, users=None, traj=None, x_hist=None):
    """Gradients of the tracker parameters under the exact-redraw procedure: d loss / d s_t flows through the graph of call t alone (call t's
    masks, positions 0..t of the envs alive at t).  All calls run as ONE backward pass: call c of env e is the pseudo-env c * B + e -- an episode
    of c + 1 rows over env e's input slots whose only d-state sits at its last row --, and because a call's masks are keyed by exactly that
    pseudo-env id (RedrawRollout._call_state), the pass regenerates every call's masks by itself.  sum_c (c + 1) live(c) rows instead of one pass
    per call: the same row-passes, 27 launches instead of 27 per call, one ordered embedding scatter.  Leaves the sum in tracker.flat_grad.
    (row_env / row_t / offsets / n_rows describe the whole buffer and are not needed; the lengths on the host -- lens_host, or one read-back -- only give the row count; the row lists are built on the device.)
    Memory: the call batch holds B T (T + 1) / 2 rows of backward workspace (~2.3 KB per row), a d-state tensor of (T + 1) C B S floats and C copies of
    the input slots: 1.3 GB at C3 (B = 1024, T = 30), 10.5 GB for the gathered buffer of 8 ranks (replicated learner, B = 8192) -- of 288 GB; the tensors are
    rebuilt per update (0.1 ms of fills at C3 next to ~5 ms of kernels)."""
    import numpy as np
    # several ranks (replicated learner): `lens`, `dstate`, `users`, `traj`, `x_hist` describe the GATHERED buffer of all B_total envs; env ids are
    # global (rank * n_env + e), which is what the rollouts keyed their masks with
    trk, tr = rollout.tracker, (rollout.traj if traj is None else traj)
    users = rollout._users if users is None else users.to(dstate.device, torch.int32)
    x_hist = trk.x_hist if x_hist is None else x_hist
    key_seed, rng_base = rollout._key
    dev = dstate.device
    lens_h = (lens.detach().cpu().numpy() if lens_host is None else np.asarray(lens_host)).astype(np.int64)
    B = lens_h.shape[0]
    C_ = int(lens_h.max())                                          # calls 0 .. C_-1 carry a gradient (s_t with t < len)
    n_q = int(sum((c + 1) * int((lens_h > c).sum()) for c in range(C_)))
    T = tr.act.shape[0]
    trk.reserve_backward(B * T * (T + 1) // 2)                      # the largest call batch (every env alive at every call): one allocation per run
    lens_d = lens.to(dev, torch.int32)
    call = torch.arange(C_, dtype=torch.int32, device=dev).repeat_interleave(B)
    lens_q = torch.where(lens_d.repeat(C_) > call, call + 1, torch.zeros_like(call))          # [C_ * B]
    offs_q = (torch.cumsum(lens_q, 0, dtype=torch.int32) - lens_q).contiguous()
    env_q = torch.repeat_interleave(torch.arange(C_ * B, dtype=torch.int32, device=dev), lens_q.long(), output_size=n_q)
    pos_q = torch.arange(n_q, dtype=torch.int32, device=dev) - torch.repeat_interleave(offs_q, lens_q.long(), output_size=n_q)
    up = lambda a: a.to(torch.int32).contiguous()      # noqa: E731
    base = rollout.dropout_env_base if B == rollout.env.n_env else 0          # (gathered buffer: the ids are global already)
    trk.set_dropout_key(key_seed, call_tag(rng_base, 0), base)
    # d s_c of env e = the gradient of the LAST row (position c) of pseudo-env (c, e) and the only one that pseudo-env carries: the last-row pass
    # (cirs_tracker_backward_last: the top layer on one row per pseudo-env); dstate[:C_] viewed as [C_ * B, S] is that gradient, pseudo-env-major
    trk.backward(users.repeat(C_), _CallBatch(tr, C_), up(env_q), up(pos_q), up(offs_q), up(lens_q), n_q,
                 dstate[:C_].reshape(C_ * B, dstate.shape[2]).contiguous(), x_hist=x_hist.repeat(C_, 1, 1).contiguous(), drop_env_base=base,
                 last_rows_only=True)
