"""Exact-redraw dropout mode of the state tracker (reference core/state_tracker.py:170-186,243-246).

The reference never switches the tracker to eval(): every `build_state` call re-runs the causal transformer over the WHOLE prefix
`data[:len]` in training mode, i.e. with FRESH dropout masks at every position, layer and site -- the state s_t the policy sees (and
the graph PPO back-propagates through) belongs to the masks of call t alone.  The production mode of this build (csrc/rng.h: masks
keyed by position, K/V-cached decode) keeps a position's masks for the rest of the episode instead: same marginal distribution of
every state, different joint distribution over an episode.  This module is the reference's procedure as a tested OPTION, so that the
two can be compared (tools/compare_dropout_modes.py):

  RedrawRollout.collect     per vector step t: the mask key becomes (seed, collect tag, CALL t), the K/V caches are rebuilt by replaying
                            positions 0..t through cirs_tracker_init / cirs_tracker_step (O(T^2) launches per collect instead of O(T):
                            an option for studies, not the benchmark path), then cirs_actor_sample and cirs_env_step as separate launches
  redraw_tracker_backward   d loss / d s_t flows through call t only: one cirs_tracker_backward per call with the call's key and a
                            d-state tensor that is zero except at position t, gradients summed over the calls."""
import ctypes as C
from typing import Optional

import torch

from . import abi
from .rollout import DeviceRollout


def call_tag(rng_base: int, call: int) -> int:
    """Mask-key tag of build_state call `call` of the collect whose sampler counters start at rng_base."""
    # 16 bits for the call index (max_turn <= 65534), the collect's counter base above them: the tag space of set_dropout_key is 64 bits wide,
    # so neither a long episode nor a long run wraps one collect's tags into another's
    assert 0 <= int(call) < 0xFFFF, "more build_state calls per collect than the mask tag has room for"
    return ((int(rng_base) & 0xFFFFFFFFFF) << 16) + int(call) + 1


class RedrawRollout(DeviceRollout):
    def _replay(self, t, users, key_seed, rng_base, out):
        """Rebuild the tracker state of call t (prefix 0..t) with that call's masks; s_t -> out [B, S]."""
        trk, tr = self.tracker, self.traj
        trk.set_dropout_key(key_seed, call_tag(rng_base, t), self.dropout_env_base)
        trk.reset()
        S = trk.dim_state
        if t == 0:
            trk.init(users, out=out, out_stride=S)
            return
        trk.init(users, out=self._scratch, out_stride=S)
        for k in range(t):
            skip = (tr.act[k] < 0).to(torch.uint8)
            trk.step(tr.act[k].clamp(min=0), tr.rew[k], skip=skip, out=out if k == t - 1 else self._scratch, out_stride=S)
        self._keep = skip

    def collect(self, users: torch.Tensor, *, seed=0, rng_base=0, sync_every: Optional[int] = None, gumbel=None):
        assert gumbel is None and self.online is None and self.visited is None and self.force_length == 0, \
            "the exact-redraw option covers the plain training rollout"
        env, tr, trk = self.env, self.traj, self.tracker
        B, T, S = env.n_env, env.max_turn, trk.dim_state
        users = users.to(self.device, torch.int32)
        key_seed = seed >> 8 if self.dropout_key_from_high_bits else seed
        self._scratch = torch.empty((B, S), dtype=torch.float32, device=self.device)
        tr.clear()
        env.reset(users)
        self._users, self._key = users, (key_seed, rng_base)
        for t in range(T):
            self._replay(t, users, key_seed, rng_base, tr.obs[t])
            done = env.done.clone()
            self.policy.sample(tr.obs[t], seed=seed, rng_step=(rng_base + t) & 0xFFFFFFFF, skip=done, act_out=tr.act[t], logp_out=tr.logp[t],
                               value_out=tr.value[t])
            _, rew, dn, ctr, _ = env.step(tr.act[t].clamp(min=0) * (done == 0))
            live = done == 0
            tr.rew[t].copy_(torch.where(live, rew, torch.zeros_like(rew)))
            tr.done[t].copy_(torch.where(live, dn, torch.zeros_like(dn)))
            tr.ctr[t].copy_(ctr)
        self._replay(T, users, key_seed, rng_base, tr.obs[T]) if T < trk.cfg.max_len else None
        return env.turn.clone()


def redraw_tracker_backward(rollout: RedrawRollout, row_env, row_t, offsets, lens, n_rows, dstate):
    """Gradients of the tracker parameters under the exact-redraw procedure: sum over the calls t of the backward pass with call t's
    masks and d-state restricted to position t.  Leaves the sum in tracker.flat_grad."""
    trk, tr = rollout.tracker, rollout.traj
    key_seed, rng_base = rollout._key
    T = rollout.env.max_turn
    total = torch.zeros_like(trk.flat_grad)
    t_max = int(lens.max())
    for t in range(t_max):
        trk.set_dropout_key(key_seed, call_tag(rng_base, t), rollout.dropout_env_base)
        d_t = torch.zeros_like(dstate)
        d_t[t] = dstate[t]
        trk.backward(rollout._users, tr, row_env, row_t, offsets, lens, n_rows, d_t)
        total += trk.flat_grad
    trk.flat_grad.copy_(total)
