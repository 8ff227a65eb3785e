"""BASELINE configs[4] in its multi-GPU form (last paragraph): a 10^6 x 10^6 problem whose embedding tables and
actor head are SPLIT over the ranks instead of replicated.

  * ShardedTable       rows of an nn.Embedding-style table distributed by `id mod W` (rank id % W holds row id // W).  `lookup(ids)`
                       = one all-to-all of the requested local row numbers + one all-to-all of the returned rows; the local side is
                       cirs_gather_rows.  Fixed-capacity messages (cap slots per destination, -1 = empty): no host synchronisation,
                       the exchange is stream-ordered like everything else on the rollout path.
  * ShardedRollout     the Collector loop for env-sharded ranks with a column-sharded actor head: per vector step
                         all-gather of the tracker states (B_local x 20 floats per rank)            -> every rank sees every env
                         cirs_actor_shard_partials on this rank's item shard                          -> (score, id, logit, max, sum-exp) per env
                         all-to-all of those tuples                                                   -> every rank holds W tuples per own env
                         cirs_actor_merge_shards (fixed rank order)                                   -> action / log-prob
                         online reward: DeepFM rows of the chosen items via ShardedTable.lookup, cirs_deepfm_forward on the
                         compact per-step tables; env step; tracker step on the looked-up item rows.
                       Action ids and rewards are bit-identical to running all envs and the whole catalogue on one device
                       (DeviceRollout + OnlineReward): noise counters use global env / item ids, every logit is the same fma chain,
                       arg-max and tie-breaks are order independent, embedding rows are copied bit for bit.
  * Comm               the collectives behind both: `DistComm` (torch.distributed: RCCL over xGMI on MI355X, gloo in the CPU test)
                       and `ThreadComm` (W virtual ranks as threads of one process sharing one GPU: the single-GPU test vehicle).

The reference has no counterpart (DeepCTR-Torch only prints a notice for use_hash, deepctr_torch/inputs.py:31-33): semantics are
"same as everything on one device", which is what tests/test_gpu_sharded.py checks.
"""
import ctypes as C
import threading
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import abi


# ------------------------------------------------------------------------------------------------------------------------------
# collectives
# ------------------------------------------------------------------------------------------------------------------------------
class DistComm:
    """torch.distributed (backend nccl = RCCL on ROCm; gloo on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """[...] on every rank -> [W, ...] (rank-major)."""
        t = t.contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        if t.device.type == "cpu":      # gloo has no all_gather_into_tensor
            self._dist.all_gather(list(out.unbind(0)), t, group=self.group)
        else:
            self._dist.all_gather_into_tensor(out, t, group=self.group)
        return out

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        """sum over the ranks, in place (identical bits on every rank)."""
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t

    def all_to_all(self, t: torch.Tensor) -> torch.Tensor:
        """t[d] goes to rank d; returns r with r[s] = what rank s sent here.  Equal splits."""
        t = t.contiguous()
        assert t.shape[0] == self.world
        out = torch.empty_like(t)
        if t.device.type == "cpu":      # gloo implements neither all_to_all nor all_to_all_single: all-gather + column pick (layout test only)
            out.copy_(self.all_gather(t)[:, self.rank])
        else:
            self._dist.all_to_all_single(out, t, group=self.group)
        return out


class LocalComm:
    """world 1: every collective is the identity (the split code path on ONE device: bench.py's c5_split probe)."""
    world, rank = 1, 0

    def all_gather(self, t):
        return t.contiguous().unsqueeze(0)

    def all_to_all(self, t):
        return t.contiguous()

    def all_reduce(self, t):
        return t


class ThreadComm:
    """W virtual ranks = W threads of one process (one GPU).  Every collective is a barrier-separated exchange through shared slots;
    device work is ordered by a device synchronisation at the barrier (test vehicle: simplicity over speed)."""

    class _Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots: List[Optional[torch.Tensor]] = [None] * world

    def __init__(self, shared: "_Shared", rank: int):
        self.s, self.rank, self.world = shared, rank, shared.world

    @classmethod
    def make(cls, world):
        shared = cls._Shared(world)
        return [cls(shared, r) for r in range(world)]

    def _exchange(self, t):
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
        self.s.slots[self.rank] = t
        self.s.barrier.wait()
        got = list(self.s.slots)
        self.s.barrier.wait()
        return got

    def all_gather(self, t):
        return torch.stack(self._exchange(t.contiguous()), 0)

    def all_to_all(self, t):
        got = self._exchange(t.contiguous())
        return torch.stack([g[self.rank] for g in got], 0)

    def all_reduce(self, t):
        got = self._exchange(t.clone())
        total = got[0].clone()
        for g in got[1:]:       # rank order: identical bits on every rank
            total += g
        t.copy_(total)
        return t


# ------------------------------------------------------------------------------------------------------------------------------
# row-sharded table
# ------------------------------------------------------------------------------------------------------------------------------
def hip_gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[k] = table[idx[k]] (idx < 0 -> zeros) through cirs_gather_rows."""
    n = idx.numel()
    out = torch.empty((n, table.shape[1]), dtype=torch.float32, device=table.device)
    abi.check(abi.lib().cirs_gather_rows(table.data_ptr(), table.shape[1], idx.data_ptr(), n, out.data_ptr(),
                                         torch.cuda.current_stream(table.device).cuda_stream), "cirs_gather_rows")
    return out


class ShardedTable:
    """Global table [n_rows, row_floats] (row_floats % 4 == 0) of which this rank stores rows {id : id % W == rank}, at local
    position id // W.  `gather` is the local row gather (product: cirs_gather_rows; the CPU layout test injects its own)."""

    def __init__(self, local_rows: torch.Tensor, n_rows_global: int, comm, gather: Callable = hip_gather_rows):
        assert local_rows.dim() == 2 and local_rows.shape[1] % 4 == 0 and local_rows.dtype == torch.float32
        self.local, self.n_rows, self.comm, self.gather = local_rows.contiguous(), int(n_rows_global), comm, gather
        W, r = comm.world, comm.rank
        assert local_rows.shape[0] == (self.n_rows - r + W - 1) // W, "local shard must hold rows rank, rank + W, rank + 2W, ..."

    @staticmethod
    def shard_of(full: torch.Tensor, rank: int, world: int) -> torch.Tensor:
        return full[rank::world].contiguous()

    def lookup(self, ids: torch.Tensor, cap: Optional[int] = None) -> torch.Tensor:
        """ids [n] int64 global row ids (>= 0) -> rows [n, row_floats].  cap: message slots per destination (default n = the worst
        case, every id owned by one rank).  No host synchronisation: the per-owner counts are a one-hot column sum (torch.bincount
        would read its output size back), and a cap smaller than the largest per-owner count does not index out of range -- the
        overflowing requests are dropped (zero rows) and counted in `self.overflow` (device scalar); `check_overflow()` reads it
        back and raises.  The FIRST call with a reduced cap checks eagerly (one sync) so that a wrong cap fails at once."""
        W = self.comm.world
        n = ids.numel()
        cap = n if cap is None else int(cap)
        dev = ids.device
        ids = ids.to(torch.int64)
        owner = ids % W
        order = torch.argsort(owner, stable=True)
        owner_s = owner[order]
        counts = (owner.unsqueeze(1) == torch.arange(W, device=dev).unsqueeze(0)).sum(0)   # [W], stays on the device
        start = torch.cumsum(counts, 0) - counts
        pos_s = torch.arange(n, device=dev) - start[owner_s]           # slot inside the message to its owner
        fits = pos_s < cap
        if cap < n:
            if self.overflow is None:
                self.overflow = torch.zeros((), dtype=torch.int64, device=dev)
                first = True
            else:
                first = False
            self.overflow += (~fits).sum()
            if first:
                self.check_overflow()
        # column `cap` is a dump slot for overflowing requests (never sent)
        send_x = torch.full((W, cap + 1), -1, dtype=torch.int64, device=dev)
        send_x[owner_s, torch.clamp(pos_s, max=cap)] = ids[order] // W
        send = send_x[:, :cap].contiguous()
        recv = self.comm.all_to_all(send)                               # [W(src), cap] local row numbers requested from this rank
        rows = self.gather(self.local, recv.reshape(-1)).reshape(W, cap, -1)
        back = self.comm.all_to_all(rows)                               # [W(owner), cap, R]
        slot = torch.empty(n, dtype=torch.int64, device=dev)
        slot[order] = torch.where(fits, owner_s * cap + pos_s, torch.full_like(pos_s, -1))   # -1 -> zero row
        return self.gather(back.reshape(W * cap, -1), slot)

    overflow: Optional[torch.Tensor] = None

    # ---- training side: gradient rows travel to their owners, the owner scatters them in a fixed order and runs Adam on its shard ----
    def enable_training(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        assert self.local.shape[1] == 32, "the ordered scatter serves dim_model == 32 rows"
        self.grad = torch.zeros_like(self.local)
        self.adam_m, self.adam_v = torch.zeros_like(self.local), torch.zeros_like(self.local)
        self.adam_steps, self.lr, self.betas, self.adam_eps = 0, lr, betas, eps
        self._scatter_ws = None

    def push_grads(self, ids: torch.Tensor, rows: torch.Tensor, cap: Optional[int] = None):
        """ids [n] int64 global row ids (< 0: no contribution), rows [n, 32] their gradient rows -> self.grad [local rows, 32] = for every
        local row the sum of ALL ranks' contributions, added in (source rank, position) order: one all-to-all of the local row numbers
        and one of the rows (the mirror image of `lookup`), then cirs_embedding_scatter (stable sort + ordered segment sums: no float
        atomics).  With envs sharded contiguously over the ranks that order is the single-device buffer order."""
        W = self.comm.world
        n = ids.numel()
        cap = n if cap is None else int(cap)
        dev = ids.device
        ids = ids.to(torch.int64)
        valid = ids >= 0
        owner = torch.where(valid, ids % W, torch.full_like(ids, W))        # pseudo-owner W: dropped
        order = torch.argsort(owner, stable=True)
        owner_s = owner[order]
        counts = (owner.unsqueeze(1) == torch.arange(W + 1, device=dev).unsqueeze(0)).sum(0)
        start = torch.cumsum(counts, 0) - counts
        pos_s = torch.arange(n, device=dev) - start[owner_s]
        keep = (owner_s < W) & (pos_s < cap)
        if cap < n:      # a reduced capacity may drop gradient rows: counted like lookup()'s dropped requests, never silent
            if self.overflow is None:
                self.overflow = torch.zeros((), dtype=torch.int64, device=dev)
            self.overflow += ((owner_s < W) & ~(pos_s < cap)).sum()
        send_id = torch.full((W + 1, cap + 1), -1, dtype=torch.int64, device=dev)
        send_id[owner_s, torch.clamp(pos_s, max=cap)] = torch.where(keep, ids[order] // W, torch.full_like(pos_s, -1))
        send_rows = torch.zeros((W + 1, cap + 1, rows.shape[1]), dtype=torch.float32, device=dev)
        send_rows[owner_s, torch.clamp(pos_s, max=cap)] = rows[order].to(torch.float32)
        recv_id = self.comm.all_to_all(send_id[:W, :cap].contiguous())            # [W(src), cap]
        recv_rows = self.comm.all_to_all(send_rows[:W, :cap].contiguous())        # [W(src), cap, 32]
        keys = recv_id.reshape(-1).to(torch.int32).contiguous()
        contrib = recv_rows.reshape(-1, rows.shape[1]).contiguous()
        lib = abi.lib()
        need = int(lib.cirs_embedding_scatter_workspace_bytes(keys.numel()))
        if self._scatter_ws is None or self._scatter_ws.numel() < need:
            self._scatter_ws = torch.empty(need, dtype=torch.uint8, device=dev)
        abi.check(lib.cirs_embedding_scatter(keys.data_ptr(), contrib.data_ptr(), keys.numel(), self.local.shape[0], self.grad.data_ptr(),
                                             self._scatter_ws.data_ptr(), self._scatter_ws.numel(),
                                             torch.cuda.current_stream(dev).cuda_stream), "cirs_embedding_scatter")
        self._keep = (keys, contrib)

    def adam_update(self):
        """torch.optim.Adam over the whole local shard (rows without a gradient still decay their moments, like the dense optimiser of
        the reference over the full table)."""
        n = self.local.numel()
        abi.check(abi.lib().cirs_adam_step(self.local.data_ptr(), self.grad.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(), n,
                                           self.adam_steps, 1, self.lr, self.betas[0], self.betas[1], self.adam_eps, None, 0,
                                           torch.cuda.current_stream(self.local.device).cuda_stream), "cirs_adam_step")
        self.adam_steps += 1

    def check_overflow(self):
        """Host check of the dropped-request counter (one synchronisation): raises if any lookup exceeded its `cap`."""
        if self.overflow is not None and int(self.overflow) > 0:
            raise RuntimeError(f"ShardedTable: {int(self.overflow)} lookup requests / gradient rows exceeded the per-destination capacity (cap too small)")


# ------------------------------------------------------------------------------------------------------------------------------
# sharded rollout
# ------------------------------------------------------------------------------------------------------------------------------
class ShardedRollout:
    """One rank of the C5 rollout.  env / tracker / traj are this rank's B_local envs (cirs_hip.env.DeviceEnv,
    cirs_hip.tracker.DeviceTracker built over PLACEHOLDER 1-row embedding tables, cirs_hip.rollout.Trajectory);
      policy_shard   dict(w1, b1, w2, b2, wc, bc replicated; wa [I_shard, 64], ba [I_shard] = this rank's contiguous item range
                     [item_base, item_base + I_shard)) as fp32 device tensors
      fm             cirs_hip.deepfm.DeviceDeepFM holding the replicated small tensors (feat tables, DNN) and PLACEHOLDER user / item
                     tables; fm_user / fm_item: ShardedTable with rows [emb (E) | lin (1) | 0 0 0]
      trk_user / trk_item   ShardedTable of the tracker's embedding_dict.feat_user / feat_item rows (dim_model floats)
      raw_uid, raw_pid, item_feats [I,4] i32, item_dur [I] f32, minmax: as cirs_hip.rollout.OnlineReward (replicated, 28 B / item)."""

    def __init__(self, comm, env, tracker, traj, policy_shard: Dict[str, torch.Tensor], item_base: int, n_items_total: int, fm, fm_user,
                 fm_item, trk_user, trk_item, raw_uid, raw_pid, item_feats, item_dur, minmax):
        self.comm, self.env, self.tracker, self.traj = comm, env, tracker, traj
        self.device = env.device
        self.B = env.n_env
        self.W, self.rank = comm.world, comm.rank
        self.item_base, self.I_total = int(item_base), int(n_items_total)
        p = {k: v.to(self.device, torch.float32).contiguous() for k, v in policy_shard.items()}
        self._pol = p
        self.I_shard = p["wa"].shape[0]
        self.pcfg = abi.PolicyCfg(n_items=self.I_shard, dim_state=p["w1"].shape[1], hidden=64)
        self.pw = abi.PolicyWeights(**{k: p[k].data_ptr() for k in ("w1", "b1", "w2", "b2", "wa", "ba", "wc", "bc")})
        self.fm, self.fm_user, self.fm_item, self.trk_user, self.trk_item = fm, fm_user, fm_item, trk_user, trk_item
        dev = self.device
        self.raw_uid = torch.as_tensor(raw_uid).to(dev, torch.int64)
        self.raw_pid = torch.as_tensor(raw_pid).to(dev, torch.int64)
        self.item_feats = torch.as_tensor(item_feats).to(dev, torch.int32).contiguous()
        self.item_dur = torch.as_tensor(item_dur).to(dev, torch.float32).contiguous()
        self.minmax = torch.as_tensor(minmax, dtype=torch.float32).to(dev).contiguous()
        self._lib = abi.lib()
        Bt = self.B * self.W
        self._ws = torch.empty(self._lib.cirs_policy_workspace_bytes(C.byref(self.pcfg), Bt), dtype=torch.uint8, device=dev)
        self._tuples = torch.empty((5, Bt), dtype=torch.float32, device=dev)
        self._value_all = torch.empty(Bt, dtype=torch.float32, device=dev)
        self._pred = torch.empty(self.B, dtype=torch.float32, device=dev)
        self._arange = torch.arange(self.B, dtype=torch.int64, device=dev)
        self._arange32 = self._arange.to(torch.int32)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # -- compact-table views: the kernels index `table[id]`; a looked-up [B, R] block with id = arange(B) is such a table --------
    def _tracker_weights(self, emb_user=None, emb_item=None):
        w = type(self.tracker.w).from_buffer_copy(self.tracker.w)
        if emb_user is not None:
            w.emb_user = emb_user.data_ptr()
        if emb_item is not None:
            w.emb_item = emb_item.data_ptr()
        cfg = type(self.tracker.cfg).from_buffer_copy(self.tracker.cfg)
        cfg.n_users = cfg.n_items = self.B
        return cfg, w

    def reset(self, users: torch.Tensor):
        """Collector.reset_env: env.reset + tracker init with the users' rows fetched from their owners."""
        users = users.to(self.device, torch.int64)
        self.users = users
        self.traj.clear()
        self.tracker.reset()
        self.env.reset(users.to(torch.int32))
        rows = self.trk_user.lookup(users)                                  # [B, D]
        cfg, w = self._tracker_weights(emb_user=rows)
        S = self.tracker.dim_state
        abi.check(self._lib.cirs_tracker_init(C.byref(cfg), C.byref(w), C.byref(self.tracker.st), self._arange32.data_ptr(), None,
                                              self.B, self.traj.obs[0].data_ptr(), S, self._stream()), "cirs_tracker_init")
        self._keep = [rows]
        fu = self.fm_user.lookup(self.raw_uid[users])                        # [B, E + 4] = [emb | lin | pad]
        E = self.fm.cfg.emb_dim
        self._fm_user_emb = fu[:, :E].contiguous()
        self._fm_user_lin = fu[:, E].contiguous()

    def step(self, t: int, seed: int, rng_base: int):
        B, W, S = self.B, self.W, self.tracker.dim_state
        tr = self.traj
        # 1) every rank sees every env's state (and which envs are finished)
        states = self.comm.all_gather(tr.obs[t]).reshape(W * B, S)
        done_all = self.comm.all_gather(self.env.done).reshape(W * B)
        # 2) this rank's item shard against all envs -> one tuple per env
        abi.check(self._lib.cirs_actor_shard_partials(
            C.byref(self.pcfg), C.byref(self.pw), states.data_ptr(), S, W * B, seed, (rng_base + t) & 0xFFFFFFFF, None, None,
            done_all.data_ptr(), self.item_base, self.I_total, self._tuples.data_ptr(), self._value_all.data_ptr(), self._ws.data_ptr(),
            self._ws.numel(), self._stream()), "cirs_actor_shard_partials")
        # 3) tuples of env block d go to rank d; 4) fold the W tuples of each own env in rank order
        recv = self.comm.all_to_all(self._tuples.view(5, W, B).permute(1, 0, 2).contiguous())      # [W(shard), 5, B]
        abi.check(self._lib.cirs_actor_merge_shards(recv.data_ptr(), W, B, self.env.done.data_ptr(), tr.act[t].data_ptr(),
                                                    tr.logp[t].data_ptr(), self._stream()), "cirs_actor_merge_shards")
        tr.value[t].copy_(self._value_all[self.rank * B:(self.rank + 1) * B])
        act = tr.act[t]
        a = act.clamp(min=0)
        # 5) online reward (simulated_env.py:88-98, commented-out variant): DeepFM rows of the chosen items from their owners
        fi = self.fm_item.lookup(self.raw_pid[a])
        E = self.fm.cfg.emb_dim
        emb_item, lin_item = fi[:, :E].contiguous(), fi[:, E].contiguous()
        w = type(self.fm.w).from_buffer_copy(self.fm.w)
        w.emb_user, w.lin_user = self._fm_user_emb.data_ptr(), self._fm_user_lin.data_ptr()
        w.emb_item, w.lin_item = emb_item.data_ptr(), lin_item.data_ptr()
        cfg = type(self.fm.cfg).from_buffer_copy(self.fm.cfg)
        cfg.n_user_vocab = cfg.n_item_vocab = B
        feats, dur = self.item_feats[a].contiguous(), self.item_dur[a].contiguous()
        abi.check(self._lib.cirs_deepfm_forward(C.byref(cfg), C.byref(w), self._arange.data_ptr(), self._arange.data_ptr(), feats.data_ptr(),
                                                dur.data_ptr(), B, self._pred.data_ptr(), self._stream()), "cirs_deepfm_forward")
        # 6) env step on the online score
        tab = type(self.env._tab).from_buffer_copy(self.env._tab)
        tab.pred_online, tab.pred_minmax = self._pred.data_ptr(), self.minmax.data_ptr()
        scratch = torch.empty(B, dtype=torch.int64, device=self.device)
        abi.check(self._lib.cirs_env_step(C.byref(self.env.cfg), C.byref(tab), C.byref(self.env._st), act.data_ptr(), None, B,
                                          scratch.data_ptr(), tr.rew[t].data_ptr(), tr.done[t].data_ptr(), tr.ctr[t].data_ptr(), None,
                                          self._stream()), "cirs_env_step")
        # 7) tracker step on the chosen items' rows (finished envs: item id -1 -> no append)
        rows = self.trk_item.lookup(a)
        tcfg, tw = self._tracker_weights(emb_item=rows)
        items = torch.where(act >= 0, self._arange, torch.full_like(self._arange, -1))
        abi.check(self._lib.cirs_tracker_step(C.byref(tcfg), C.byref(tw), C.byref(self.tracker.st), items.data_ptr(), tr.rew[t].data_ptr(),
                                              None, None, B, tr.obs[t + 1].data_ptr(), S, self._stream()), "cirs_tracker_step")
        self._keep = [rows, emb_item, lin_item, feats, dur, scratch, items, recv, states, done_all]   # alive until the stream consumed them

    def collect(self, users: torch.Tensor, seed=0, rng_base=0):
        if self.tracker.cfg.dropout_p > 0:   # masks per GLOBAL env of the job, fresh per collect (as DeviceRollout.collect)
            self.tracker.set_dropout_key(seed, rng_base, self.rank * self.B)
        self.reset(users)
        for t in range(self.env.max_turn):
            self.step(t, seed, rng_base)
        return self.env.turn.clone()


# ------------------------------------------------------------------------------------------------------------------------------
# sharded trainer: the PPO update of the split configuration
# ------------------------------------------------------------------------------------------------------------------------------
class _CollAdapter:
    """cirs_hip.distributed.Collectives' (out, inp) calling convention over a Comm of this module."""

    def __init__(self, comm):
        self.comm = comm

    def all_gather(self, out, inp):
        out.copy_(self.comm.all_gather(inp).reshape(out.shape))

    def all_reduce(self, t):
        self.comm.all_reduce(t)


class ShardedTrainer:
    """One rank of `policy.update` for BASELINE configs[4] on top of a ShardedRollout (reference semantics: core/policy/ppo.py:96-246 on
    the gathered buffer; no reference counterpart for the split itself):
      1. all-gather of the packed trajectory records (as C4) -> every rank holds the buffer, GAE / returns replicated;
      2. DeviceLearner.learn_tp over this rank's ITEM shard of the actor head (the tensors the rollout samples from ARE the learner's
         flat shard parameters) -- trunk / critic replicated;
      3. tracker BPTT over this rank's envs with COMPACT embedding tables (the rows its users / actions looked up: id = slot), so the
         kernels never see a 10^6-row table; dense tracker gradients are all-reduced and stepped on every rank;
      4. the per-slot embedding gradient rows go to their owners (ShardedTable.push_grads: all-to-all + ordered scatter) and every
         owner runs Adam on its shard of feat_user / feat_item."""

    def __init__(self, rollout: ShardedRollout, policy_flat: torch.Tensor, B_total: int, *, gamma=0.95, gae_lambda=0.95, eps_clip=0.2,
                 vf_coef=0.25, ent_coef=0.0, max_grad_norm=0.5, lr=1e-3, norm_adv=True, value_clip=True, rew_norm=True):
        from .learner import DeviceLearner
        from .rollout import Trajectory
        self.ro = rollout
        self.comm, self.W, self.rank, self.Bl = rollout.comm, rollout.W, rollout.rank, rollout.B
        self.device = rollout.device
        T, S = rollout.env.max_turn, rollout.tracker.dim_state
        self.T, self.S, self.B_total = T, S, B_total
        self.learner = DeviceLearner(policy_flat, rollout.I_shard, B_total, T, dim_state=S, gamma=gamma, gae_lambda=gae_lambda, eps_clip=eps_clip,
                                     vf_coef=vf_coef, ent_coef=ent_coef, max_grad_norm=max_grad_norm, lr=lr, norm_adv=norm_adv,
                                     value_clip=value_clip, rew_norm=rew_norm)
        self.gtraj = Trajectory(B_total, T, S, self.device)
        self.coll = _CollAdapter(self.comm)
        for tab in (rollout.trk_user, rollout.trk_item):
            tab.enable_training(lr=lr)
        D = rollout.tracker.cfg.dim_model
        self._gU = torch.zeros((self.Bl, D), dtype=torch.float32, device=self.device)
        self._gI = torch.zeros((T * self.Bl, D), dtype=torch.float32, device=self.device)
        self._bws = None

    def _gather(self, lens_local):
        from . import distributed
        tr, trk = self.ro.traj, self.ro.tracker
        fields = dict(obs=tr.obs, act=tr.act, rew=tr.rew, done=tr.done, logp=tr.logp, value=tr.value, ctr=tr.ctr, x_hist=trk.x_hist,
                      lens=lens_local.to(torch.int32), users=self.ro.users.to(torch.int32))
        local = distributed.pack_records(fields)
        g = distributed.unpack_records(self.comm.all_gather(local), self.W, self.T, self.Bl, self.S, trk.cfg.dim_model)
        for name in ("obs", "act", "rew", "done", "logp", "value", "ctr"):
            getattr(self.gtraj, name).copy_(g[name])
        return g["lens"]

    def update(self, lens_local: torch.Tensor, batch_size=1024, repeat=2, perms=None, perm_key=(20230, 0)):
        ro, ln, W, r, Bl, T = self.ro, self.learner, self.W, self.rank, self.Bl, self.T
        lens_d = self._gather(lens_local)
        lens = lens_d.cpu().numpy().astype(np.int32)
        # (the read-back above is the update's one synchronisation: the dropped-message counters of the sharded tables -- lookups of the
        # rollout just finished, gradient pushes of the previous update -- are read here, where it costs no extra stall)
        for tab in (ro.trk_user, ro.trk_item, getattr(ro, "fm_user", None), getattr(ro, "fm_item", None)):
            if tab is not None:
                tab.check_overflow()
        n = ln.prepare(self.gtraj, lens, lens_dev=lens_d)
        if perms is None:
            ln.perm_seed, ln.perm_tag = perm_key          # the same key on every rank: identical shuffles
        losses = ln.learn_tp(batch_size, repeat, perms, r, W, ro.item_base, self.coll)
        # ---- tracker BPTT over this rank's envs, compact embedding tables --------------------------------------------------------
        offsets = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        lo_env, hi_env = r * Bl, (r + 1) * Bl
        r0, r1 = int(offsets[lo_env]), int(offsets[hi_env - 1] + lens[hi_env - 1])
        off_l = (ln.offsets_dev[lo_env:hi_env] - r0).contiguous()
        lens_l = ln.lens_dev[lo_env:hi_env].contiguous()
        row_env_l = (ln.b_env[r0:r1] - lo_env).contiguous()
        row_t_l = ln.b_t[r0:r1].contiguous()
        dstate_l = ln.dobs[:, lo_env:hi_env, :].contiguous()
        tr, trk = ro.traj, ro.tracker
        act = tr.act                                                     # [T, Bl] global item ids, -1 once finished
        slot = (torch.arange(T, device=self.device).unsqueeze(1) * Bl + torch.arange(Bl, device=self.device).unsqueeze(0))
        act_c = torch.where(act >= 0, slot, torch.full_like(slot, -1)).contiguous()      # item id = its own slot t * Bl + b
        rows_u = ro.trk_user.lookup(ro.users)                                               # [Bl, D]
        rows_i = ro.trk_item.lookup(act.clamp(min=0).reshape(-1))                           # [T * Bl, D]
        cfg = type(trk.cfg).from_buffer_copy(trk.cfg)
        cfg.n_users, cfg.n_items = Bl, T * Bl
        w = type(trk.w).from_buffer_copy(trk.w)
        w.emb_user, w.emb_item = rows_u.data_ptr(), rows_i.data_ptr()
        g = type(trk.g).from_buffer_copy(trk.g)
        g.emb_user, g.emb_item = self._gU.data_ptr(), self._gI.data_ptr()
        lib = abi.lib()
        need = lib.cirs_tracker_backward_workspace_bytes(C.byref(cfg), r1 - r0)
        if self._bws is None or self._bws.numel() < need:
            self._bws = torch.empty(need, dtype=torch.uint8, device=self.device)
        users_c = self._arange32(Bl)
        abi.check(lib.cirs_tracker_backward(C.byref(cfg), C.byref(w), C.byref(trk.st), users_c.data_ptr(), act_c.data_ptr(), tr.rew.data_ptr(),
                                            row_env_l.data_ptr(), row_t_l.data_ptr(), off_l.data_ptr(), lens_l.data_ptr(), r1 - r0,
                                            dstate_l.data_ptr(), C.byref(g), self._bws.data_ptr(), self._bws.numel(),
                                            torch.cuda.current_stream(self.device).cuda_stream), "cirs_tracker_backward")
        # dense tracker parameters: replicated, gradients summed over the ranks (the placeholder embedding rows of the flat buffer ride along)
        self.comm.all_reduce(trk.flat_grad)
        trk.adam_update()
        # embedding rows: to their owners, in buffer (env-major) order
        ro.trk_user.push_grads(ro.users.to(torch.int64), self._gU)
        rows_env_major = slot.t().reshape(-1)                               # slot ids env by env, t ascending
        ids_env_major = act.t().reshape(-1)
        ro.trk_item.push_grads(ids_env_major, self._gI[rows_env_major])
        ro.trk_user.adam_update()
        ro.trk_item.adam_update()
        self._keep = (rows_u, rows_i, act_c, off_l, lens_l, row_env_l, row_t_l, dstate_l)
        return losses, n

    def _arange32(self, n):
        return torch.arange(n, dtype=torch.int32, device=self.device)
