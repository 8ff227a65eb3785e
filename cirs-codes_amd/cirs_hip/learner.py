"""Device engine of the PPO learner (csrc/ppo.hip): flat parameter / gradient / Adam buffers, process_fn + learn."""
import ctypes as C
import math
from typing import Dict, List, Optional

import numpy as np
import torch

from . import abi
from .policy import POLICY_FIELDS

# flat layout of include/cirs_hip.h: [w1 | b1 | w2 | b2 | wa | ba | wc | bc]
FLAT_ORDER = ["w1", "b1", "w2", "b2", "wa", "ba", "wc", "bc"]
FIELD_TO_NAME = dict(POLICY_FIELDS)


def padded_param_count(total, world):
    """P_pad: the flat gradient incl. its 4 loss partials, rounded up so that `world` equal shards are whole float4s."""
    q = 4 * max(1, int(world))
    return (total + 4 + q - 1) // q * q


def flat_policy_params(n_items, dim_state=20, hidden=64, device="cuda", init: Optional[Dict[str, torch.Tensor]] = None, world=1):
    """Allocate the flat fp32 parameter buffer and return (flat, {reference state_dict name: view}).  The allocation holds
    padded_param_count(P, world) floats (flat = its first P): the sharded optimiser all-gathers parameter shards straight into it."""
    shapes = dict(w1=(hidden, dim_state), b1=(hidden,), w2=(hidden, hidden), b2=(hidden,), wa=(n_items, hidden),
                  ba=(n_items,), wc=(1, hidden), bc=(1,))
    total = sum(int(np.prod(shapes[k])) for k in FLAT_ORDER)
    flat = torch.zeros(padded_param_count(total, world), dtype=torch.float32, device=device)[:total]
    views, off = {}, 0
    for k in FLAT_ORDER:
        n = int(np.prod(shapes[k]))
        views[FIELD_TO_NAME[k]] = flat[off:off + n].view(shapes[k])
        off += n
    if init is not None:
        for name, t in init.items():
            if name in views:
                views[name].copy_(t.to(device=device, dtype=torch.float32).reshape(views[name].shape))
    return flat, views


def minibatch_slices(n, batch_size):
    """Batch.split(size, merge_last=True) index ranges (tianshou/data/batch.py:734-744)."""
    merge_last = n % batch_size > 0
    out = []
    for s0 in range(0, n, batch_size):
        if merge_last and s0 + 2 * batch_size >= n:
            out.append((s0, n))
            break
        out.append((s0, min(n, s0 + batch_size)))
    return out


class DeviceLearner:
    def __init__(self, flat_params: torch.Tensor, n_items, n_env, max_turn, *, dim_state=20, hidden=64, gamma=0.99,
                 gae_lambda=0.95, eps_clip=0.2, vf_coef=0.5, ent_coef=0.01, max_grad_norm=None, lr=1e-3, norm_adv=True,
                 value_clip=False, rew_norm=False, betas=(0.9, 0.999), adam_eps=1e-8, world=1, dual_clip=None):
        self.device = flat_params.device
        self.cfg = abi.PpoCfg(n_items=n_items, dim_state=dim_state, hidden=hidden, norm_adv=int(bool(norm_adv)),
                              value_clip=int(bool(value_clip)), rew_norm=int(bool(rew_norm)), gamma=gamma,
                              gae_lambda=gae_lambda, eps_clip=eps_clip, vf_coef=vf_coef, ent_coef=ent_coef,
                              max_grad_norm=float(max_grad_norm or 0.0), lr=lr, beta1=betas[0], beta2=betas[1],
                              adam_eps=adam_eps, dual_clip=float(dual_clip or 0.0))
        assert dual_clip is None or dual_clip > 1.0, "Dual-clip PPO parameter should greater than 1.0."   # core/policy/ppo.py:79-80
        # recompute_advantage (core/policy/ppo.py:176-177): value_fn(traj) refreshes traj.value with the CURRENT critic; learn() then
        # re-runs prepare() (GAE, returns, a second RunningMeanStd update -- a2c.py:80-109) before every repeat but the first
        self.value_fn = None
        self._prep = None
        self._lib = abi.lib()
        assert flat_params.numel() == self._lib.cirs_ppo_param_count(C.byref(self.cfg))
        self.params = flat_params
        # flat gradient + 4 loss partials, padded to `world` equal float4-aligned shards (the padding stays zero); the Adam moments
        # and -- when the caller allocated it that way (flat_policy_params(world=...)) -- the parameters have the same padded extent,
        # so that shard r of every buffer is the slice [r * P_pad / world, (r + 1) * P_pad / world)
        P = flat_params.numel()
        self.P, self.P_pad, self.world = P, padded_param_count(P, world), int(world)
        self.grads = torch.zeros(self.P_pad, dtype=torch.float32, device=self.device)
        self._m_pad = torch.zeros(self.P_pad, dtype=torch.float32, device=self.device)
        self._v_pad = torch.zeros(self.P_pad, dtype=torch.float32, device=self.device)
        self.adam_m, self.adam_v = self._m_pad[:P], self._v_pad[:P]
        st = flat_params.untyped_storage()
        if flat_params.storage_offset() == 0 and st.nbytes() >= 4 * self.P_pad and flat_params.is_contiguous():
            self._p_pad = torch.empty(0, dtype=torch.float32, device=self.device).set_(st, 0, (self.P_pad,), (1,))
        else:
            self._p_pad = None   # sharded optimiser then gathers into a scratch buffer and copies P floats back
        self.opt_step = 0
        self.perm_seed, self.perm_tag = 20230, 0   # key of the minibatch shuffles (see _perms_on_device)
        self.n_env, self.max_turn, self.S = n_env, max_turn, dim_state
        self.rms_state = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64, device=self.device)  # RunningMeanStd()
        self._ws = None
        self._batch_cap = 0
        self.dobs = torch.zeros((max_turn + 1, n_env, dim_state), dtype=torch.float32, device=self.device)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- in-launch hand-offs of the minibatch step: a wait that gave up is counted on the device (csrc/ppo.hip: flags_wait) ---------------------
    def request_handoff_status(self):
        """Enqueue a copy of the sticky "lost hand-off" count into a pinned host word (no synchronisation); read it with handoff_lost() once the
        stream has passed this point -- CirsEngine.update() piggybacks on its one read-back."""
        if getattr(self, "_ho_dev", None) is None:
            self._ho_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._ho_pin = torch.zeros(1, dtype=torch.int32).pin_memory()
        abi.check(self._lib.cirs_ppo_handoff_status(self._ho_dev.data_ptr(), 0, self._stream()), "cirs_ppo_handoff_status")
        self._ho_pin.copy_(self._ho_dev, non_blocking=True)

    def readback_lens(self, lens_dev, lens_pinned):
        """The update's one read-back from one launch: the lengths and the hand-off count straight into pinned host words (cirs_ppo_update_readback) --
        and, for this learner's own envs, the buffer offsets / row count of process_fn (prepare_async then starts at the GAE)."""
        if getattr(self, "_ho_dev", None) is None:
            self._ho_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._ho_pin = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._offsets_of = None
        off = nrow = None
        if int(lens_dev.numel()) == self.n_env:
            self._ensure_prep_buffers()
            off, nrow = self._off_buf.data_ptr(), self._n_dev.data_ptr()
            self._offsets_of = lens_dev.data_ptr()
        abi.check(self._lib.cirs_ppo_update_readback(lens_dev.data_ptr(), int(lens_dev.numel()), lens_pinned.data_ptr(), self._ho_pin.data_ptr(), off, nrow,
                                                     self._stream()), "cirs_ppo_update_readback")

    def _ensure_prep_buffers(self):
        B, T = self.n_env, self.max_turn
        if getattr(self, "_prep_scratch", None) is None:
            self._prep_scratch = torch.empty(B * T, dtype=torch.float64, device=self.device)
            self._off_buf = torch.empty(B, dtype=torch.int32, device=self.device)
            self._n_dev = torch.zeros(1, dtype=torch.int32, device=self.device)

    def handoff_lost(self):
        """The count last copied by request_handoff_status() (the caller has synchronised with the stream)."""
        return int(self._ho_pin[0]) if getattr(self, "_ho_pin", None) is not None else 0

    def check_handoffs(self, reset=False):
        """Synchronous check (tests, end of a run): raises when a bounded wait inside the minibatch step ever gave up -- the update that contained it
        used incomplete gradient sums or pre-Adam trunk weights."""
        self.request_handoff_status()
        torch.cuda.current_stream(self.device).synchronize()
        lost = self.handoff_lost()
        if reset and lost:
            abi.check(self._lib.cirs_ppo_handoff_status(self._ho_dev.data_ptr(), 1, self._stream()), "cirs_ppo_handoff_status")
            self._ho_pin.zero_()
        if lost:
            raise abi.CirsHipError(f"{lost} in-launch hand-off wait(s) of the PPO minibatch step timed out: the update is invalid "
                                   "(csrc/ppo.hip: flags_wait; GPU shared with another process or a dispatch order that is not block-id order?)")

    def _alloc_batch(self, n):
        if n > self._batch_cap:
            dev, S = self.device, self.S
            self.b_obs = torch.empty((n, S), dtype=torch.float32, device=dev)
            self.b_act = torch.empty(n, dtype=torch.int32, device=dev)
            self.b_adv = torch.empty(n, dtype=torch.float32, device=dev)
            self.b_ret = torch.empty(n, dtype=torch.float32, device=dev)
            self.b_vs = torch.empty(n, dtype=torch.float32, device=dev)
            self.b_logp = torch.empty(n, dtype=torch.float32, device=dev)
            self.b_env = torch.empty(n, dtype=torch.int32, device=dev)
            self.b_t = torch.empty(n, dtype=torch.int32, device=dev)
            self._batch_cap = n
        self.batch = abi.PpoBatch(obs=self.b_obs.data_ptr(), act=self.b_act.data_ptr(), adv=self.b_adv.data_ptr(),
                                  ret=self.b_ret.data_ptr(), v_s=self.b_vs.data_ptr(), logp_old=self.b_logp.data_ptr(),
                                  row_env=self.b_env.data_ptr(), row_t=self.b_t.data_ptr())

    def reserve(self, max_rows, max_minibatch):
        """Allocate the buffer-order batch and the minibatch workspace for their maximum sizes up front."""
        self._alloc_batch(max_rows)
        self.workspace(max_minibatch)
        self._idx_dev = torch.empty(max_rows, dtype=torch.int32, device=self.device)

    def prepare(self, traj, lens_host: np.ndarray, lens_dev: Optional[torch.Tensor] = None):
        """process_fn: GAE + return normalisation + compaction into buffer order.  lens_host: episode lengths [B];
        lens_dev: the same on the device (saves the upload; the offsets are then formed there as well)."""
        lens_host = np.asarray(lens_host, dtype=np.int32)
        n = int(lens_host.sum())
        self._alloc_batch(n)
        self.n_rows = n
        if lens_dev is not None:
            lens_d = lens_dev.to(self.device, torch.int32).contiguous()
            off_d = (torch.cumsum(lens_d, 0, dtype=torch.int32) - lens_d).contiguous()
        else:
            offsets = np.concatenate([[0], np.cumsum(lens_host)[:-1]]).astype(np.int32)
            lens_d = torch.as_tensor(lens_host).to(self.device)
            off_d = torch.as_tensor(offsets).to(self.device)
        self.lens_dev, self.offsets_dev = lens_d, off_d
        self._prep = (traj, lens_host, lens_dev)
        abi.check(self._lib.cirs_ppo_prepare(C.byref(self.cfg), C.byref(traj.struct), lens_d.data_ptr(), off_d.data_ptr(),
                                             self.n_env, self.max_turn, n, self.rms_state.data_ptr(), C.byref(self.batch),
                                             self._stream()), "cirs_ppo_prepare")
        return n

    def prepare_async(self, traj, lens_dev: torch.Tensor, perm_repeat: int = 0):
        """prepare() enqueued WITHOUT the host knowing the row count (cirs_ppo_prepare_async: offsets and N are formed on the device from
        lens_dev).  The caller reads the lengths back meanwhile and completes the call with finish_prepare(lens_host).
        perm_repeat > 0: the update's `perm_repeat` minibatch permutations (key (perm_seed, perm_tag ..)) come out of process_fn's last launch
        (cirs_ppo_prepare_async_perms); _perms_on_device() hands them out if the key still is the one they were drawn with."""
        B, T = self.n_env, self.max_turn
        self._alloc_batch(B * T)
        self._ensure_prep_buffers()
        lens_d = lens_dev if (lens_dev.dtype == torch.int32 and lens_dev.is_contiguous()) else lens_dev.to(self.device, torch.int32).contiguous()
        self.lens_dev, self.offsets_dev = lens_d, self._off_buf
        self._perm_pre = None
        offsets_ready = 1 if getattr(self, "_offsets_of", None) == lens_d.data_ptr() else 0      # (readback_lens formed them from these very lengths)
        self._offsets_of = None
        if (0 < perm_repeat <= 8) or offsets_ready:
            perm_repeat = perm_repeat if 0 < perm_repeat <= 8 else 0
            if getattr(self, "_perm_buf", None) is None or self._perm_buf.numel() < max(perm_repeat, 1) * B * T:
                self._perm_buf = torch.empty(max(perm_repeat, 1) * B * T, dtype=torch.int32, device=self.device)
            abi.check(self._lib.cirs_ppo_prepare_async_perms(C.byref(self.cfg), C.byref(traj.struct), lens_d.data_ptr(), B, T, self._off_buf.data_ptr(),
                                                             self._n_dev.data_ptr(), self.rms_state.data_ptr(), C.byref(self.batch),
                                                             self._prep_scratch.data_ptr(), int(self.perm_seed), int(self.perm_tag), int(perm_repeat),
                                                             self._perm_buf.data_ptr(), offsets_ready, self._stream()), "cirs_ppo_prepare_async_perms")
            if perm_repeat:
                self._perm_pre = (int(self.perm_seed), int(self.perm_tag), int(perm_repeat))
        else:
            abi.check(self._lib.cirs_ppo_prepare_async(C.byref(self.cfg), C.byref(traj.struct), lens_d.data_ptr(), B, T, self._off_buf.data_ptr(),
                                                       self._n_dev.data_ptr(), self.rms_state.data_ptr(), C.byref(self.batch),
                                                       self._prep_scratch.data_ptr(), self._stream()), "cirs_ppo_prepare_async")
        self._prep = (traj, None, lens_dev)

    def finish_prepare(self, lens_host: np.ndarray):
        lens_host = np.asarray(lens_host, dtype=np.int32)
        self.n_rows = int(lens_host.sum())
        self._prep = (self._prep[0], lens_host, self._prep[2])
        return self.n_rows

    def workspace(self, mb):
        need = self._lib.cirs_ppo_workspace_bytes(C.byref(self.cfg), mb)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---- data-parallel building blocks (one global minibatch = the union of the ranks' row shards) -----------------
    def mb_phase1(self, idx_local, idx_global, want_dobs, loss_slot):
        """forward + backward of this rank's shard -> self.grads (incl. loss partials in the tail), ready to be summed."""
        ws = self.workspace(max(int(idx_local.numel()), 2))
        abi.check(self._lib.cirs_ppo_minibatch_dp(
            C.byref(self.cfg), self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
            self.opt_step, C.byref(self.batch), idx_local.data_ptr(), int(idx_local.numel()), idx_global.data_ptr(),
            int(idx_global.numel()), self.dobs.data_ptr() if want_dobs else None, self.n_env, loss_slot.data_ptr(),
            ws.data_ptr(), ws.numel(), 1, self._stream()), "cirs_ppo_minibatch_dp(phase 1)")

    def mb_phase2(self, mb_local, mb_global, loss_slot):
        """clip_grad_norm_ + Adam from the (all-reduced) gradients in self.grads."""
        ws = self.workspace(max(mb_local, 2))
        abi.check(self._lib.cirs_ppo_minibatch_dp(
            C.byref(self.cfg), self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
            self.opt_step, C.byref(self.batch), None, mb_local, None, mb_global, None, self.n_env, loss_slot.data_ptr(),
            ws.data_ptr(), ws.numel(), 2, self._stream()), "cirs_ppo_minibatch_dp(phase 2)")
        self.opt_step += 1

    def _perms_on_device(self, n, repeat, perms):
        """[repeat, n] int32 on the device: the recorded permutations (parity tests) or keyed pseudo-random permutations.
        Batch.split(shuffle=True) draws np.random.permutation(n) on the host (tianshou/data/batch.py).  Here every repeat is ONE
        launch of cirs_random_permutation (Feistel network, one thread per index, key = (perm_seed, running tag)): no host work,
        no upload and none of the sort / duplicate-handling launches of torch.randperm (A/B on one box: +0.5 % env-steps/s).  Ranks of a data-parallel learner set the same (perm_seed, tag) (CirsEngine.update), so they shuffle
        identically."""
        if perms is not None:
            return torch.as_tensor(np.stack([np.asarray(perms[rep]).astype(np.int32) for rep in range(repeat)])).to(self.device)
        pre = getattr(self, "_perm_pre", None)
        if pre is not None and pre == (int(self.perm_seed), int(self.perm_tag), int(repeat)) and n == self.n_rows:
            self._perm_pre = None                      # drawn in process_fn's last launch with this very key: [repeat][n], row stride n
            self.perm_tag += repeat
            return self._perm_buf[:repeat * n].view(repeat, n)
        out = torch.empty((repeat, n), dtype=torch.int32, device=self.device)
        abi.check(self._lib.cirs_random_permutations(int(n), int(self.perm_seed), int(self.perm_tag), int(repeat), out.data_ptr(), self._stream()),
                  "cirs_random_permutations")      # (one launch for the repeats: tags perm_tag .. perm_tag + repeat - 1, as one call per repeat drew them)
        self.perm_tag += repeat
        return out

    def learn_dp(self, global_batch, repeat, perms, rank, world, all_reduce, want_tracker_grad=True):
        """Data-parallel learn(): GLOBAL minibatches of `global_batch` rows (the reference's batch_size, CIRS-RL-kuaishou.py:89 /
        core/policy/ppo.py:180-181: the PPO configuration does not change with the number of ranks), rows rank::world of each
        belong to this rank; `all_reduce(tensor)` sums a flat tensor over the ranks in place (torch.distributed.all_reduce).
        The steps run as a chain (cirs_ppo_minibatch_dp_chain): phase 2 of a step also runs the head of the next one (this rank's rows of the
        next global minibatch on the weights it has just formed), so a step is 3 launches + all-reduce + 2 launches (7 + 2 before round 5)."""
        n = self.n_rows
        slices = minibatch_slices(n, global_batch)
        steps = [(rep, s0, e0) for rep in range(repeat) for s0, e0 in slices]
        losses = torch.zeros((len(steps), 4), dtype=torch.float32, device=self.device)
        perm_all_d = self._perms_on_device(n, repeat, perms)
        n_local = lambda s0, e0: len(range(rank, e0 - s0, world))
        max_ml = max(max(n_local(s0, e0) for _, s0, e0 in steps), 2)
        assert min(n_local(s0, e0) for _, s0, e0 in steps) >= 1, "global minibatch smaller than the world size"
        ws = self.workspace(max_ml)
        if getattr(self, "_lidx2", None) is None or self._lidx2[0].numel() < max_ml:
            self._lidx2 = [torch.empty(max(max_ml, 2048), dtype=torch.int32, device=self.device) for _ in range(2)]   # (step k's rows in buffer k & 1)

        def rows_of(k):
            rep, s0, e0 = steps[k]
            g = perm_all_d[rep][s0:e0]
            l = self._lidx2[k & 1][:n_local(s0, e0)]
            l.copy_(g[rank::world])
            return g, l

        def call(phase, k, g, l, head_done, nxt):
            rep = steps[k][0]
            want = want_tracker_grad and rep == repeat - 1
            ng, nl = nxt if nxt is not None else (None, None)
            abi.check(self._lib.cirs_ppo_minibatch_dp_chain(
                C.byref(self.cfg), self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.opt_step,
                C.byref(self.batch), l.data_ptr(), int(l.numel()), g.data_ptr(), int(g.numel()), self.dobs.data_ptr() if (want and phase == 1) else None,
                self.n_env, losses[k].data_ptr(), ws.data_ptr(), ws.numel(), phase, max_ml, int(head_done),
                nl.data_ptr() if nl is not None else None, int(nl.numel()) if nl is not None else 0,
                ng.data_ptr() if ng is not None else None, int(ng.numel()) if ng is not None else 0, self._stream()), f"cirs_ppo_minibatch_dp_chain(phase {phase})")

        cur, head_done = rows_of(0), False
        for k in range(len(steps)):
            if want_tracker_grad and steps[k][0] == repeat - 1 and steps[k][1] == 0:
                self.dobs.zero_()
            call(1, k, cur[0], cur[1], head_done, None)
            all_reduce(self.grads)
            nxt = rows_of(k + 1) if k + 1 < len(steps) else None
            call(2, k, cur[0], cur[1], False, nxt)
            self.opt_step += 1
            cur, head_done = nxt, nxt is not None
        return losses

    def learn_dp_sharded(self, global_batch, repeat, perms, rank, world, coll, want_tracker_grad=True):
        """learn_dp with the optimiser sharded over the ranks (ZeRO-1 form): per global minibatch
            phase 1                                  this rank's row shard -> flat gradient (+ loss partials)
            reduce_scatter(grads[P_pad])             -> this rank's summed shard of P_pad / world floats
            cirs_ppo_shard_norm + all_gather(stats)  -> squared-norm partials of every shard on every rank (72 floats per rank)
            cirs_ppo_shard_adam                      clip coefficient (fixed (rank, block) order) + Adam on the shard only
            all_gather(parameter shards)             -> every rank holds the updated parameters
        Same bytes on the wire as the all-reduce of learn_dp (a ring all-reduce IS reduce-scatter + all-gather), Adam and the
        norm pass over 1 / world of the buffer, Adam moments only maintained for the rank's own shard.  `coll` provides
        reduce_scatter(out, inp) / all_gather(out, inp) (cirs_hip.distributed.Collectives)."""
        assert world == self.world, "DeviceLearner(world=...) fixes the shard layout"
        n = self.n_rows
        slices = minibatch_slices(n, global_batch)
        losses = torch.zeros((repeat * len(slices), 4), dtype=torch.float32, device=self.device)
        perm_all_d = self._perms_on_device(n, repeat, perms)
        sl = self.P_pad // world
        b0 = rank * sl
        nstat = int(self._lib.cirs_ppo_shard_stat_floats())
        if getattr(self, "_gshard", None) is None or self._gshard.numel() != sl:
            self._gshard = torch.zeros(sl, dtype=torch.float32, device=self.device)
            self._stats = torch.zeros(nstat, dtype=torch.float32, device=self.device)
            self._stats_all = torch.zeros(world * nstat, dtype=torch.float32, device=self.device)
            self._p_scratch = None if self._p_pad is not None else torch.zeros(self.P_pad, dtype=torch.float32, device=self.device)
        p_pad = self._p_pad if self._p_pad is not None else self._p_scratch
        if self._p_pad is None:
            p_pad[:self.P].copy_(self.params)
        k = 0
        for rep in range(repeat):
            perm_d = perm_all_d[rep]
            last = rep == repeat - 1
            if last and want_tracker_grad:
                self.dobs.zero_()
            for s0, e0 in slices:
                g_idx = perm_d[s0:e0]
                l_idx = self._local_rows(g_idx, rank, world)
                self.mb_phase1(l_idx, g_idx, last and want_tracker_grad, losses[k])
                coll.reduce_scatter(self._gshard, self.grads)
                abi.check(self._lib.cirs_ppo_shard_norm(C.byref(self.cfg), self._gshard.data_ptr(), b0, sl, self._stats.data_ptr(),
                                                        self._stream()), "cirs_ppo_shard_norm")
                coll.all_gather(self._stats_all, self._stats)
                abi.check(self._lib.cirs_ppo_shard_adam(
                    C.byref(self.cfg), p_pad.data_ptr() + 4 * b0, self._gshard.data_ptr(), self._m_pad.data_ptr() + 4 * b0,
                    self._v_pad.data_ptr() + 4 * b0, b0, sl, self.opt_step, self._stats_all.data_ptr(), world, losses[k].data_ptr(),
                    self._stream()), "cirs_ppo_shard_adam")
                coll.all_gather(p_pad, p_pad[b0:b0 + sl])
                if self._p_pad is None:
                    self.params.copy_(p_pad[:self.P])
                self.opt_step += 1
                k += 1
        return losses

    def _local_rows(self, g_idx, rank, world):
        """Rows rank::world of a global minibatch in a preallocated buffer (stream-ordered reuse: no allocation per minibatch)."""
        m = len(range(rank, g_idx.numel(), world))
        assert m >= 1, "global minibatch smaller than the world size"
        if getattr(self, "_lidx_buf", None) is None or self._lidx_buf.numel() < m:
            self._lidx_buf = torch.empty(max(m, 2048), dtype=g_idx.dtype, device=self.device)
        l_idx = self._lidx_buf[:m]
        l_idx.copy_(g_idx[rank::world])
        return l_idx

    def learn_tp(self, batch_size, repeat, perms, rank, world, item_base, coll, want_tracker_grad=True):
        """learn() for an ITEM-SHARDED actor head (tensor-parallel; BASELINE configs[4]): this learner was built over the shard
        (n_items = the rank's item count, flat parameters [trunk | wa_shard | ba_shard | critic]); the batch holds global item ids
        and every rank runs every row.  Per minibatch: phase 1 -> all-gather of 16 B per row (max, sum-exp, sum exp z, the
        action's logit from its owner) -> phase 2 (merge in rank order, fused backward over the local items: the shard's head
        gradient is complete) -> all-reduce of the d h2 partials (+ entropy clamp partials + squared-norm slots) -> phase 3
        (replicated trunk backward, global clip coefficient, Adam on the shard + trunk).  Trunk, critic and the row statistics
        stay bit-identical across ranks; `coll`: all_gather(out, inp), all_reduce(t) (cirs_hip.distributed.Collectives)."""
        n = self.n_rows
        slices = minibatch_slices(n, batch_size)
        max_mb = max(e - s for s, e in slices)
        ws = self.workspace(max_mb)
        losses = torch.zeros((repeat * len(slices), 4), dtype=torch.float32, device=self.device)
        perm_all_d = self._perms_on_device(n, repeat, perms)
        pad = lambda m: (m + 31) // 32 * 32
        nred_max = int(self._lib.cirs_ppo_tp_exchange_floats(max_mb, world))
        if getattr(self, "_tp_red", None) is None or self._tp_red.numel() < nred_max or self._tp_all.numel() < world * 4 * pad(max_mb):
            self._tp_stats = torch.zeros(4 * pad(max_mb), dtype=torch.float32, device=self.device)
            self._tp_all = torch.zeros(world * 4 * pad(max_mb), dtype=torch.float32, device=self.device)
            self._tp_red = torch.zeros(nred_max, dtype=torch.float32, device=self.device)
            self._tp_fm = torch.zeros(world * 4 * pad(max_mb), dtype=torch.float32, device=self.device)

        nred_of = {e - s: int(self._lib.cirs_ppo_tp_exchange_floats(e - s, world)) for s, e in slices}      # (one ABI call per distinct size, not per minibatch)

        def call(phase, idx_ptr, mb, stats4, stats_all, red, want_dobs, loss_ptr):
            abi.check(self._lib.cirs_ppo_minibatch_tp(
                C.byref(self.cfg), self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
                self.opt_step, C.byref(self.batch), idx_ptr, mb, int(item_base), int(rank), int(world), stats4, stats_all, red,
                self.dobs.data_ptr() if want_dobs else None, self.n_env, loss_ptr, ws.data_ptr(), ws.numel(), phase, self._stream()),
                f"cirs_ppo_minibatch_tp(phase {phase})")

        k = 0
        for rep in range(repeat):
            perm_d = perm_all_d[rep]
            last = rep == repeat - 1
            if last and want_tracker_grad:
                self.dobs.zero_()
            for s0, e0 in slices:
                mb, npad = e0 - s0, pad(e0 - s0)
                idx_ptr = perm_d.data_ptr() + 4 * s0
                stats4, gathered = self._tp_stats[:4 * npad], self._tp_all[:world * 4 * npad]
                red = self._tp_red[:nred_of[mb]]
                call(1, idx_ptr, mb, stats4.data_ptr(), None, red.data_ptr(), False, None)
                coll.all_gather(gathered, stats4)
                stats_all = self._tp_fm[:world * 4 * npad].view(4, world, npad)             # field-major: [4][world][n_pad]
                stats_all.copy_(gathered.view(world, 4, npad).permute(1, 0, 2))            # (preallocated: no allocation per minibatch)
                call(2, idx_ptr, mb, None, stats_all.data_ptr(), red.data_ptr(), False, None)
                coll.all_reduce(red)
                call(3, idx_ptr, mb, None, None, red.data_ptr(), last and want_tracker_grad, losses.data_ptr() + 16 * k)
                self.opt_step += 1
                k += 1
        return losses

    def learn(self, batch_size, repeat, perms: Optional[List[np.ndarray]] = None, want_tracker_grad=True, recompute_adv=False, step_calls=False):
        """learn(): `repeat` passes of shuffled minibatches.  Returns loss arrays + leaves d loss / d obs of the LAST
        repeat in self.dobs ([T+1, B, S]) for the tracker backward.  perms: recorded permutations (parity tests);
        default: draws of the seeded device generator (_perms_on_device).  recompute_adv: before every repeat but the first the
        stored states are valued again with the current critic and process_fn's return computation is redone (ppo.py:176-177).
        step_calls: one cirs_ppo_minibatch call per step from this loop (what recompute_adv needs; also the tests' reference for cirs_ppo_learn)."""
        n = self.n_rows
        slices = minibatch_slices(n, batch_size)
        max_mb = max(e - s for s, e in slices)
        ws = self.workspace(max_mb)
        n_steps = repeat * len(slices)
        losses = (torch.empty if (not recompute_adv and not step_calls) else torch.zeros)((n_steps, 4), dtype=torch.float32, device=self.device)   # (cirs_ppo_learn writes every row)
        # all permutations of this update at once, before the first minibatch: the launches of the following repeats then
        # queue back to back
        perm_all_d = self._perms_on_device(n, repeat, perms)
        if not recompute_adv and not step_calls:
            # the whole loop from one call (cirs_ppo_learn: the same step kernels; the optimiser launch of a step also runs the head of the next)
            assert int(self._lib.cirs_ppo_learn_steps(n, batch_size, repeat)) == n_steps
            abi.check(self._lib.cirs_ppo_learn(
                C.byref(self.cfg), self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.opt_step,
                C.byref(self.batch), perm_all_d.data_ptr(), n, batch_size, repeat, self.dobs.data_ptr() if want_tracker_grad else None,
                self.dobs.numel(), self.n_env, losses.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), "cirs_ppo_learn")
            self.opt_step += n_steps
            return losses
        k = 0
        for rep in range(repeat):
            perm_d = perm_all_d[rep]
            last = rep == repeat - 1
            if last and want_tracker_grad:
                self.dobs.zero_()  # optim_state.zero_grad() at the top of each repeat (ppo.py:174)
            if recompute_adv and rep > 0:
                assert self.value_fn is not None and self._prep is not None, "recompute_adv needs value_fn (critic over the stored states)"
                self.value_fn(self._prep[0])
                assert self.prepare(*self._prep) == n
            for s0, e0 in slices:
                mb = e0 - s0
                abi.check(self._lib.cirs_ppo_minibatch(
                    C.byref(self.cfg), self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(),
                    self.adam_v.data_ptr(), self.opt_step, C.byref(self.batch), perm_d.data_ptr() + 4 * s0, mb,
                    self.dobs.data_ptr() if (last and want_tracker_grad) else None, self.n_env,
                    losses.data_ptr() + 16 * k, ws.data_ptr(), ws.numel(), self._stream()), "cirs_ppo_minibatch")
                self.opt_step += 1
                k += 1
        return losses
