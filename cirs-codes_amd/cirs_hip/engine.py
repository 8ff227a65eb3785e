"""CirsEngine: the whole hot path on one GPU (or one rank of a multi-GPU job): device-resident rollout + PPO update.

Wires csrc/{env,tracker,policy,rollout,ppo,tracker_bwd}.hip through the C ABI.  Host-side counterpart of what
CIRS-RL-kuaishou.py builds (reference :141-292) and of onpolicy_trainer's inner loop
(core/trainer/onpolicy.py:170-209):   collect(n_episode = n_env)  ->  policy.update(0, buffer, batch_size, repeat).
"""
import math
from typing import Dict, Optional

import os

import numpy as np
import torch

from . import abi, distributed
from .env import DeviceEnv, DeviceEnvTables
from .learner import DeviceLearner, flat_policy_params
from .policy import DevicePolicy
from .rollout import DeviceRollout, Trajectory
from .tracker import DeviceTracker, flat_tracker_params, positional_encoding, tracker_param_shapes


def init_tracker_params(n_users, n_items, max_turn, seed=2021, dim_model=32, dim_state=20, nhead=4, d_hid=128, nlayers=2,
                        init_std=1e-4) -> Dict[str, torch.Tensor]:
    """Fresh parameters with the reference's initialisers (core/state_tracker.py:129-168, core/user_model.py:559-581):
    embeddings ~ N(0, init_std), nn.Linear / nn.TransformerEncoderLayer defaults, decoder ~ U(-0.1, 0.1), bias 0.
    torch modules are used as parameter factories only (their forward is never called)."""
    g = torch.Generator().manual_seed(seed)
    torch_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        D = dim_model
        p = {"embedding_dict.feat_user.weight": torch.randn(n_users, D, generator=g) * init_std,
             "embedding_dict.feat_item.weight": torch.randn(n_items, D, generator=g) * init_std}
        ffn_user = torch.nn.Linear(D, D)
        gate = torch.nn.Linear(1 + D, D)
        layer = torch.nn.TransformerEncoderLayer(D, nhead, d_hid, 0.1)
        enc = torch.nn.TransformerEncoder(layer, nlayers, enable_nested_tensor=False)
        dec = torch.nn.Linear(D, dim_state)
        dec.bias.data.zero_()
        dec.weight.data.uniform_(-0.1, 0.1)
        p.update({"ffn_user.weight": ffn_user.weight.data, "ffn_user.bias": ffn_user.bias.data,
                  "fnn_gate.weight": gate.weight.data, "fnn_gate.bias": gate.bias.data,
                  "decoder.weight": dec.weight.data, "decoder.bias": dec.bias.data})
        for k, v in enc.state_dict().items():
            p["transformer_encoder." + k] = v
        p["pos_encoder.pe"] = positional_encoding(max_turn + 1, D).unsqueeze(1)
    finally:
        torch.random.set_rng_state(torch_state)
    return {k: v.detach().clone().float() for k, v in p.items()}


def init_policy_params(n_items, seed=0, dim_state=20, hidden=64) -> Dict[str, torch.Tensor]:
    """Orthogonal weights, zero biases (CIRS-RL-kuaishou.py:250-254)."""
    g = torch.Generator().manual_seed(seed)
    shapes = {"actor.preprocess.model.model.0": (hidden, dim_state), "actor.preprocess.model.model.2": (hidden, hidden),
              "actor.last.model.0": (n_items, hidden), "critic.last.model.0": (1, hidden)}
    out = {}
    for k, shp in shapes.items():
        w = torch.empty(shp)
        torch.nn.init.orthogonal_(w, generator=g)
        out[k + ".weight"] = w
        out[k + ".bias"] = torch.zeros(shp[0])
    return out


class CirsEngine:
    def __init__(self, tables: DeviceEnvTables, n_env: int, *, max_turn=30, num_leave_compute=1, leave_threshold=0,
                 tau=100.0, gamma_exposure=10.0, version="v1", r_decay=1.0, dim_model=32, dim_state=20, nhead=4,
                 hidden=64, gamma=0.95, gae_lambda=0.95, eps_clip=0.2, vf_coef=0.25, ent_coef=0.0, max_grad_norm=0.5,
                 lr=1e-3, rew_norm=True, value_clip=True, norm_adv=True, seed=2023, tracker_params=None,
                 policy_params=None, dist_group=None, world_size=1, rank=0, force_gather=False, learner_mode="dp",
                 online_reward=None, batch_size_hint=1024, dropout=0.0, tracker_backward=None, dropout_redraw=False, coll=None):
        """dropout: probability of the tracker's five dropout sites.  0.0 (default) is the mode of every parity fixture and of
        the benchmark; 0.1 reproduces the reference's training procedure, whose tracker is never put in eval()."""
        self.device = tables.device
        self.tables = tables
        self.n_env, self.max_turn, self.S, self.D = n_env, max_turn, dim_state, dim_model
        U, I = tables.n_users, tables.n_items
        self.n_items = I
        self.world, self.rank, self.group = world_size, rank, dist_group
        self.force_gather = force_gather  # exercise the packed all-gather path even with one rank (tests)
        self.force_dp = force_gather and os.environ.get("CIRS_FORCE_DP", "0") == "1"   # + the data-parallel learner with one rank (tests)
        # Learners over the gathered buffer.  In EVERY mode `update(batch_size)` means the reference's PPO configuration: global
        # minibatches of batch_size rows (CIRS-RL-kuaishou.py:89, core/policy/ppo.py:180-181), however many ranks there are.
        #   "dp":         each global minibatch is sharded by rows (rank r takes rows r::W), ONE all-reduce of the flat gradient per
        #                 minibatch, identical clip + Adam on every rank
        #   "dp_sharded": same gradients; reduce-scatter -> clip + Adam on this rank's 1/W of the parameters -> all-gather of the
        #                 parameter shards (Adam moments sharded)
        #   "replicated": every rank runs the identical single-device learner on the gathered buffer (no further communication)
        # (the batch_size x W variant that keeps the number of optimiser steps per update constant is the caller's choice:
        #  update(batch_size * W); bench.py reports it as an extra key, never as the headline)
        #   "tp":         the actor head is sharded by ITEMS for the update (rank r owns rows r*Is .. of wa / ba and their Adam moments;
        #                 trunk / critic replicated): every rank runs every row of a minibatch against its 1/W of the catalogue, two
        #                 small collectives per minibatch (16 B per row of statistics, the d h2 partials), no gradient all-reduce; the
        #                 updated head shards are all-gathered into the replicated rollout policy once per update
        assert learner_mode in ("dp", "dp_sharded", "replicated", "tp")
        self.learner_mode = learner_mode
        # tracker BPTT of a multi-rank job: "sharded" = own envs + one gradient all-reduce (always for dp / dp_sharded / tp);
        # "replicated" = every rank over all envs, no communication (default of learner "replicated": results identical to one device)
        self.tracker_backward = tracker_backward or ("replicated" if learner_mode == "replicated" else "sharded")
        assert self.tracker_backward in ("sharded", "replicated") and (learner_mode == "replicated" or self.tracker_backward == "sharded")
        self.coll = coll or distributed.Collectives(group=dist_group, device=self.device)   # (coll: e.g. distributed.EmulatedPeers)
        self.env = DeviceEnv(tables, n_env, num_leave_compute=num_leave_compute, leave_threshold=leave_threshold,
                             max_turn=max_turn, tau=tau, gamma_exposure=gamma_exposure, version=version, r_decay=r_decay)
        tp = tracker_params or init_tracker_params(U, I, max_turn, seed=seed, dim_model=dim_model, dim_state=dim_state, nhead=nhead)
        self.tracker_flat, tviews = flat_tracker_params(tracker_param_shapes(U, I, dim_model, dim_state), device=self.device, init=tp)
        tparams = dict(tviews)
        tparams["pos_encoder.pe"] = tp["pos_encoder.pe"].to(self.device).float().contiguous()
        self.tracker = DeviceTracker(tparams, U, I, n_env, max_turn, dim_model=dim_model, dim_state=dim_state, nhead=nhead, device=self.device,
                                     dropout_p=dropout)
        self.tracker.enable_training(self.tracker_flat, lr=lr)
        pp = policy_params or init_policy_params(I, seed=seed, dim_state=dim_state, hidden=hidden)
        self.policy_flat, pviews = flat_policy_params(I, dim_state, hidden, device=self.device, init=pp, world=world_size)
        self.policy_views = pviews
        self.tracker_views = tviews
        self.hidden = hidden
        self.policy = DevicePolicy(pviews, I, dim_state=dim_state, hidden=hidden, device=self.device)
        # dropout_redraw: the reference's exact procedure (fresh masks over the whole prefix at every build_state call,
        # core/state_tracker.py:170-186,243-246): one batched prefix pass per call, one batched backward over all calls (cirs_hip/redraw.py)
        self.dropout_redraw = bool(dropout_redraw)
        if self.dropout_redraw:
            from .redraw import RedrawRollout
            assert online_reward is None and (world_size == 1 or (learner_mode == "replicated" and self.tracker_backward == "replicated")), \
                "the exact-redraw option runs on one device or with the replicated learner (every rank over the gathered buffer)"
            self.rollout = RedrawRollout(self.env, self.tracker, self.policy)
        else:
            self.rollout = DeviceRollout(self.env, self.tracker, self.policy, online=online_reward)
        self.rollout.dropout_env_base = rank * n_env
        self.rollout.dropout_key_from_high_bits = True
        self.B_total = n_env * world_size
        self.rollout.B_total = self.B_total
        self.learner = DeviceLearner(self.policy_flat, I, self.B_total, max_turn, dim_state=dim_state, hidden=hidden, gamma=gamma,
                                     gae_lambda=gae_lambda, eps_clip=eps_clip, vf_coef=vf_coef, ent_coef=ent_coef,
                                     max_grad_norm=max_grad_norm, lr=lr, norm_adv=norm_adv, value_clip=value_clip, rew_norm=rew_norm,
                                     world=world_size)
        # size every lazily grown buffer for the worst case now (B*T rows, merged last minibatch < 2*batch_size):
        # no allocation (= implicit device sync) ever happens inside the collect/update loop
        self.learner.reserve(self.B_total * max_turn, 2 * batch_size_hint)
        self.tracker.reserve_backward((self.B_total if world_size == 1 or self.tracker_backward == "replicated" else n_env) * max_turn)
        self.tp_learner = None
        if learner_mode == "tp" and world_size > 1:
            Is = -(-(-(-I // world_size)) // 32) * 32          # items per shard: ceil(I / W) rounded up to whole 32-item tiles
            self.tp_Is, self.tp_base = Is, rank * Is
            Il = min(Is, I - self.tp_base)
            assert Il > 0, "more ranks than 32-item tiles of the catalogue"
            self.tp_Il = Il
            init = {k: v for k, v in pviews.items() if not k.startswith("actor.last")}
            init["actor.last.model.0.weight"] = pviews["actor.last.model.0.weight"][self.tp_base:self.tp_base + Il]
            init["actor.last.model.0.bias"] = pviews["actor.last.model.0.bias"][self.tp_base:self.tp_base + Il]
            self.tp_flat, self.tp_views = flat_policy_params(Il, dim_state, hidden, device=self.device, init=init)
            self.tp_learner = DeviceLearner(self.tp_flat, Il, self.B_total, max_turn, dim_state=dim_state, hidden=hidden, gamma=gamma,
                                            gae_lambda=gae_lambda, eps_clip=eps_clip, vf_coef=vf_coef, ent_coef=ent_coef,
                                            max_grad_norm=max_grad_norm, lr=lr, norm_adv=norm_adv, value_clip=value_clip, rew_norm=rew_norm)
            self.tp_learner.reserve(self.B_total * max_turn, 2 * batch_size_hint)
            self._tp_send = torch.zeros(Is * (hidden + 1), dtype=torch.float32, device=self.device)
            self._tp_recv = torch.zeros(world_size * Is * (hidden + 1), dtype=torch.float32, device=self.device)
        self.seed = seed
        self.collect_count = 0
        self.users = None
        self.lengths = None
        self._gtraj = None
        self._users_pinned = None
        self._lens_pinned = None
        # user draws: the reference uses Python's (unseeded) random.randint per env reset (kuaishouEnv.py:155-159);
        # here a seeded generator per rank
        self._user_rng = np.random.RandomState(seed * 1000003 + rank)

    # ---- rollout ------------------------------------------------------------------------------------------------
    def collect(self, users: Optional[torch.Tensor] = None, sync_every: Optional[int] = None):
        if users is None:
            # pinned double buffer + asynchronous upload: the host does not block on the stream (it is usually one whole
            # update ahead of the GPU here), so the rollout launches queue behind the update without a bubble
            if self._users_pinned is None:
                self._users_pinned = [torch.empty(self.n_env, dtype=torch.int32).pin_memory() for _ in range(2)]
                self._users_dev = [torch.empty(self.n_env, dtype=torch.int32, device=self.device) for _ in range(2)]
                self._users_uploaded = [None, None]
            k = self.collect_count & 1
            if self._users_uploaded[k] is not None:
                # the upload issued from this pinned buffer two collects ago must have been consumed before the host overwrites
                # it (collect() may be called back to back without an update(), whose read-back would order it)
                self._users_uploaded[k].synchronize()
            self._users_pinned[k].numpy()[:] = self._user_rng.randint(0, self.tables.n_users, self.n_env)
            self._users_dev[k].copy_(self._users_pinned[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._users_uploaded[k] = ev
            self.users = self._users_dev[k]
        else:
            self.users = users.to(self.device, torch.int32)
        rng_base = (self.collect_count * self.max_turn) & 0xFFFFFFFF
        # RNG key = (seed, rank) so ranks draw independent noise; counter = (item, local env id, step)
        self.lengths = self.rollout.collect(self.users, seed=(self.seed << 8) + self.rank, rng_base=rng_base, sync_every=sync_every)
        self.collect_count += 1
        return self.lengths

    def collect_stats(self):
        """Collector.collect's result dict (collector.py:343-362) from the device trajectory (one host sync)."""
        tr = self.rollout.traj
        lens = self.lengths.cpu().numpy()
        valid = (tr.act >= 0)
        rews = (tr.rew * valid).sum(0).cpu().numpy()
        return {"n/ep": int(len(lens)), "n/st": int(lens.sum()), "rews": rews, "lens": lens, "rew": float(rews.mean()),
                "len": float(lens.mean()), "rew_std": float(rews.std()), "len_std": float(lens.std())}

    # ---- learner ------------------------------------------------------------------------------------------------
    def _gather(self):
        """All ranks' trajectories -> one global buffer (single all-gather); world == 1: zero-copy views."""
        tr = self.rollout.traj
        if self.world == 1 and not self.force_gather:
            return tr, self.tracker.x_hist, self.lengths, self.users
        fields = dict(obs=tr.obs, act=tr.act, rew=tr.rew, done=tr.done, logp=tr.logp, value=tr.value, ctr=tr.ctr,
                      x_hist=self.tracker.x_hist, lens=self.lengths.to(torch.int32), users=self.users)
        g = distributed.all_gather_records(fields, self.max_turn, self.n_env, self.S, self.D, group=self.group, coll=self.coll)
        if self._gtraj is None:
            self._gtraj = Trajectory(self.B_total, self.max_turn, self.S, self.device)
        gt = self._gtraj
        for name in ("obs", "act", "rew", "done", "logp", "value", "ctr"):
            getattr(gt, name).copy_(g[name])
        return gt, g["x_hist"], g["lens"], g["users"]

    def update(self, batch_size=1024, repeat=2, perms=None):
        """policy.update(0, buffer, batch_size, repeat): process_fn + learn + tracker step (base.py:219-244)."""
        traj, x_hist, lens_d, users = self._gather()
        ln = self.tp_learner if self.tp_learner is not None else self.learner
        # The host needs N to schedule the minibatches: the only read-back of an update.  The copy is enqueued FIRST and process_fn (GAE, returns,
        # compaction, with the row count left on the device) right behind it on the same stream; the host then waits for the copy alone, and
        # while it computes N and enqueues the permutations and the first minibatch the GPU is already running process_fn -- instead of
        # idling behind a synchronous copy until the host has enqueued those kernels.  (A side stream for the copy was measured too: the
        # cross-queue waits of this runtime cost more than the bubble, 7.25 vs 6.93 ms per step.)
        cur = torch.cuda.current_stream(self.device)
        if self._lens_pinned is None:
            self._lens_pinned = torch.empty(self.B_total, dtype=torch.int32).pin_memory()
        lens_i32 = lens_d if lens_d.dtype == torch.int32 else lens_d.to(torch.int32)
        if lens_i32.is_contiguous() and not os.environ.get("CIRS_READBACK_COPIES"):
            ln.readback_lens(lens_i32, self._lens_pinned)      # (one launch: the lengths + 4 more bytes -- did a hand-off wait of an earlier update's minibatch steps give up?)
        else:
            self._lens_pinned.copy_(lens_i32, non_blocking=True)
            ln.request_handoff_status()
        done = torch.cuda.Event(); done.record(cur)
        if perms is None and self.world > 1:
            # identical permutations on every rank (same key): learners stay bit-identical
            ln.perm_seed, ln.perm_tag = self.seed * 7919 + 1, self.collect_count * 64
        # (the update's minibatch permutations ride in process_fn's last launch -- the learner modes that shuffle with their own key draw theirs later)
        ln.prepare_async(traj, lens_i32, perm_repeat=repeat if (perms is None and not os.environ.get("CIRS_PERMS_SEPARATE")) else 0)
        done.synchronize()
        lens = self._lens_pinned.numpy().copy()
        if ln.handoff_lost():
            ln.check_handoffs(reset=True)   # raises CirsHipError
        self._last_prepared = (traj, lens, lens_d)      # bench.py's kernel probe re-prepares the full-catalogue learner from it in tp mode
        n = ln.finish_prepare(lens)
        offsets = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
        if (self.world > 1 or self.force_dp) and self.learner_mode in ("dp", "dp_sharded"):
            return self._update_dp(traj, lens, offsets, n, batch_size, repeat, perms)
        if self.tp_learner is not None:
            return self._update_dp(traj, lens, offsets, n, batch_size, repeat, perms, ln=ln)
        losses = ln.learn(batch_size, repeat, perms=perms)
        if self.world > 1 and self.tracker_backward == "sharded":
            # replicated policy learner, but the BPTT through the tracker (independent per env) over this rank's envs only + one
            # all-reduce of the tracker gradients per update: its cost does not grow with the number of ranks
            self._tracker_backward_sharded(ln, lens, offsets)
            return losses, n
        if self.dropout_redraw:
            from .redraw import redraw_tracker_backward
            redraw_tracker_backward(self.rollout, ln.b_env, ln.b_t, ln.offsets_dev, ln.lens_dev, n, ln.dobs, lens_host=lens, users=users, traj=traj,
                                    x_hist=x_hist)
            self.tracker.adam_update()
            return losses, n
        self.tracker.backward(users, traj, ln.b_env, ln.b_t, ln.offsets_dev, ln.lens_dev, n, ln.dobs,
                              x_hist=x_hist if (self.world > 1 or self.force_gather) else None)
        self.tracker.adam_update()
        return losses, n

    def _publish_tp(self):
        """The updated head shards -> the replicated rollout policy (one all-gather per update); trunk / critic are replicated."""
        H, Is, Il, W, I = self.hidden, self.tp_Is, self.tp_Il, self.world, self.n_items
        v, pv = self.tp_views, self.policy_views
        self._tp_send[:Il * H].copy_(v["actor.last.model.0.weight"].reshape(-1))
        self._tp_send[Is * H:Is * H + Il].copy_(v["actor.last.model.0.bias"])
        self.coll.all_gather(self._tp_recv, self._tp_send)
        r = self._tp_recv.view(W, Is * (H + 1))
        pv["actor.last.model.0.weight"].copy_(r[:, :Is * H].reshape(W * Is, H)[:I])
        pv["actor.last.model.0.bias"].copy_(r[:, Is * H:].reshape(W * Is)[:I])
        for k in pv:
            if not k.startswith("actor.last"):
                pv[k].copy_(v[k])

    def _update_dp(self, traj, lens, offsets, n, batch_size, repeat, perms, ln=None):
        """Data-parallel learner over global minibatches of batch_size rows: per minibatch one all-reduce of the flat policy
        gradients ("dp") or reduce-scatter + sharded Adam + all-gather ("dp_sharded"); per update one all-reduce of d loss/d obs and
        one of the tracker gradients.  Every rank ends with identical parameters."""
        ln = ln or self.learner
        all_reduce = self.coll.all_reduce
        if self.learner_mode == "tp":
            losses = ln.learn_tp(batch_size, repeat, perms, self.rank, self.world, self.tp_base, self.coll)
            self._publish_tp()          # d loss / d obs is replicated (the trunk backward runs on every rank): no all-reduce
        elif self.learner_mode == "dp_sharded":
            losses = ln.learn_dp_sharded(batch_size, repeat, perms, self.rank, self.world, self.coll)
            all_reduce(ln.dobs)  # each (t, env) row was written by exactly one rank
        else:
            losses = ln.learn_dp(batch_size, repeat, perms, self.rank, self.world, all_reduce)
            all_reduce(ln.dobs)  # each (t, env) row was written by exactly one rank
        self._tracker_backward_sharded(ln, lens, offsets)
        return losses, n

    def _tracker_backward_sharded(self, ln, lens, offsets):
        """Tracker backward over THIS rank's envs only (its own trajectory / slots), then the sum of the gradients over the ranks."""
        all_reduce = self.coll.all_reduce
        Bl = self.n_env
        lo_env, hi_env = self.rank * Bl, (self.rank + 1) * Bl
        r0 = int(offsets[lo_env])
        r1 = int(offsets[hi_env - 1] + lens[hi_env - 1])
        # this rank's slice of the episode offsets / lengths, taken from the device copies (no upload, no host sync)
        off_l = (ln.offsets_dev[lo_env:hi_env] - r0).contiguous()
        lens_l = ln.lens_dev[lo_env:hi_env].contiguous()
        row_env_l = (ln.b_env[r0:r1] - lo_env).contiguous()
        row_t_l = ln.b_t[r0:r1].contiguous()
        dstate_l = ln.dobs[:, lo_env:hi_env, :].contiguous()
        self.tracker.backward(self.users, self.rollout.traj, row_env_l, row_t_l, off_l, lens_l, r1 - r0, dstate_l)
        all_reduce(self.tracker.flat_grad)
        self.tracker.adam_update()
