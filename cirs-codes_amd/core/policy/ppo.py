"""PPOPolicy (reference core/policy/ppo.py:14-246 on top of tianshou A2CPolicy/PGPolicy/BasePolicy) on the device engines.

Same constructor keywords as the reference script passes (CIRS-RL-kuaishou.py:267-285) and the same protocol:
  policy(batch, buffer, state=None, remove_recommended_ids=False) -> Batch(act, ...)        (collector.py:233)
  policy.update(0, buffer, batch_size=, repeat=) -> {"loss", "loss/clip", "loss/vf", "loss/ent"}   (onpolicy.py:199-201)
`optim` is the list [optim_RL, optim_state]; the torch optimisers are used for their hyper-parameters only -- the
update itself (clip_grad_norm_ + Adam, incl. the duplicated-trunk quirk) runs in csrc/ppo.hip on flat buffers."""
from typing import Any, Dict, List, Optional

import numpy as np
import torch
from torch import nn

from cirs_hip.learner import DeviceLearner, flat_policy_params
from cirs_hip.policy import DevicePolicy
from tianshou.data import Batch


class PPOPolicy(nn.Module):
    def __new__(cls, actor=None, critic=None, optim=None, dist_fn=None, *args, **kwargs):
        """A continuous actor (tianshou.utils.net.continuous.ActorProb: CIRS-RL-taobao.py:208, BASELINE configs[0], CPU plumbing) is
        served by the host PPO of core.host_rl; the discrete catalogue actor by the device learner below."""
        from tianshou.utils.net.continuous import ActorProb
        if cls is PPOPolicy and isinstance(actor, ActorProb):
            from core.host_rl import HostPPOPolicy
            return HostPPOPolicy(actor, critic, optim, dist_fn, *args, **kwargs)
        return super().__new__(cls)

    def __init__(self, actor, critic, optim, dist_fn=None, eps_clip=0.2, dual_clip=None, value_clip=False,
                 advantage_normalization=True, recompute_advantage=False, discount_factor=0.99, vf_coef=0.5, ent_coef=0.01,
                 max_grad_norm=None, gae_lambda=0.95, max_batchsize=256, reward_normalization=False, action_scaling=True,
                 action_bound_method="clip", action_space=None, lr_scheduler=None, deterministic_eval=False, **kwargs):
        super().__init__()
        assert dual_clip is None or dual_clip > 1.0, "Dual-clip PPO parameter should greater than 1.0."   # reference ppo.py:79-80
        assert not deterministic_eval, "the reference never enables deterministic_eval for KuaishouEnv (SURVEY Q10)"
        if not reward_normalization:
            assert not value_clip, "value clip is available only when `reward_normalization` is True"
        self.actor, self.critic = actor, critic
        self.optim = optim
        self.callbacks: List[Any] = []
        self.updating = False
        self.lr_scheduler = lr_scheduler
        self._hyper = dict(gamma=discount_factor, gae_lambda=gae_lambda, eps_clip=eps_clip, vf_coef=vf_coef, ent_coef=ent_coef,
                           max_grad_norm=max_grad_norm, norm_adv=advantage_normalization, value_clip=value_clip,
                           rew_norm=reward_normalization, dual_clip=dual_clip)
        self._recompute_adv = bool(recompute_advantage)
        optim_RL = optim[0] if isinstance(optim, (list, tuple)) else optim
        self._read_optim_hyper()
        # bind the modules' parameters into one flat device buffer (layout of include/cirs_hip.h)
        net = actor.preprocess
        assert critic.preprocess is net, "CIRS shares the trunk between actor and critic (CIRS-RL-kuaishou.py:245-247)"
        lin1, lin2 = net.model.model[0], net.model.model[2]
        head_a, head_c = actor.last.model[0], critic.last.model[0]
        self.n_items, self.dim_state, self.hidden = head_a.out_features, lin1.in_features, lin1.out_features
        dev = torch.device("cuda")
        init = {"actor.preprocess.model.model.0.weight": lin1.weight.data, "actor.preprocess.model.model.0.bias": lin1.bias.data,
                "actor.preprocess.model.model.2.weight": lin2.weight.data, "actor.preprocess.model.model.2.bias": lin2.bias.data,
                "actor.last.model.0.weight": head_a.weight.data, "actor.last.model.0.bias": head_a.bias.data,
                "critic.last.model.0.weight": head_c.weight.data, "critic.last.model.0.bias": head_c.bias.data}
        self.flat, self.views = flat_policy_params(self.n_items, self.dim_state, self.hidden, device=dev, init=init)
        for mod, pre in ((lin1, "actor.preprocess.model.model.0"), (lin2, "actor.preprocess.model.model.2"),
                         (head_a, "actor.last.model.0"), (head_c, "critic.last.model.0")):
            mod.weight.data = self.views[pre + ".weight"]
            mod.bias.data = self.views[pre + ".bias"]
        self._dev_policy = DevicePolicy(self.views, self.n_items, dim_state=self.dim_state, hidden=self.hidden, device=dev)
        self._learner: Optional[DeviceLearner] = None
        # plain attributes (NOT sub-modules: the reference's policy does not own the tracker, CIRS-RL-kuaishou.py:267-285)
        self.__dict__["_tracker"] = None   # set by the Collector (preprocess_fn's owner)
        self.__dict__["_train_n_env"] = None
        self.seed = int(torch.initial_seed() & 0x7FFFFFFF)
        # checkpoints: optim[i].state_dict() / load_state_dict() carry the device Adam state (CIRS-RL-kuaishou.py:340-358)
        from cirs_hip import optim_bridge
        self._restored_RL = None  # Adam state loaded before the learner exists
        optim_bridge.bind(optim_RL, self._adam_state_RL, flat=self.flat)
        if isinstance(optim, (list, tuple)) and len(optim) > 1:
            optim_bridge.bind(optim[1], self._adam_state_tracker)

    def _read_optim_hyper(self):
        """lr / betas / eps of both torch optimisers are read on EVERY update (a scheduler or the user may change
        param_groups between updates); what the device Adam does not implement is refused, not ignored."""
        optim = self.optim
        opts = list(optim) if isinstance(optim, (list, tuple)) else [optim]
        for o in opts:
            for g in o.param_groups:
                if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
                    raise NotImplementedError("the device Adam implements torch.optim.Adam(lr, betas, eps) only: weight_decay / amsgrad / maximize are not built")
        g = opts[0].param_groups[0]
        self._hyper.update(lr=float(g["lr"]), betas=tuple(g.get("betas", (0.9, 0.999))), adam_eps=float(g.get("eps", 1e-8)))
        gt = opts[1].param_groups[0] if len(opts) > 1 else g
        self._tracker_lr, self._tracker_betas, self._tracker_eps = float(gt["lr"]), tuple(gt.get("betas", (0.9, 0.999))), float(gt.get("eps", 1e-8))

    def _adam_state_RL(self, create=False):
        ln = self._learner
        if ln is None:
            if not create:
                return None
            # resume before the first update: keep the moments until the learner is built
            m, v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
            self._restored_RL = dict(m=m, v=v, opt_step=0)
            holder = self._restored_RL
            trunk_end = self.views["actor.last.model.0.weight"].data_ptr() - self.flat.data_ptr()

            def set_steps(steps):
                heads = [s for off, s in steps if off * 4 >= trunk_end]
                holder["opt_step"] = heads[0] if heads else 0
            return dict(flat=self.flat, m=m, v=v, steps=lambda off: 0, set_steps=set_steps)
        trunk_floats = (self.views["actor.last.model.0.weight"].data_ptr() - self.flat.data_ptr()) // 4

        def set_steps(steps):
            heads = [s for off, s in steps if off >= trunk_floats]
            if heads:
                ln.opt_step = heads[0]
        # the shared trunk appears twice in the reference's parameter list: two Adam sub-steps per optimiser step
        return dict(flat=self.flat, m=ln.adam_m, v=ln.adam_v, steps=lambda off: (2 if off < trunk_floats else 1) * ln.opt_step,
                    set_steps=set_steps)

    def _adam_state_tracker(self, create=False):
        trk = self._tracker
        if trk is None:
            return None
        # the Adam moments are shared by every engine of the tracker (StateTrackerTransformer._train_state); a 1-env engine keyed by
        # the tracker itself creates them when no collector has built one yet -- never a full-size owner-less engine (it would
        # allocate K/V caches nobody uses and redirect the per-step build_state protocol to another env count)
        if trk._train_state is None:
            trk.engine(1, owner=trk)
        _, _, _, adam_m, adam_v = trk._train_state

        def set_steps(steps):
            if steps:
                trk.adam_steps = steps[0][1]
                for eng in trk._engines.values():
                    eng.adam_steps = trk.adam_steps
        return dict(flat=trk.flat, m=adam_m, v=adam_v, steps=lambda off: trk.adam_steps, set_steps=set_steps)

    # ---- protocol pieces the Collector / trainer call ----------------------------------------------------------------
    def device_policy(self) -> DevicePolicy:
        return self._dev_policy

    def map_action(self, act):
        return act  # action_scaling=False, action_bound_method="" for KuaishouEnv (CIRS-RL-kuaishou.py:283-284)

    def exploration_noise(self, act, batch):
        return act

    def forward(self, batch, buffer=None, remove_recommended_ids=False, state=None, **kwargs):
        """The per-step protocol (core/policy/ppo.py:111-163): one cirs_actor_sample launch over the rows of `batch`.
        remove_recommended_ids: the ids every live env has already recommended in its running episode are read back from the
        buffer exactly like core/policy/utils.py:7-27 does (walk `buffer.prev` from the last index of every unfinished sub-buffer)
        and masked through the sampler's visited bitmap (the reference drops them from the probability vector and renormalises,
        :30-58 -- the same distribution).  Rows are the live envs in ascending order, as in the reference."""
        obs = batch.obs
        obs = obs if isinstance(obs, torch.Tensor) else torch.as_tensor(np.asarray(obs), dtype=torch.float32)
        obs = obs.to(self.flat.device, torch.float32).contiguous()
        n = obs.shape[0]
        visited = env_ids = None
        if remove_recommended_ids and buffer is not None and len(buffer) > 0:
            live = buffer.last_index[~np.asarray(buffer.done)[buffer.last_index] & (buffer._lengths > 0)]
            assert len(live) == n, "rows of the batch must be the unfinished envs of the buffer (core/policy/utils.py:11)"
            words = (self.n_items + 31) // 32
            bm = np.zeros((n, words), dtype=np.uint32)
            idx = live.copy()
            acts = np.asarray(buffer.act)
            while True:
                a = acts[idx].astype(np.int64)
                np.bitwise_or.at(bm, (np.arange(n), a >> 5), (np.uint32(1) << (a & 31).astype(np.uint32)))
                prv = buffer.prev(idx)
                if np.all(prv == idx):
                    break
                idx = prv
            visited = torch.as_tensor(bm.view(np.int32)).to(self.flat.device)
            env_ids = torch.arange(n, dtype=torch.int32, device=self.flat.device)   # bitmap rows = batch rows
        self._step_counter = getattr(self, "_step_counter", 0) + 1
        act, logp, value = self._dev_policy.sample(obs, seed=self.seed, rng_step=self._step_counter & 0xFFFFFFFF, env_ids=env_ids, visited=visited)
        return Batch(logits=None, act=act, state=None, dist=None, policy=Batch(logp=logp, value=value))

    def _get_learner(self, n_env, max_turn):
        if self._learner is None or self._learner.n_env != n_env or self._learner.max_turn != max_turn:
            h = self._hyper
            rms = None if self._learner is None else self._learner.rms_state
            self._learner = DeviceLearner(self.flat, self.n_items, n_env, max_turn, dim_state=self.dim_state, hidden=self.hidden,
                                          gamma=h["gamma"], gae_lambda=h["gae_lambda"], eps_clip=h["eps_clip"], vf_coef=h["vf_coef"],
                                          ent_coef=h["ent_coef"], max_grad_norm=h["max_grad_norm"], lr=h["lr"], norm_adv=h["norm_adv"],
                                          value_clip=h["value_clip"], rew_norm=h["rew_norm"], betas=h["betas"], adam_eps=h["adam_eps"],
                                          dual_clip=h["dual_clip"])
            T1, S = max_turn, self.dim_state     # critic over every stored state obs[t, b], t < T (recompute_advantage, ppo.py:176-177)
            self._learner.value_fn = lambda traj: self._dev_policy.values(traj.obs.view(-1, S), n=T1 * n_env, value_out=traj.value.view(-1))
            if rms is not None:
                self._learner.rms_state.copy_(rms)
            if self._restored_RL is not None:  # optimiser state restored from a checkpoint before the first update
                self._learner.adam_m.copy_(self._restored_RL["m"]); self._learner.adam_v.copy_(self._restored_RL["v"])
                self._learner.opt_step = self._restored_RL["opt_step"]
                self._restored_RL = None
        return self._learner

    def update(self, sample_size: int, buffer, batch_size: int = 1024, repeat: int = 2, perms=None, **kwargs) -> Dict[str, List[float]]:
        """BasePolicy.update (base.py:219-244): process_fn + learn on the whole buffer, then the tracker's Adam step."""
        if buffer is None:
            return {}
        assert sample_size == 0, "on-policy: the whole buffer is used (onpolicy.py:199-201)"
        ro = getattr(buffer, "_rollout", None)
        assert ro is not None and buffer._traj is ro.traj, "update() consumes a buffer filled by Collector.collect()"
        self.updating = True
        lens = np.asarray(buffer._lengths, dtype=np.int32)
        self._read_optim_hyper()
        ln = self._get_learner(ro.env.n_env, ro.env.max_turn)
        ln.cfg.lr, (ln.cfg.beta1, ln.cfg.beta2), ln.cfg.adam_eps = self._hyper["lr"], self._hyper["betas"], self._hyper["adam_eps"]
        ln.perm_seed = (self.seed * 7919 + 20230) & 0x7FFFFFFF
        n = ln.prepare(ro.traj, lens)
        losses = ln.learn(batch_size, repeat, perms=perms, want_tracker_grad=self._tracker is not None, recompute_adv=self._recompute_adv)
        if self._tracker is not None:
            eng = ro.tracker      # the slots / caches of THIS buffer's rollout (each Collector owns its engine)
            eng.lr, eng.betas, eng.adam_eps = self._tracker_lr, self._tracker_betas, self._tracker_eps
            eng.adam_steps = self._tracker.adam_steps
            offsets = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
            dev = self.flat.device
            eng.backward(buffer._users, ro.traj, ln.b_env, ln.b_t, torch.as_tensor(offsets).to(dev), torch.as_tensor(lens).to(dev), n, ln.dobs)
            eng.adam_update()
            self._tracker.adam_steps = eng.adam_steps
        if self.lr_scheduler is not None:      # ppo.py:239-240: stepped once per learn(); its optimisers are re-read next update
            self.lr_scheduler.step()
        self.updating = False
        lo = losses.cpu().numpy()
        return {"loss": lo[:, 0].tolist(), "loss/clip": lo[:, 1].tolist(), "loss/vf": lo[:, 2].tolist(), "loss/ent": lo[:, 3].tolist()}
