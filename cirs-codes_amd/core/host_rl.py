"""The RL loop of BASELINE configs[0] (VirtualTaobao, 4 parallel envs, CPU plumbing, no GPU) on the host.

The reference runs this configuration on the CPU through the same Python classes as the Kuaishou one (CIRS-RL-taobao.py:192-300):
dense-feature state tracker, ActorProb / Critic with an Independent(Normal) policy, PPO with action scaling, a per-step Collector.
None of it is on the MI355X hot path; the mirror's device-backed classes dispatch here when they are built for VirtualTB-v0:

  HostStateTracker    core/state_tracker.py:89-115 (dense branch: features pass through), :129-250 (causal transformer re-run over
                      the whole prefix at every build_state, retained autograd graph, nn.Dropout live in training mode)
  HostPPOPolicy       behaviour of core/policy/ppo.py:96-246 + tianshou/policy/base.py:179-313,380-396 (map_action, update, GAE) +
                      modelfree/a2c.py:80-109 (returns, running return statistics) for any torch distribution, written in this repo's
                      own three-stage form (ReturnScale / lambda_returns -> ppo_objective -> optimiser schedule in learn())
  HostCollector       core/collector.py:147-367 (per-step loop, finished envs dropped, result dict)

Plain PyTorch on the host, in the reference's order of operations and of random draws: with the same seeds the collect and the update
reproduce the reference's (tests/test_c1_rl_cpu.py, fixture recorded from the reference by oracle/gen_golden.py gen_c1rl)."""
import dataclasses
import math
import time
from typing import Any, Callable, Dict, List, Optional

import numpy as np
import torch
from torch import nn

from tianshou.data import Batch, VectorReplayBuffer, to_numpy, to_torch_as


# ------------------------------------------------------------------------------------------------------------------------------
# state tracker
# ------------------------------------------------------------------------------------------------------------------------------
class _PositionalEncoding(nn.Module):
    """x + pe[:len] followed by dropout; sin on the even columns, cos on the odd ones (an odd width drops the last cos column)."""

    def __init__(self, d_model, dropout=0.1, max_len=100):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term)
        n_odd = pe[:, 0, 1::2].shape[-1]
        pe[:, 0, 1::2] = torch.cos(position * div_term)[:, :n_odd]
        self.register_buffer("pe", pe)

    def forward(self, x):
        return self.dropout(x + self.pe[:x.size(0)])


class HostStateTracker(nn.Module):
    def __init__(self, user_columns, action_columns, feedback_columns, dim_model, dim_state, dim_max_batch, dropout=0.1,
                 dataset="VirtualTB-v0", has_user_embedding=True, has_action_embedding=True, has_feedback_embedding=False,
                 nhead=8, d_hid=128, nlayers=2, device="cpu", seed=2021, init_std=0.0001, padding_idx=None, MAX_TURN=100):
        super().__init__()
        from core.user_model import compute_input_dim
        assert has_user_embedding and has_action_embedding and has_feedback_embedding, \
            "the host tracker serves the dense-feature configuration (VirtualTB-v0: every column is passed through as is)"
        self.dataset, self.device = dataset, torch.device("cpu")
        self.dim_model, self.dim_state, self.MAX_TURN = dim_model, dim_state, MAX_TURN + 1
        self.user_columns, self.action_columns, self.feedback_columns = user_columns, action_columns, feedback_columns
        self.embedding_dict = nn.ModuleDict()       # no sparse column in this configuration
        self.ffn_user = nn.Linear(compute_input_dim(user_columns), dim_model)
        self.fnn_gate = nn.Linear(1 + compute_input_dim(action_columns), dim_model)
        self.sigmoid = nn.Sigmoid()
        self.pos_encoder = _PositionalEncoding(dim_model, dropout, max_len=self.MAX_TURN)
        layer = nn.TransformerEncoderLayer(dim_model, nhead, d_hid, dropout)
        self.transformer_encoder = nn.TransformerEncoder(layer, nlayers)
        self.decoder = nn.Linear(dim_model, dim_state)
        self.decoder.bias.data.zero_()
        self.decoder.weight.data.uniform_(-0.1, 0.1)
        self.data, self.len_data = None, None

    def forward(self, src0, src_mask):
        out = self.transformer_encoder(self.pos_encoder(src0 * math.sqrt(self.dim_model)), src_mask)
        return self.decoder(out[-1, :, :])

    @staticmethod
    def _dense(x):
        return torch.as_tensor(np.asarray(x), dtype=torch.float32)

    def _causal_mask(self, length):
        return torch.triu(torch.ones(length, length) * float("-inf"), diagonal=1)

    def build_state(self, obs=None, env_id=None, obs_next=None, rew=None, done=None, info=None, policy=None, dim_batch=None,
                    reset=False):
        if reset and dim_batch:
            self.data = torch.zeros(self.MAX_TURN, dim_batch, self.dim_model)     # (length, batch, dim)
            self.len_data = torch.zeros(dim_batch, dtype=torch.int64)
            return
        if obs is not None:            # slot 0: the user (the last three entries of a VirtualTB observation are not user features)
            e_u = self._dense(np.asarray(obs)[:, :-3])
            self.len_data[env_id] = 1
            self.data[0, env_id, :] = self.ffn_user(e_u)
            return {"obs": self.forward(self.data[:1, env_id, :], self._causal_mask(1))}
        if obs_next is not None:       # append the gated action of this step
            a_t = self._dense(np.asarray(obs_next)[:, :-3])
            self.len_data[env_id] += 1
            length = int(self.len_data[env_id[0]])
            r_t = self._dense(np.asarray(rew).reshape((-1, 1)))
            g_t = self.sigmoid(self.fnn_gate(torch.cat((r_t, a_t), -1)))
            self.data[length - 1, env_id, :] = g_t * a_t
            return {"obs_next": self.forward(self.data[:length, env_id, :], self._causal_mask(length))}
        return {}


# ------------------------------------------------------------------------------------------------------------------------------
# PPO (written as three explicit stages, the way csrc/ppo.hip is organised: returns stage -> row objective -> optimiser schedule)
# ------------------------------------------------------------------------------------------------------------------------------
class ReturnScale:
    """Running mean / variance / weight of the un-normalised returns (parallel-variance merge of one block per update); the stored
    variance of the PREVIOUS updates scales the critic's outputs and the regression targets of the current one."""

    def __init__(self):
        self.mean, self.var, self.count = 0.0, 1.0, float(np.finfo(np.float32).eps)

    def scale(self, floor):
        return float(np.sqrt(self.var + floor))

    def update(self, block):
        block = np.asarray(block, dtype=np.float64)
        n, mu, s2 = block.shape[0], block.mean(axis=0), block.var(axis=0)
        w_old, w_all = self.count, self.count + n
        shift = mu - self.mean
        self.var = (self.var * w_old + s2 * n + shift * shift * (w_old * n / w_all)) / w_all
        self.mean = self.mean + shift * (n / w_all)
        self.count = w_all


@dataclasses.dataclass
class PpoHyper:
    """The knobs of one PPO update (constructor keywords of the policy class on the left of each field's comment)."""
    clip: float = 0.2                  # eps_clip
    dual: Optional[float] = None       # dual_clip (> 1)
    clip_value: bool = False           # value_clip
    whiten_adv: bool = True            # advantage_normalization
    refresh_adv: bool = False          # recompute_advantage
    c_value: float = 0.5               # vf_coef
    c_entropy: float = 0.01            # ent_coef
    max_norm: Optional[float] = None   # max_grad_norm
    lam: float = 0.95                  # gae_lambda
    discount: float = 0.99             # discount_factor
    chunk: int = 256                   # max_batchsize: rows per no-grad network pass
    scale_returns: bool = False        # reward_normalization
    floor: float = 1e-8


def row_ranges(n, size):
    """[0, n) cut into consecutive ranges of `size` rows; a short tail joins the range before it."""
    cuts = list(range(0, n, size)) + [n]
    if len(cuts) > 2 and n % size:
        del cuts[-2]
    return list(zip(cuts[:-1], cuts[1:]))


def lambda_returns(value, value_next, reward, boundary, discount, lam):
    """Generalised advantage estimates of a buffer-ordered slice, float64.  boundary[i]: row i closes its episode segment (terminal, or
    the newest row of an unfinished episode), so nothing is carried across it; value_next is already zero behind a terminal row."""
    td = reward + discount * value_next - value
    carry = np.where(boundary, 0.0, discount * lam)
    out = np.empty_like(td)
    run = 0.0
    for i in reversed(range(td.shape[0])):
        run = td[i] + carry[i] * run
        out[i] = run
    return out


def ppo_objective(logp, logp_old, adv, value, value_old, target, entropy, h: PpoHyper):
    """Scalar PPO objective of a minibatch and its three reported terms (all means over the rows).
    policy: pessimistic (clipped) importance-weighted advantage, optionally floored by dual * adv; value: squared error against the
    regression target, optionally the worse of the free and the trust-region-clipped prediction; entropy bonus."""
    if h.whiten_adv:
        adv = (adv - adv.mean()) / adv.std()
    w = torch.exp(logp - logp_old).float().reshape(-1)
    gain = torch.minimum(w * adv, torch.clamp(w, 1.0 - h.clip, 1.0 + h.clip) * adv)
    if h.dual:
        gain = torch.maximum(gain, h.dual * adv)
    policy_term = -gain.mean()
    err = (target - value) ** 2
    if h.clip_value:
        bounded = value_old + torch.clamp(value - value_old, -h.clip, h.clip)
        err = torch.maximum(err, (target - bounded) ** 2)
    value_term = err.mean()
    entropy_term = entropy.mean()
    return policy_term + h.c_value * value_term - h.c_entropy * entropy_term, policy_term, value_term, entropy_term


class HostPPOPolicy(nn.Module):
    def __init__(self, actor, critic, optim, dist_fn, eps_clip=0.2, dual_clip=None, value_clip=False, advantage_normalization=True,
                 recompute_advantage=False, vf_coef=0.5, ent_coef=0.01, max_grad_norm=None, gae_lambda=0.95, max_batchsize=256,
                 discount_factor=0.99, reward_normalization=False, action_scaling=True, action_bound_method="clip",
                 deterministic_eval=False, action_space=None, lr_scheduler=None, observation_space=None, **kwargs):
        super().__init__()
        assert dual_clip is None or dual_clip > 1.0
        self.actor, self.critic, self.optim, self.dist_fn = actor, critic, optim, dist_fn
        self.action_space = action_space
        self.action_type = "continuous" if hasattr(action_space, "low") and np.asarray(action_space.low).dtype.kind == "f" else "discrete"
        self.action_scaling = action_scaling and self.action_type == "continuous"
        self.action_bound_method = action_bound_method if self.action_type == "continuous" else ""
        self.hyper = PpoHyper(clip=eps_clip, dual=dual_clip, clip_value=bool(value_clip), whiten_adv=bool(advantage_normalization),
                              refresh_adv=bool(recompute_advantage), c_value=vf_coef, c_entropy=ent_coef, max_norm=max_grad_norm,
                              lam=gae_lambda, discount=discount_factor, chunk=max_batchsize, scale_returns=bool(reward_normalization))
        self.ret_rms = ReturnScale()
        self._deterministic_eval = deterministic_eval
        self.lr_scheduler = lr_scheduler
        self.updating = False
        self.callbacks: List[Any] = []

    # ---- acting -----------------------------------------------------------------------------------------------------------
    def _distribution(self, obs, state=None):
        head, hidden = self.actor(obs, state=state)
        return (self.dist_fn(*head) if isinstance(head, tuple) else self.dist_fn(head)), head, hidden

    def _distribution_as_forward(self, obs):
        """The action distribution of `obs`, leaving torch's generator where a whole forward() call leaves it.  forward() draws an action even
        when only its distribution is wanted, and the reference's process_fn / learn go through forward (core/policy/ppo.py:107,183): a run that
        seeds once must find the generator in the same place after an update, or every later collect samples other actions and other masks."""
        dist = self._distribution(obs)[0]
        if not (self._deterministic_eval and not self.training):
            dist.sample()
        return dist

    def forward(self, batch, buffer=None, remove_recommended_ids=False, state=None, **kwargs):
        assert not remove_recommended_ids, "id masking is a discrete-catalogue feature (KuaishouEnv)"
        dist, head, hidden = self._distribution(batch.obs, state)
        greedy = self._deterministic_eval and not self.training
        if not greedy:
            chosen = dist.sample()
        else:       # mode of the distribution: arg-max logit / the mean of (mu, sigma)
            chosen = head.argmax(-1) if self.action_type == "discrete" else head[0]
        return Batch(logits=head, act=chosen, state=hidden, dist=dist)

    _SQUASH = {"clip": lambda a: np.minimum(np.maximum(a, -1.0), 1.0), "tanh": np.tanh}

    def map_action(self, act):
        """Raw network action -> [-1, 1] (clip or tanh) -> the env's box [low, high]; discrete actions pass through."""
        if self.action_type != "continuous" or not isinstance(act, np.ndarray):
            return act
        squash = self._SQUASH.get(self.action_bound_method)
        unit = squash(act) if squash else act
        if not self.action_scaling:
            return unit
        box = self.action_space
        return box.low + (box.high - box.low) * (unit + 1.0) / 2.0

    def exploration_noise(self, act, batch):
        return act

    # ---- learning: stage 1, returns -----------------------------------------------------------------------------------------
    def _critic_rows(self, rows):
        """critic(rows) without a graph, `chunk` rows per pass, as one flat tensor."""
        with torch.no_grad():
            parts = [self.critic(rows[a:b]) for a, b in row_ranges(len(rows), self.hyper.chunk)]
        return torch.cat(parts, dim=0).flatten()

    def _returns_stage(self, batch, buffer, rows):
        """Fills batch.v_s (critic of the stored states, network scale), batch.adv and batch.returns for the sampled buffer rows.
        Values live on the normalised-return scale: they are multiplied by the running scale before the recurrence, and the targets
        are divided by the SAME (pre-update) scale afterwards; then this update's returns enter the running statistics."""
        h = self.hyper
        batch.v_s = self._critic_rows(batch.obs)
        now = batch.v_s.cpu().numpy().astype(np.float64)
        nxt = self._critic_rows(batch.obs_next).cpu().numpy().astype(np.float64)
        unit = self.ret_rms.scale(h.floor) if h.scale_returns else 1.0
        terminal = np.asarray(buffer.done)[rows].astype(bool)
        nxt = np.where(terminal, 0.0, nxt * unit)
        boundary = np.asarray(batch.done).astype(bool) | np.isin(rows, buffer.unfinished_index())
        adv = lambda_returns(now * unit, nxt, np.asarray(batch.rew, dtype=np.float64), boundary, h.discount, h.lam)
        total = adv + now * unit
        if h.scale_returns:
            self.ret_rms.update(total)
        batch.returns = to_torch_as(total / unit, batch.v_s)
        batch.adv = to_torch_as(adv, batch.v_s)
        return batch

    def process_fn(self, batch, buffer, indice):
        """Protocol hook of the trainer: returns stage + the behaviour policy's log-probabilities of the stored actions."""
        self._sampled = (buffer, indice)
        batch = self._returns_stage(batch, buffer, indice)
        batch.act = to_torch_as(batch.act, batch.v_s)      # stored actions as a tensor of the critic's dtype / device
        with torch.no_grad():                                # behaviour log-probabilities carry no graph
            parts = [self._distribution_as_forward(batch.obs[a:b]).log_prob(batch.act[a:b]) for a, b in row_ranges(len(batch), self.hyper.chunk)]
        batch.logp_old = torch.cat(parts, dim=0)
        return batch

    # ---- learning: stages 2 + 3, objective and optimiser schedule -------------------------------------------------------------
    def learn(self, batch, batch_size, repeat, **kwargs) -> Dict[str, List[float]]:
        """`repeat` passes over the batch in shuffled minibatches of `batch_size` rows (a short tail joins the last one).  Two optimisers
        with two clocks: the policy optimiser (actor + critic) steps on every minibatch with the clipped gradient; the tracker optimiser
        is cleared at the start of every pass and steps ONCE at the end of the update, i.e. on what the last pass accumulated through the
        stored states' graph (which is why every backward keeps that graph)."""
        h = self.hyper
        opt_policy, opt_tracker = self.optim
        clipped = [p for m in (self.actor, self.critic) for p in m.parameters()]
        report = {"loss": [], "loss/clip": [], "loss/vf": [], "loss/ent": []}
        n = len(batch)
        for sweep in range(repeat):
            opt_tracker.zero_grad()
            if h.refresh_adv and sweep:
                batch = self._returns_stage(batch, *self._sampled)
            order = np.random.permutation(n)
            for a, b in row_ranges(n, batch_size):
                mb = batch[order[a:b]]
                dist = self._distribution_as_forward(mb.obs)
                total, p_term, v_term, e_term = ppo_objective(dist.log_prob(mb.act), mb.logp_old, mb.adv, self.critic(mb.obs).flatten(),
                                                              mb.v_s, mb.returns, dist.entropy(), h)
                opt_policy.zero_grad()
                total.backward(retain_graph=True)
                if h.max_norm:
                    nn.utils.clip_grad_norm_(clipped, max_norm=h.max_norm)
                opt_policy.step()
                for key, val in zip(report, (total, p_term, v_term, e_term)):
                    report[key].append(val.item())
        opt_tracker.step()
        if self.lr_scheduler:
            self.lr_scheduler.step()
        return report

    def update(self, sample_size, buffer, **kwargs):
        if buffer is None:
            return dict()
        batch, rows = buffer.sample(sample_size)
        self.updating = True       # (flag read by exploration wrappers)
        try:
            return self.learn(self.process_fn(batch, buffer, rows), **kwargs)
        finally:
            self.updating = False  # also when the learner raises


# ------------------------------------------------------------------------------------------------------------------------------
# collector
# ------------------------------------------------------------------------------------------------------------------------------
class HostCollector:
    """Per-step collector over host vector envs.  Public surface = what the trainer, test_episode and the scripts touch (call shapes of
    core/collector.py:40-147): the constructor keywords, `collect(n_episode=)`, the three reset hooks, `buffer`, the three running counters.
    Inside, the envs of one collect() call form a `_Cohort` (below); the collector itself only keeps the states the tracker produced at reset."""

    def __init__(self, policy, env, buffer: Optional[VectorReplayBuffer] = None, preprocess_fn: Optional[Callable[..., Any]] = None,
                 exploration_noise: bool = False, remove_recommended_ids=False, force_length=0):
        self.policy, self.env, self.preprocess_fn = policy, env, preprocess_fn
        self.env_num = len(env)
        self.options = dict(noise=bool(exploration_noise), mask_seen=remove_recommended_ids, horizon=int(force_length))
        self.buffer = VectorReplayBuffer(self.env_num, self.env_num) if buffer is None else buffer
        assert self.buffer.buffer_num >= self.env_num, "one sub-buffer per env"
        self.reset()

    # the reference's attribute names, for callers that read them back
    exploration_noise = property(lambda self: self.options["noise"])
    remove_recommended_ids = property(lambda self: self.options["mask_seen"])
    force_length = property(lambda self: self.options["horizon"])

    def reset_stat(self):
        self.collect_step = self.collect_episode = 0
        self.collect_time = 0.0

    def reset_buffer(self, keep_statistics=False):
        self.buffer = type(self.buffer)(self.buffer.maxsize, self.buffer.buffer_num)      # every collect fills a brand-new buffer

    def reset_env(self):
        """Fresh episodes everywhere; with a tracker attached the stored observation is the tracker's state of the reset observation."""
        track = self.preprocess_fn
        if track:
            track(dim_batch=self.env_num, reset=True)
        first = self.env.reset()
        if track:
            first = track(obs=first, env_id=np.arange(self.env_num)).get("obs", first)
        self.front = first

    def reset(self):
        for hook in (self.reset_env, self.reset_buffer, self.reset_stat):
            hook()

    def collect(self, n_step=None, n_episode=None, random=False, render=None, no_grad=True) -> Dict[str, Any]:
        assert n_step is None and n_episode is not None and n_episode > 0, "the CIRS scripts collect whole episodes (n_episode)"
        self.reset()       # states rebuilt by the tracker from the updated parameters (core/collector.py:200)
        cohort = _Cohort(self, n_episode)
        clock = time.time()
        while cohort.episodes < n_episode:
            actions, extra = cohort.decide(random, no_grad)
            cohort.advance(actions, extra)
        elapsed = max(time.time() - clock, 1e-9)
        self.front = cohort.obs
        self.collect_time += elapsed
        self.collect_step += cohort.transitions
        self.collect_episode += cohort.episodes
        return cohort.summary()


class _Cohort:
    """The envs still playing inside one HostCollector.collect() call.  Plain arrays per field (ids, current states); a Batch is only assembled
    where a protocol asks for one (the policy call, ReplayBuffer.add).  Finished envs are never reset: they leave once the episode quota is
    covered by the envs that remain."""

    def __init__(self, host: HostCollector, quota: int):
        n = min(host.env_num, quota)
        self.host, self.quota = host, quota
        self.ids, self.obs = np.arange(n), host.front[:n]
        self.turn = self.transitions = self.episodes = 0
        self.closed: List[tuple] = []      # (returns, lengths, first buffer rows) of the episodes each vector step closed, in completion order

    def decide(self, random: bool, no_grad: bool):
        """-> (raw actions [n, ...] as numpy, the policy's own per-step record or an empty Batch)."""
        host = self.host
        if random:
            return np.stack([host.env.action_space[i].sample() for i in self.ids]), Batch()
        view = Batch(obs=self.obs, info=Batch())
        with torch.set_grad_enabled(not no_grad):
            out = host.policy(view, host.buffer, state=None, remove_recommended_ids=host.options["mask_seen"])
        raw = to_numpy(out.act)
        if host.options["noise"]:
            raw = host.policy.exploration_noise(raw, view)
        return raw, (out.get("policy", Batch()) or Batch())

    def advance(self, raw, extra):
        host = self.host
        nxt, rew, done, info = host.env.step(host.policy.map_action(raw), self.ids)
        self.turn += 1
        if host.options["horizon"] > 0:      # fixed-horizon evaluation: the env's own exit decision is overridden
            done = np.full(len(self.ids), self.turn >= host.options["horizon"])
        row = dict(obs=self.obs, act=raw, rew=rew, done=done, obs_next=nxt, info=info, policy=extra)
        if host.preprocess_fn:               # the tracker turns (next observation, reward) into the next state
            row.update(host.preprocess_fn(env_id=self.ids, **{k: row[k] for k in ("obs_next", "rew", "done", "info", "policy")}))
        _, ep_return, ep_length, ep_first = host.buffer.add(Batch(**row), buffer_ids=self.ids)
        self.transitions += len(self.ids)
        self.obs = row["obs_next"]
        over = np.flatnonzero(np.asarray(row["done"], dtype=bool))
        if len(over):
            self.episodes += len(over)
            self.closed.append((ep_return[over], ep_length[over], ep_first[over]))
            surplus = len(self.ids) - max(self.quota - self.episodes, 0)
            if surplus > 0:
                keep = np.setdiff1d(np.arange(len(self.ids)), over[:surplus])
                self.ids, self.obs = self.ids[keep], self.obs[keep]

    def summary(self) -> Dict[str, Any]:
        returns, lengths, firsts = (np.concatenate(col) for col in zip(*self.closed))
        stats = {"rews": returns, "lens": lengths, "idxs": firsts, "n/st": self.transitions, "n/ep": self.episodes}
        for name, arr in (("rew", returns), ("len", lengths)):
            stats[name], stats[name + "_std"] = arr.mean(), arr.std()
        return stats
