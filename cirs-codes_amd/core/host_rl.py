"""The RL loop of BASELINE configs[0] (VirtualTaobao, 4 parallel envs, CPU plumbing, no GPU) on the host.

The reference runs this configuration on the CPU through the same Python classes as the Kuaishou one (CIRS-RL-taobao.py:192-300):
dense-feature state tracker, ActorProb / Critic with an Independent(Normal) policy, PPO with action scaling, a per-step Collector.
None of it is on the MI355X hot path; the mirror's device-backed classes dispatch here when they are built for VirtualTB-v0:

  HostStateTracker    core/state_tracker.py:89-115 (dense branch: features pass through), :129-250 (causal transformer re-run over
                      the whole prefix at every build_state, retained autograd graph, nn.Dropout live in training mode)
  HostPPOPolicy       core/policy/ppo.py:96-246 + tianshou/policy/base.py:179-313,380-396 (map_action, update, GAE) +
                      modelfree/a2c.py:80-109 (_compute_returns, RunningMeanStd) for any torch distribution
  HostCollector       core/collector.py:147-367 (per-step loop, finished envs dropped, result dict)

Plain PyTorch on the host, in the reference's order of operations and of random draws: with the same seeds the collect and the update
reproduce the reference's (tests/test_c1_rl_cpu.py, fixture recorded from the reference by oracle/gen_golden.py gen_c1rl)."""
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
# PPO
# ------------------------------------------------------------------------------------------------------------------------------
class _RunningMeanStd:
    def __init__(self):
        self.mean, self.var, self.count = 0.0, 1.0, np.finfo(np.float32).eps.item()

    def update(self, x):
        b_mean, b_var, b_count = np.mean(x, axis=0), np.var(x, axis=0), len(x)
        delta, total = b_mean - self.mean, self.count + b_count
        m2 = self.var * self.count + b_var * b_count + delta ** 2 * self.count * b_count / total
        self.mean, self.var, self.count = self.mean + delta * b_count / total, m2 / total, total


def _gae(v_s, v_s_, rew, end_flag, gamma, gae_lambda):
    out = np.zeros(rew.shape)
    delta = rew + v_s_ * gamma - v_s
    m = (1.0 - end_flag) * (gamma * gae_lambda)
    gae = 0.0
    for i in range(len(rew) - 1, -1, -1):
        gae = delta[i] + m[i] * gae
        out[i] = gae
    return out


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
        self._eps_clip, self._dual_clip, self._value_clip = eps_clip, dual_clip, bool(value_clip)
        self._norm_adv, self._recompute_adv = bool(advantage_normalization), bool(recompute_advantage)
        self._weight_vf, self._weight_ent, self._grad_norm = vf_coef, ent_coef, max_grad_norm
        self._lambda, self._gamma, self._batch = gae_lambda, discount_factor, max_batchsize
        self._rew_norm, self._deterministic_eval = bool(reward_normalization), deterministic_eval
        self.ret_rms, self._eps = _RunningMeanStd(), 1e-8
        self.lr_scheduler = lr_scheduler
        self.updating = False
        self.callbacks: List[Any] = []

    # ---- acting -----------------------------------------------------------------------------------------------------------
    def forward(self, batch, buffer=None, remove_recommended_ids=False, state=None, **kwargs):
        assert not remove_recommended_ids, "id masking is a discrete-catalogue feature (KuaishouEnv)"
        logits, h = self.actor(batch.obs, state=state)
        dist = self.dist_fn(*logits) if isinstance(logits, tuple) else self.dist_fn(logits)
        if self._deterministic_eval and not self.training:
            act = logits.argmax(-1) if self.action_type == "discrete" else logits[0]
        else:
            act = dist.sample()
        return Batch(logits=logits, act=act, state=h, dist=dist)

    def map_action(self, act):
        """Bound the raw action to [-1, 1], then scale to the action space (tianshou/policy/base.py:179-210)."""
        if self.action_type == "continuous" and isinstance(act, np.ndarray):
            if self.action_bound_method == "clip":
                act = np.clip(act, -1.0, 1.0)
            elif self.action_bound_method == "tanh":
                act = np.tanh(act)
            if self.action_scaling:
                low, high = self.action_space.low, self.action_space.high
                act = low + (high - low) * (act + 1.0) / 2.0
        return act

    def exploration_noise(self, act, batch):
        return act

    # ---- learning ---------------------------------------------------------------------------------------------------------
    def _compute_returns(self, batch, buffer, indice):
        v_s, v_s_ = [], []
        with torch.no_grad():
            for b in batch.split(self._batch, shuffle=False, merge_last=True):
                v_s.append(self.critic(b.obs))
                v_s_.append(self.critic(b.obs_next))
        batch.v_s = torch.cat(v_s, dim=0).flatten()
        v_s = batch.v_s.cpu().numpy()
        v_s_ = torch.cat(v_s_, dim=0).flatten().cpu().numpy()
        if self._rew_norm:       # values are learned on the normalised scale
            scale = np.sqrt(self.ret_rms.var + self._eps)
            v_s, v_s_ = v_s * scale, v_s_ * scale
        rew = np.asarray(batch.rew, dtype=float)
        done = np.asarray(batch.done).astype(bool)
        v_s_ = v_s_ * ~np.asarray(buffer.done)[indice]                       # value mask: no bootstrap across an episode end
        end_flag = done.copy()
        end_flag[np.isin(indice, buffer.unfinished_index())] = True
        adv = _gae(v_s, v_s_, rew, end_flag.astype(float), self._gamma, self._lambda)
        unnormalized_returns = adv + v_s
        if self._rew_norm:
            batch.returns = unnormalized_returns / np.sqrt(self.ret_rms.var + self._eps)
            self.ret_rms.update(unnormalized_returns)
        else:
            batch.returns = unnormalized_returns
        batch.returns = to_torch_as(batch.returns, batch.v_s)
        batch.adv = to_torch_as(adv, batch.v_s)
        return batch

    def process_fn(self, batch, buffer, indice):
        if self._recompute_adv:
            self._buffer, self._indice = buffer, indice
        batch = self._compute_returns(batch, buffer, indice)
        batch.act = to_torch_as(batch.act, batch.v_s)
        old = []
        with torch.no_grad():
            for b in batch.split(self._batch, shuffle=False, merge_last=True):
                old.append(self(b).dist.log_prob(b.act))
        batch.logp_old = torch.cat(old, dim=0)
        return batch

    def learn(self, batch, batch_size, repeat, **kwargs) -> Dict[str, List[float]]:
        losses, clip_losses, vf_losses, ent_losses = [], [], [], []
        optim_RL, optim_state = self.optim
        params = list(self.actor.parameters()) + list(self.critic.parameters())
        for step in range(repeat):
            optim_state.zero_grad()
            if self._recompute_adv and step > 0:
                batch = self._compute_returns(batch, self._buffer, self._indice)
            for b in batch.split(batch_size, merge_last=True):
                dist = self(b).dist
                if self._norm_adv:
                    b.adv = (b.adv - b.adv.mean()) / b.adv.std()
                ratio = (dist.log_prob(b.act) - b.logp_old).exp().float()
                ratio = ratio.reshape(ratio.size(0), -1).transpose(0, 1)
                surr1 = ratio * b.adv
                surr2 = ratio.clamp(1.0 - self._eps_clip, 1.0 + self._eps_clip) * b.adv
                if self._dual_clip:
                    clip_loss = -torch.max(torch.min(surr1, surr2), self._dual_clip * b.adv).mean()
                else:
                    clip_loss = -torch.min(surr1, surr2).mean()
                value = self.critic(b.obs).flatten()
                if self._value_clip:
                    v_clip = b.v_s + (value - b.v_s).clamp(-self._eps_clip, self._eps_clip)
                    vf_loss = torch.max((b.returns - value).pow(2), (b.returns - v_clip).pow(2)).mean()
                else:
                    vf_loss = (b.returns - value).pow(2).mean()
                ent_loss = dist.entropy().mean()
                loss = clip_loss + self._weight_vf * vf_loss - self._weight_ent * ent_loss
                optim_RL.zero_grad()
                loss.backward(retain_graph=True)      # the graph of the stored states reaches back into the tracker
                if self._grad_norm:
                    nn.utils.clip_grad_norm_(params, max_norm=self._grad_norm)
                optim_RL.step()
                clip_losses.append(clip_loss.item()); vf_losses.append(vf_loss.item())
                ent_losses.append(ent_loss.item()); losses.append(loss.item())
        optim_state.step()        # the tracker moves once per update, on the gradient accumulated over the last repeat
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        return {"loss": losses, "loss/clip": clip_losses, "loss/vf": vf_losses, "loss/ent": ent_losses}

    def update(self, sample_size, buffer, **kwargs):
        if buffer is None:
            return {}
        batch, indice = buffer.sample(sample_size)
        self.updating = True
        batch = self.process_fn(batch, buffer, indice)
        result = self.learn(batch, **kwargs)
        self.updating = False
        return result


# ------------------------------------------------------------------------------------------------------------------------------
# collector
# ------------------------------------------------------------------------------------------------------------------------------
class HostCollector:
    def __init__(self, policy, env, buffer: Optional[VectorReplayBuffer] = None, preprocess_fn: Optional[Callable[..., Any]] = None,
                 exploration_noise: bool = False, remove_recommended_ids=False, force_length=0):
        self.policy, self.env, self.env_num = policy, env, len(env)
        self.preprocess_fn, self.exploration_noise = preprocess_fn, exploration_noise
        self.remove_recommended_ids, self.force_length = remove_recommended_ids, force_length
        self._action_space = env.action_space
        self.buffer = buffer if buffer is not None else VectorReplayBuffer(self.env_num, self.env_num)
        assert self.buffer.buffer_num >= self.env_num
        self.reset()

    def reset(self):
        self.data = Batch(obs=Batch(), act=Batch(), rew=Batch(), done=Batch(), obs_next=Batch(), info=Batch(), policy=Batch())
        self.reset_env()
        self.reset_buffer()
        self.reset_stat()

    def reset_stat(self):
        self.collect_step, self.collect_episode, self.collect_time = 0, 0, 0.0

    def reset_buffer(self, keep_statistics=False):
        self.buffer = VectorReplayBuffer(self.buffer.maxsize, self.buffer.buffer_num)      # a brand-new buffer per collect

    def reset_env(self):
        if self.preprocess_fn:
            self.preprocess_fn(dim_batch=self.env_num, reset=True)
        obs = self.env.reset()
        if self.preprocess_fn:
            obs = self.preprocess_fn(obs=obs, env_id=np.arange(self.env_num)).get("obs", obs)
        self.data.obs = obs

    def collect(self, n_step=None, n_episode=None, random=False, render=None, no_grad=True) -> Dict[str, Any]:
        assert n_step is None and n_episode is not None and n_episode > 0, "the CIRS scripts collect whole episodes (n_episode)"
        ready = np.arange(min(self.env_num, n_episode))
        self.reset()       # fresh observations from the updated parameters (core/collector.py:200)
        self.data = self.data[:min(self.env_num, n_episode)] if len(ready) < self.env_num else self.data
        start = time.time()
        step_count, episode_count, cnt_loop = 0, 0, 0
        ep_rews, ep_lens, ep_idxs = [], [], []
        while True:
            assert len(self.data) == len(ready)
            if random:
                self.data.update(act=np.stack([self._action_space[i].sample() for i in ready]))
            else:
                if no_grad:
                    with torch.no_grad():
                        result = self.policy(self.data, self.buffer, state=None, remove_recommended_ids=self.remove_recommended_ids)
                else:
                    result = self.policy(self.data, self.buffer, state=None, remove_recommended_ids=self.remove_recommended_ids)
                act = to_numpy(result.act)
                if self.exploration_noise:
                    act = self.policy.exploration_noise(act, self.data)
                self.data.update(policy=result.get("policy", Batch()) or Batch(), act=act)
            obs_next, rew, done, info = self.env.step(self.policy.map_action(self.data.act), ready)
            cnt_loop += 1
            if self.force_length > 0:
                done = np.full_like(done, cnt_loop >= self.force_length, dtype=bool)
            self.data.update(obs_next=obs_next, rew=rew, done=done, info=info)
            if self.preprocess_fn:
                self.data.update(self.preprocess_fn(obs_next=self.data.obs_next, rew=self.data.rew, done=self.data.done,
                                                    info=self.data.info, policy=self.data.policy, env_id=ready))
            ptr, e_rew, e_len, e_idx = self.buffer.add(self.data, buffer_ids=ready)
            step_count += len(ready)
            if np.any(done):
                fin = np.where(done)[0]
                episode_count += len(fin)
                ep_lens.append(e_len[fin]); ep_rews.append(e_rew[fin]); ep_idxs.append(e_idx[fin])
                surplus = len(ready) - (n_episode - episode_count)      # finished envs leave the ready set (they are not reset)
                if surplus > 0:
                    mask = np.ones_like(ready, dtype=bool)
                    mask[fin[:surplus]] = False
                    ready = ready[mask]
                    self.data = self.data[mask]
            self.data.obs = self.data.obs_next
            if episode_count >= n_episode:
                break
        self.collect_step += step_count
        self.collect_episode += episode_count
        self.collect_time += max(time.time() - start, 1e-9)
        rews, lens, idxs = np.concatenate(ep_rews), np.concatenate(ep_lens), np.concatenate(ep_idxs)
        return {"n/ep": episode_count, "n/st": step_count, "rews": rews, "lens": lens, "idxs": idxs, "rew": rews.mean(), "len": lens.mean(),
                "rew_std": rews.std(), "len_std": lens.std()}
