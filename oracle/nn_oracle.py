"""Plain PyTorch fp32 (CPU) restatement of the floating-point networks on the hot path.  TEST INFRASTRUCTURE ONLY.

This is the "plain PyTorch fp32 reference of the same op" the HIP kernels are compared with (tolerances in the
tests).  It is written functionally from the reference's state_dict layout, *not* by calling
the reference's modules, and is itself pinned to golden vectors recorded from the reference
(tests/golden/tracker.npz, policy.npz; tests/test_oracle_nn.py).

  tracker_*  core/state_tracker.py:170-250 (StateTrackerTransformer.forward/build_state), eval mode
             torch.nn.TransformerEncoderLayer post-norm semantics (norm_first=False, relu, eps 1e-5)
  policy_*   tianshou/utils/net/common.py:87-92,184-197 (MLP/Net), utils/net/discrete.py:56-67,113-114 (Actor/Critic),
             core/policy/ppo.py:111-163 (forward/sample), torch.distributions.Categorical(probs=...)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))


def tracker_params(sd, prefix="sd_"):
    """state_dict (numpy or torch) -> dict of torch fp32 tensors keyed by the reference's parameter names."""
    out = {}
    for k, v in sd.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = _t(v).float()
    return out


def tracker_inputs(p, users, acts, rews):
    """x_hist[B, 1+T, D]: slot 0 = ffn_user(Emb_user[u]) (state_tracker.py:205-215); slot k = g*a with
    g = sigmoid(fnn_gate([r, a])), a = Emb_item[a_k] (state_tracker.py:225-242).  rew is float64 -> float32 (:97)."""
    users, acts = _t(users).long(), _t(acts).long()
    r = _t(rews).to(torch.float32)
    e_u = p["embedding_dict.feat_user.weight"][users]
    x0 = e_u @ p["ffn_user.weight"].T + p["ffn_user.bias"]
    a = p["embedding_dict.feat_item.weight"][acts]  # [B,T,D]
    gate_in = torch.cat([r.unsqueeze(-1), a], dim=-1)
    g = torch.sigmoid(gate_in @ p["fnn_gate.weight"].T + p["fnn_gate.bias"])
    return torch.cat([x0.unsqueeze(1), g * a], dim=1)


# ---- counter-based dropout masks (production mode): numpy restatement of csrc/rng.h -------------------------------
DROP_POS, DROP_ATTN, DROP_RES1, DROP_FF, DROP_RES2 = 0, 1, 2, 3, 4
_STREAM_DROPOUT = 0x44524F50


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11) on uint32 arrays of one shape -> 4 uint32 arrays."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0 & mask, p1 & mask, n2 & mask, p0 & mask
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def dropout_key(seed, tag=0):
    """cirs_hip.tracker.DeviceTracker.set_dropout_key's mix of (seed, collect tag) -> 64-bit Philox key."""
    mix = (int(seed) * 0x9E3779B97F4A7C15 + (int(tag) + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    return mix ^ (mix >> 29)


def dropout_scale(key, p, envs, pos, layer, site, n_elem):
    """Keep masks already scaled by 1/(1-p): float32 [len(envs), len(pos), n_elem] for (global env ids, positions); the element e
    of (env, position, layer, site) is kept iff word (e & 3) of Philox(counter = (e >> 2, env, pos | site << 12 | layer << 16,
    'DROP'), key) >= floor(p * 2^32)."""
    thr = np.uint32(int(float(np.float32(p)) * 4294967296.0))
    envs = np.asarray(envs, dtype=np.uint32)[:, None, None]
    pos = np.asarray(pos, dtype=np.uint32)[None, :, None]
    grp = (np.arange((n_elem + 3) // 4, dtype=np.uint32))[None, None, :]
    words = philox4x32_10(grp, envs, pos | np.uint32((site << 12) | (layer << 16)), np.uint32(_STREAM_DROPOUT), key & 0xFFFFFFFF, key >> 32)
    w = np.stack(np.broadcast_arrays(*words), axis=-1).reshape(envs.shape[0], pos.shape[1], -1)[:, :, :n_elem]
    return torch.as_tensor(np.where(w >= thr, np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32))


def tracker_forward_all(p, x_hist, nhead=4, nlayers=2, dropout=None):
    """Causal transformer over the whole episode, decoder applied at EVERY position: states[B, L, S].
    Because the mask is causal and dropout is off, states[:, j] equals the reference's
    forward(data[:j+1])[-1] (state_tracker.py:170-186,246) -- the equivalence the KV-cache design relies on.
    dropout = dict(p, key, envs [B] global env ids): the production mode -- inverted dropout at the reference's five kinds of sites
    (PositionalEncoding output; per layer: attention probabilities, dropout1, FF hidden, dropout2) with the counter-based masks
    of csrc/rng.h, every position keeping its masks for the whole episode."""
    B, L, D = x_hist.shape
    hd = D // nhead
    pe = p["pos_encoder.pe"][:L, 0, :]  # [L, D]
    h = x_hist * math.sqrt(D) + pe.unsqueeze(0)
    if dropout is not None:
        dk, dp, denv = dropout["key"], dropout["p"], dropout["envs"]
        dmask = lambda layer, site, n: dropout_scale(dk, dp, denv, np.arange(L), layer, site, n)  # noqa: E731
        h = h * dmask(0, DROP_POS, D)
    causal = torch.triu(torch.full((L, L), float("-inf")), diagonal=1)
    for l in range(nlayers):
        pre = f"transformer_encoder.layers.{l}."
        qkv = h @ p[pre + "self_attn.in_proj_weight"].T + p[pre + "self_attn.in_proj_bias"]
        q, k, v = qkv.split(D, dim=-1)
        q = q.view(B, L, nhead, hd).transpose(1, 2)
        k = k.view(B, L, nhead, hd).transpose(1, 2)
        v = v.view(B, L, nhead, hd).transpose(1, 2)
        sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + causal
        prob = torch.softmax(sc, dim=-1)  # [B, H, Lq, Lk]
        if dropout is not None:   # element of (env, query position): key * nhead + head
            prob = prob * dmask(l, DROP_ATTN, L * nhead).view(B, L, L, nhead).permute(0, 3, 1, 2)
        att = prob @ v  # [B, H, L, hd]
        att = att.transpose(1, 2).reshape(B, L, D)
        sa = att @ p[pre + "self_attn.out_proj.weight"].T + p[pre + "self_attn.out_proj.bias"]
        if dropout is not None:
            sa = sa * dmask(l, DROP_RES1, D)
        h = F.layer_norm(h + sa, (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-5)
        ff = torch.relu(h @ p[pre + "linear1.weight"].T + p[pre + "linear1.bias"])
        if dropout is not None:
            ff = ff * dmask(l, DROP_FF, ff.shape[-1])
        ff = ff @ p[pre + "linear2.weight"].T + p[pre + "linear2.bias"]
        if dropout is not None:
            ff = ff * dmask(l, DROP_RES2, D)
        h = F.layer_norm(h + ff, (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-5)
    return h @ p["decoder.weight"].T + p["decoder.bias"]


def tracker_states(p, users, acts, rews, nhead=4, nlayers=2, dropout=None):
    return tracker_forward_all(p, tracker_inputs(p, users, acts, rews), nhead, nlayers, dropout=dropout)


# ------------------------------------------------------------------------------------------------------------
def policy_params(z):
    g = lambda k: _t(z[k]).float()  # noqa: E731
    return dict(w1=g("actor_preprocess.model.model.0.weight"), b1=g("actor_preprocess.model.model.0.bias"),
                w2=g("actor_preprocess.model.model.2.weight"), b2=g("actor_preprocess.model.model.2.bias"),
                wa=g("actor_last.model.0.weight"), ba=g("actor_last.model.0.bias"),
                wc=g("critic_last.model.0.weight"), bc=g("critic_last.model.0.bias"))


def policy_forward(pp, s):
    """-> logits [B,I], value [B].  Actor and Critic share the Net trunk (CIRS-RL-kuaishou.py:245-247)."""
    s = _t(s).float()
    h1 = torch.relu(s @ pp["w1"].T + pp["b1"])
    h2 = torch.relu(h1 @ pp["w2"].T + pp["b2"])
    return h2 @ pp["wa"].T + pp["ba"], (h2 @ pp["wc"].T + pp["bc"]).flatten()


def categorical_logp_entropy(logits, act):
    """torch.distributions.Categorical(probs=softmax(logits)): probs re-normalised, log of probs clamped to
    [eps, 1-eps] (torch/distributions/utils.py probs_to_logits), entropy = -sum p * log p_clamped."""
    probs = torch.softmax(logits, dim=-1)
    probs = probs / probs.sum(-1, keepdim=True)
    eps = torch.finfo(probs.dtype).eps
    logp_all = torch.log(probs.clamp(min=eps, max=1 - eps))
    logp = logp_all.gather(1, _t(act).long().view(-1, 1)).squeeze(1)
    ent = -(logp_all * probs).sum(-1)
    return probs, logp, ent


def sample_with_gumbel(logits, gumbel, visited=None):
    """argmax_i (logit_i + g_i) over unmasked items; ties -> lowest index.  With g = -log(q), q ~ Exp(1), this is
    torch.multinomial's exponential race argmax(p/q) (ppo.py:148-155)."""
    score = logits + _t(gumbel).float()
    if visited is not None:
        score = score.scatter(1, _t(visited).long(), float("-inf"))
    return torch.argmax(score, dim=-1)


# ------------------------------------------------------------------------------------------------------------
# PPO update restatement (process_fn + learn), including the reference's quirks.  Pinned to tests/golden/learn.npz.
#   returns / GAE : tianshou/policy/modelfree/a2c.py:80-109, policy/base.py:271-313,380-396, utils/statistics.py:80-95
#   learn         : core/policy/ppo.py:166-246
# Quirks restated here:
#   * actor and critic share the trunk and the optimiser / clip_grad_norm_ receive the trunk parameters TWICE
#     (list(actor.parameters()) + list(critic.parameters())): the total norm counts the trunk gradient twice, the
#     clip coefficient multiplies trunk gradients twice, and Adam applies two sequential sub-steps to trunk
#     parameters per optimiser step (state['step'] advances by 2).
#   * gradients reach the state tracker through the stored obs; they are accumulated over the minibatches of the
#     LAST repeat only and applied by one Adam step at the end.
# ------------------------------------------------------------------------------------------------------------
def flatten_episodes(x_bt, lens, shift=0):
    """[B, T(+1), ...] -> buffer order (env-major concatenation of each env's first len transitions)."""
    return torch.cat([x_bt[b, shift:shift + int(lens[b])] for b in range(len(lens))], dim=0)


def gae_numpy(v_s, v_s_, rew, end_flag, gamma, lam):
    returns = np.zeros(rew.shape)
    delta = rew + v_s_ * gamma - v_s
    m = (1.0 - end_flag) * (gamma * lam)
    gae = 0.0
    for i in range(len(rew) - 1, -1, -1):
        gae = delta[i] + m[i] * gae
        returns[i] = gae
    return returns


class RunningMeanStd:
    def __init__(self):
        self.mean, self.var, self.count = 0.0, 1.0, 0

    def update(self, x):
        bm, bv, bc = np.mean(x), np.var(x), len(x)
        delta = bm - self.mean
        tot = self.count + bc
        new_mean = self.mean + delta * bc / tot
        m2 = self.var * self.count + bv * bc + delta ** 2 * self.count * bc / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot


def adam_substeps(p, g, state, lr, n_sub, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update applied n_sub times with the same gradient (duplicate-param quirk)."""
    for _ in range(n_sub):
        state["step"] += 1
        t = state["step"]
        state["m"].lerp_(g, 1 - b1)
        state["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        denom = (state["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(state["m"], denom, value=-lr / bc1)


def ppo_update(tp, pp, users, acts, rews, dones, lens, perms, *, gamma=0.95, lam=0.95, eps_clip=0.2, vf_coef=0.25,
               ent_coef=0.0, max_grad_norm=0.5, lr=1e-3, batch_size=16, repeat=2, ret_rms=None, nhead=4, opt_state=None,
               snapshot_steps=(), dual_clip=None, recompute_adv=False):
    """Runs one policy.update on teacher-forced episodes.  tp/pp are dicts of torch fp32 tensors and are updated
    IN PLACE.  Returns dict(losses..., returns, adv, v_s, logp_old, ret_rms, opt_state).  Pass the previous call's
    `ret_rms` and `opt_state` (Adam moments + step counters of optim_RL / optim_state) to continue a run: the reference keeps
    both optimisers and policy.ret_rms alive across policy.update calls (CIRS-RL-kuaishou.py:256-283).
    snapshot_steps: minibatch-step indices k for which out["snap"][k] records the policy parameters, Adam moments and step counters
    BEFORE step k, the row indices of the minibatch and the parameters AFTER it -- so a second implementation can be teacher-forced
    onto exactly this state (two fp32 implementations of a long run drift apart chaotically; single steps do not)."""
    lens = np.asarray(lens)
    ret_rms = ret_rms or RunningMeanStd()
    tparams = {k: v for k, v in tp.items() if k != "pos_encoder.pe"}
    for v in tparams.values():
        v.requires_grad_(True)
    states = tracker_forward_all(tp, tracker_inputs(tp, users, np.maximum(acts, 0), rews), nhead)
    obs = flatten_episodes(states, lens, 0)          # carries the graph into the tracker
    obs_next = flatten_episodes(states, lens, 1).detach()
    act = flatten_episodes(_t(acts), lens).long()
    rew = flatten_episodes(_t(rews), lens).double().numpy()
    done = flatten_episodes(_t(dones), lens).bool().numpy()
    N = len(act)
    with torch.no_grad():
        logits_old, _ = policy_forward(pp, obs.detach())
        _, logp_old, _ = categorical_logp_entropy(logits_old, act)

    def compute_returns():
        """A2CPolicy._compute_returns (a2c.py:80-109): critic over obs / obs_next with the CURRENT parameters, GAE, returns normalised by
        the CURRENT running variance, which is then updated.  Runs once in process_fn and, with recompute_advantage, again before every
        repeat but the first (ppo.py:176-177)."""
        with torch.no_grad():
            _, v_s_t = policy_forward(pp, obs.detach())
            _, v_ns_t = policy_forward(pp, obs_next)
        scale = np.sqrt(ret_rms.var + 1e-8)
        v_s = v_s_t.numpy().astype(np.float64) * scale
        v_ns = v_ns_t.numpy().astype(np.float64) * scale * (~done)
        end_flag = done.copy()  # every episode in the buffer is finished (n_episode == env_num)
        adv = gae_numpy(v_s, v_ns, rew, end_flag.astype(np.float64), gamma, lam)
        unnorm_returns = adv + v_s
        returns = torch.as_tensor(unnorm_returns / scale).float()
        ret_rms.update(unnorm_returns)
        return v_s_t, returns, torch.as_tensor(adv).float()
    v_s_t, returns, adv_t = compute_returns()
    first = dict(returns=returns.numpy().copy(), adv=adv_t.numpy().copy(), v_s=v_s_t.numpy().copy())

    names_trunk = ["w1", "b1", "w2", "b2"]
    names_head = ["wa", "ba", "wc", "bc"]
    if opt_state is None:
        opt_state = dict(pol={k: dict(step=0, m=torch.zeros_like(pp[k]), v=torch.zeros_like(pp[k])) for k in pp},
                         trk={k: dict(step=0, m=torch.zeros_like(v), v=torch.zeros_like(v)) for k, v in tparams.items()})
    st_pol, st_trk = opt_state["pol"], opt_state["trk"]
    out = dict(loss=[], clip=[], vf=[], ent=[], snap={})
    pi = 0
    step_k = 0
    trk_grads = None
    for rep in range(repeat):
        trk_grads = {k: torch.zeros_like(v) for k, v in tparams.items()}  # optim_state.zero_grad()
        if recompute_adv and rep > 0:
            v_s_t, returns, adv_t = compute_returns()
        dobs_rows = torch.zeros(N, obs.shape[1])
        perm = np.asarray(perms[pi]); pi += 1
        starts = list(range(0, N, batch_size))
        merge_last = N % batch_size > 0
        idx_list = []
        for s0 in starts:
            if merge_last and s0 + 2 * batch_size >= N:
                idx_list.append(perm[s0:]); break
            idx_list.append(perm[s0:s0 + batch_size])
        for idx in idx_list:
            idx_t = torch.as_tensor(idx).long()
            if step_k in snapshot_steps:
                out["snap"][step_k] = dict(idx=np.asarray(idx).copy(), pp={k: v.detach().clone() for k, v in pp.items()},
                                           m={k: st_pol[k]["m"].clone() for k in pp}, v={k: st_pol[k]["v"].clone() for k in pp},
                                           steps={k: st_pol[k]["step"] for k in pp})
            for k in pp:
                pp[k].requires_grad_(True)
            b_obs = obs[idx_t]
            logits, value = policy_forward(pp, b_obs)
            _, logp, ent = categorical_logp_entropy(logits, act[idx_t])
            a = adv_t[idx_t]
            a = (a - a.mean()) / a.std()
            ratio = (logp - logp_old[idx_t]).exp()
            surr1, surr2 = ratio * a, ratio.clamp(1 - eps_clip, 1 + eps_clip) * a
            if dual_clip:
                clip_loss = -torch.max(torch.min(surr1, surr2), dual_clip * a).mean()    # ppo.py:190-193 (for every sign of a)
            else:
                clip_loss = -torch.min(surr1, surr2).mean()
            vs = v_s_t[idx_t]
            v_clip = vs + (value - vs).clamp(-eps_clip, eps_clip)
            vf_loss = torch.max((returns[idx_t] - value).pow(2), (returns[idx_t] - v_clip).pow(2)).mean()
            ent_loss = ent.mean()
            loss = clip_loss + vf_coef * vf_loss - ent_coef * ent_loss
            plist = [pp[k] for k in names_trunk + names_head]
            tlist = list(tparams.values())
            grads = torch.autograd.grad(loss, plist + tlist + [obs], retain_graph=True, allow_unused=True)
            gp = dict(zip(names_trunk + names_head, grads[:len(plist)]))
            for k, gk in zip(tparams.keys(), grads[len(plist):-1]):
                if gk is not None:
                    trk_grads[k] += gk
            dobs_rows += grads[-1]
            for k in pp:
                pp[k].requires_grad_(False)
            # clip_grad_norm_ over [trunk..., wa, ba, trunk..., wc, bc]: trunk counted twice, scaled twice
            sq = sum(2.0 * float(gp[k].pow(2).sum()) for k in names_trunk) + sum(float(gp[k].pow(2).sum()) for k in names_head)
            total_norm = math.sqrt(sq)
            coef = min(max_grad_norm / (total_norm + 1e-6), 1.0)
            with torch.no_grad():
                for k in names_trunk:
                    adam_substeps(pp[k], gp[k] * coef * coef, st_pol[k], lr, 2)
                for k in names_head:
                    adam_substeps(pp[k], gp[k] * coef, st_pol[k], lr, 1)
            out["loss"].append(float(loss.detach())); out["clip"].append(float(clip_loss.detach())); out["vf"].append(float(vf_loss.detach())); out["ent"].append(float(ent_loss.detach()))
            if step_k in out["snap"]:
                out["snap"][step_k]["pp_after"] = {k: v.detach().clone() for k, v in pp.items()}
            step_k += 1
    with torch.no_grad():
        for k, v in tparams.items():
            v.requires_grad_(False)
            adam_substeps(v, trk_grads[k], st_trk[k], lr, 1)
    out.update(returns=first["returns"], adv=first["adv"], v_s=first["v_s"], logp_old=logp_old.numpy(), ret_rms=ret_rms,
               trk_grads={k: g.clone() for k, g in trk_grads.items()}, obs=obs.detach().numpy(),
               dobs_rows=dobs_rows.numpy(), logits_old=None, opt_state=opt_state)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# User-model training: plain-PyTorch restatement of one epoch of fit_data's inner loop for UserModel_Pairwise
#   forward        core/user_model_pairwise.py:98-132 (_deepfm: linear + FM + DNN), deepctr_torch layers
#   loss           CIRS-UserModel-kuaishou.py:262-278 (loss_kuaishou_pairwise), core/user_model_pairwise.py:134-151 (get_loss)
#   regulariser    core/user_model.py:401-417 with the three weight lists of :60-63 and user_model_pairwise.py:93
#   optimiser      torch.optim.Adam(lr=1e-3) over every parameter (core/user_model.py:_get_optim)
# ---------------------------------------------------------------------------------------------------------------------
def deepfm_pair_forward(sd, X):
    """X [n,7] float = [user, photo, feat0..3, duration] -> y [n]"""
    ids = X[:, :6].long()
    dur = X[:, 6:7]
    vs = [sd["embedding_dict.user_id.weight"][ids[:, 0]], sd["embedding_dict.photo_id.weight"][ids[:, 1]]] + \
         [sd["embedding_dict.feat.weight"][ids[:, 2 + q]] for q in range(4)]
    lin = sd["linear.embedding_dict.user_id.weight"][ids[:, 0], 0] + sd["linear.embedding_dict.photo_id.weight"][ids[:, 1], 0]
    for q in range(4):
        lin = lin + sd["linear.embedding_dict.feat.weight"][ids[:, 2 + q], 0]
    lin = lin + dur[:, 0] * sd["linear.weight"].reshape(())
    S = sum(vs)
    fm = 0.5 * ((S * S) - sum(v * v for v in vs)).sum(1)
    x = torch.cat(vs + [dur], dim=1)
    h1 = torch.relu(x @ sd["dnn.linears.0.weight"].T + sd["dnn.linears.0.bias"])
    h2 = torch.relu(h1 @ sd["dnn.linears.1.weight"].T + sd["dnn.linears.1.bias"])
    dnn = (h2 @ sd["last.weight"].T)[:, 0] + sd["out.bias"].reshape(())
    return lin + fm + dnn


def deepfm_train(init, x, y, score, n, steps, use_ab, lambda_ab, l2_embedding=1e-5, l2_linear=1e-5, l2_all=0.1, lr=1e-3):
    """-> (losses [steps,2] = (loss, reg_loss), state_dict after the first step, state_dict after the last step)"""
    sd = {k: torch.tensor(np.asarray(v), dtype=torch.float32, requires_grad=True) for k, v in init.items()}
    opt = torch.optim.Adam(list(sd.values()), lr=lr)
    x = torch.as_tensor(x, dtype=torch.float32); y = torch.as_tensor(y, dtype=torch.float32).reshape(-1)
    score = torch.as_tensor(score, dtype=torch.float32).reshape(-1)
    losses, first = [], None
    for st in range(steps):
        xb, yb, sb = x[st * n:(st + 1) * n], y[st * n:(st + 1) * n], score[st * n:(st + 1) * n]
        yp, yn = deepfm_pair_forward(sd, xb[:, :7]), deepfm_pair_forward(sd, xb[:, 7:])
        if use_ab:
            a = sd["ab_embedding_dict.alpha_u.weight"][xb[:, 0].long(), 0]; b = sd["ab_embedding_dict.beta_i.weight"][xb[:, 1].long(), 0]
            ex_new = sb * a * b
            loss_ab = ((a - 1) ** 2).mean() + ((b - 1) ** 2).mean()
        else:
            ex_new, loss_ab = sb, 0.0
        loss = ((yp / (1 + ex_new) - yb) ** 2).mean() - torch.log(torch.sigmoid(yp - yn)).mean() + lambda_ab * loss_ab
        reg = 0.0
        for k, p in sd.items():
            c = l2_all
            if k.startswith("embedding_dict."):
                c += l2_embedding
            if k.startswith("linear_model."):
                c += l2_linear
            reg = reg + c * (p * p).sum()
        opt.zero_grad()
        (loss + reg).backward()
        sd["embedding_dict.feat.weight"].grad[0] = 0       # nn.Embedding(padding_idx=0): the padding row never receives a gradient...
        sd["embedding_dict.feat.weight"].grad[0] += 2 * (l2_all + l2_embedding) * sd["embedding_dict.feat.weight"].detach()[0]  # ...but it is regularised
        opt.step()
        losses.append([float(loss), float(reg)])
        if st == 0:
            first = {k: v.detach().clone().numpy() for k, v in sd.items()}
    return np.array(losses), first, {k: v.detach().clone().numpy() for k, v in sd.items()}
