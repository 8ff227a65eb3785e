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


def tracker_forward_all(p, x_hist, nhead=4, nlayers=2):
    """Causal transformer over the whole episode, decoder applied at EVERY position: states[B, L, S].
    Because the mask is causal and dropout is off, states[:, j] equals the reference's
    forward(data[:j+1])[-1] (state_tracker.py:170-186,246) -- the equivalence the KV-cache design relies on."""
    B, L, D = x_hist.shape
    hd = D // nhead
    pe = p["pos_encoder.pe"][:L, 0, :]  # [L, D]
    h = x_hist * math.sqrt(D) + pe.unsqueeze(0)
    causal = torch.triu(torch.full((L, L), float("-inf")), diagonal=1)
    for l in range(nlayers):
        pre = f"transformer_encoder.layers.{l}."
        qkv = h @ p[pre + "self_attn.in_proj_weight"].T + p[pre + "self_attn.in_proj_bias"]
        q, k, v = qkv.split(D, dim=-1)
        q = q.view(B, L, nhead, hd).transpose(1, 2)
        k = k.view(B, L, nhead, hd).transpose(1, 2)
        v = v.view(B, L, nhead, hd).transpose(1, 2)
        sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + causal
        att = torch.softmax(sc, dim=-1) @ v  # [B, H, L, hd]
        att = att.transpose(1, 2).reshape(B, L, D)
        sa = att @ p[pre + "self_attn.out_proj.weight"].T + p[pre + "self_attn.out_proj.bias"]
        h = F.layer_norm(h + sa, (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-5)
        ff = torch.relu(h @ p[pre + "linear1.weight"].T + p[pre + "linear1.bias"])
        ff = ff @ p[pre + "linear2.weight"].T + p[pre + "linear2.bias"]
        h = F.layer_norm(h + ff, (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-5)
    return h @ p["decoder.weight"].T + p["decoder.bias"]


def tracker_states(p, users, acts, rews, nhead=4, nlayers=2):
    return tracker_forward_all(p, tracker_inputs(p, users, acts, rews), nhead, nlayers)


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
