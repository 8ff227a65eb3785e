"""Shared builders for rollout / learner tests (GPU) -- random weights keyed by the reference's state_dict names."""
import numpy as np
import torch

import envcase


def tracker_param_dict(U, I, T, seed, D=32, S=20, H=128, nlayers=2, emb_scale=0.5):
    from cirs_hip.tracker import positional_encoding
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    p = {"embedding_dict.feat_user.weight": rn(U, D, scale=emb_scale), "embedding_dict.feat_item.weight": rn(I, D, scale=emb_scale),
         "ffn_user.weight": rn(D, D, scale=0.25), "ffn_user.bias": rn(D, scale=0.1),
         "fnn_gate.weight": rn(D, D + 1, scale=0.25), "fnn_gate.bias": rn(D, scale=0.1),
         "decoder.weight": rn(S, D, scale=0.25), "decoder.bias": rn(S, scale=0.1),
         "pos_encoder.pe": positional_encoding(T + 1, D).unsqueeze(1)}
    for l in range(nlayers):
        pre = f"transformer_encoder.layers.{l}."
        p[pre + "self_attn.in_proj_weight"] = rn(3 * D, D, scale=0.25); p[pre + "self_attn.in_proj_bias"] = rn(3 * D, scale=0.1)
        p[pre + "self_attn.out_proj.weight"] = rn(D, D, scale=0.25); p[pre + "self_attn.out_proj.bias"] = rn(D, scale=0.1)
        p[pre + "linear1.weight"] = rn(H, D, scale=0.25); p[pre + "linear1.bias"] = rn(H, scale=0.1)
        p[pre + "linear2.weight"] = rn(D, H, scale=0.12); p[pre + "linear2.bias"] = rn(D, scale=0.1)
        for nm in ("norm1", "norm2"):
            p[pre + nm + ".weight"] = 1 + rn(D, scale=0.2); p[pre + nm + ".bias"] = rn(D, scale=0.1)
    return p


POLICY_NAMES = dict(w1="actor.preprocess.model.model.0.weight", b1="actor.preprocess.model.model.0.bias",
                    w2="actor.preprocess.model.model.2.weight", b2="actor.preprocess.model.model.2.bias",
                    wa="actor.last.model.0.weight", ba="actor.last.model.0.bias",
                    wc="critic.last.model.0.weight", bc="critic.last.model.0.bias")


def build_device_stack(tab, B, T, *, N=10, thr=4, tau=10.0, gamma_exposure=10.0, r_decay=1.0, seed=2, with_dist=False,
                       head_scale=1.0, **rollout_kw):
    """-> (DeviceRollout, tracker params (cpu), policy arrays (numpy), env params dict)"""
    import policycase
    from cirs_hip.env import DeviceEnv, DeviceEnvTables
    from cirs_hip.policy import DevicePolicy
    from cirs_hip.rollout import DeviceRollout
    from cirs_hip.tracker import DeviceTracker
    U, I = tab.n_users, tab.n_items
    a_env, b_env = envcase.ab_env_tables(tab.raw_uid, tab.raw_pid, tab.alpha_u, tab.beta_i, U, I)
    dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, dist=tab.dist if with_dist else None, alpha_env=a_env, beta_env=b_env)
    env = DeviceEnv(dt, B, num_leave_compute=N, leave_threshold=thr, max_turn=T, tau=tau, gamma_exposure=gamma_exposure,
                    version="v1", r_decay=r_decay)
    tp = tracker_param_dict(U, I, T, seed)
    trk = DeviceTracker({k: v.float().cuda().contiguous() for k, v in tp.items()}, U, I, B, T)
    rng = np.random.RandomState(seed)
    arrs = policycase.random_weights(rng, I, head_scale=head_scale)
    pol = DevicePolicy({POLICY_NAMES[k]: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)).cuda() for k, v in arrs.items()}, I)
    ro = DeviceRollout(env, trk, pol, **rollout_kw)
    envp = dict(num_leave_compute=N, leave_threshold=thr, max_turn=T, tau=tau, gamma_exposure=gamma_exposure, version=1,
                r_decay=r_decay, has_ab=True, a_env=a_env, b_env=b_env)
    return ro, tp, arrs, envp
