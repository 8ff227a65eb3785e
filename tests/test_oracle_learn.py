"""CPU: the PPO-update restatement (with the reference's duplicate-parameter quirks) reproduces the reference's
recorded losses and post-update parameters of actor, critic AND state tracker (tests/golden/learn.npz)."""
import os

import numpy as np
import torch

import nn_oracle

POL = dict(w1="actor.preprocess.model.model.0.weight", b1="actor.preprocess.model.model.0.bias",
           w2="actor.preprocess.model.model.2.weight", b2="actor.preprocess.model.model.2.bias",
           wa="actor.last.model.0.weight", ba="actor.last.model.0.bias",
           wc="critic.last.model.0.weight", bc="critic.last.model.0.bias")


def load_learn(golden_dir):
    z = np.load(os.path.join(golden_dir, "learn.npz"))
    tp = {k[len("trk_"):]: torch.as_tensor(z[k]).float().clone() for k in z.files if k.startswith("trk_")}
    pp = {k: torch.as_tensor(z["pol_" + v]).float().clone() for k, v in POL.items()}
    perms = [z[f"perm{i}"] for i in range(int(z["n_perm"]))]
    return z, tp, pp, perms


def test_ppo_update_matches_reference(golden_dir):
    z, tp, pp, perms = load_learn(golden_dir)
    gamma, lam, eps_clip, vf_coef, ent_coef, mgn, lr, bs, rep = z["hyper"]
    out = nn_oracle.ppo_update(tp, pp, z["users"], z["acts"], z["rews"], z["dones"], z["lens"], perms, gamma=gamma, lam=lam,
                               eps_clip=eps_clip, vf_coef=vf_coef, ent_coef=ent_coef, max_grad_norm=mgn, lr=lr,
                               batch_size=int(bs), repeat=int(rep))
    # processed batch
    np.testing.assert_allclose(out["obs"], nn_oracle.flatten_episodes(torch.as_tensor(z["obs"]), z["lens"]).numpy(), atol=2e-5)
    np.testing.assert_allclose(out["v_s"], z["b_v_s"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["returns"], z["b_returns"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["adv"], z["b_adv"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["logp_old"], z["b_logp_old"], rtol=1e-4, atol=1e-5)
    rms = out["ret_rms"]
    np.testing.assert_allclose([rms.mean, rms.var, rms.count], z["ret_rms"], rtol=1e-5)
    # per-minibatch losses (<= 1e-5 rel; fp32 summation order gives a little slack)
    np.testing.assert_allclose(out["loss"], z["loss"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(out["clip"], z["loss_clip"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(out["vf"], z["loss_vf"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(out["ent"], z["loss_ent"], rtol=2e-4, atol=2e-5)
    # post-update parameters: policy (duplicate-param Adam/clip quirk) and tracker (BPTT through the rollout)
    for k, name in POL.items():
        pre, post = z["pol_" + name], z["post_pol_" + name]
        assert np.abs(post - pre).max() > 0
        np.testing.assert_allclose(pp[k].numpy(), post, rtol=1e-4, atol=2e-6, err_msg=name)
    moved = 0
    for k, v in tp.items():
        if k == "pos_encoder.pe":
            continue
        pre, post = z["trk_" + k], z["post_trk_" + k]
        moved += int(np.abs(post - pre).max() > 0)
        got = v.numpy()
        if k.endswith("self_attn.in_proj_bias"):
            # d loss / d key-bias == 0 analytically (softmax is invariant to a per-query constant); the reference's
            # Adam normalises the fp32 round-off it gets instead into +-lr steps -> noise, not a parity target.
            got, post = np.delete(got, slice(32, 64)), np.delete(post, slice(32, 64))
        np.testing.assert_allclose(got, post, rtol=1e-4, atol=2e-6, err_msg=k)
    assert moved >= 20  # every tracker tensor received gradient through the stored obs
