"""GPU: the PPO head kernels compute their fp32 products on the bf16 matrix pipe (three bf16 pieces per operand, six MFMAs
per 16 k, fp32 accumulation: csrc/bf16x6.h).  This test pins the claim that the result is fp32-accurate: the raw minibatch
gradient (before clipping / Adam) is compared with a float64 evaluation of the same loss, next to the error a plain
float32 evaluation (torch on the host) makes against the same float64 result."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


EPS32 = float(np.finfo(np.float32).eps)


def ppo_loss(p, obs, act, adv, ret, v_s, logp_old, eps_clip, vf_coef, ent_coef):
    """ppo.py:183-212 with Categorical's fp32 clamp constant in every dtype (the clamp blocks gradients of clamped probs)."""
    h1 = torch.relu(obs @ p["w1"].T + p["b1"])
    h2 = torch.relu(h1 @ p["w2"].T + p["b2"])
    z = h2 @ p["wa"].T + p["ba"]
    value = (h2 @ p["wc"].T + p["bc"]).flatten()
    probs = torch.softmax(z, dim=-1)
    logp_all = torch.log(probs.clamp(min=EPS32, max=1 - EPS32))
    logp = logp_all.gather(1, act.view(-1, 1)).squeeze(1)
    ent = -(logp_all * probs).sum(-1).mean()
    a = (adv - adv.mean()) / adv.std()
    ratio = (logp - logp_old).exp()
    clip = -torch.min(ratio * a, ratio.clamp(1 - eps_clip, 1 + eps_clip) * a).mean()
    v_clip = v_s + (value - v_s).clamp(-eps_clip, eps_clip)
    vf = torch.max((ret - value).pow(2), (ret - v_clip).pow(2)).mean()
    return clip + vf_coef * vf - ent_coef * ent


@pytest.mark.parametrize("I,mb,ent_coef,sharp", [(3000, 512, 0.0, 1.0), (10728, 1024, 0.01, 1.0), (10728, 1024, 0.0, 1.0), (10728, 1024, 0.0, 6.0),
                                                  (10728, 992, 0.0, 3.0), (777, 96, 0.0, 2.0)])
def test_minibatch_gradient_is_fp32_accurate(I, mb, ent_coef, sharp):
    """ent_coef == 0 and != 0 are two instantiations of the fused backward kernel (the entropy term of dZ compiled in or not).  sharp > 1 scales the head weights and draws the actions FROM the policy, so that taken
    actions carry most of their row's probability (the trained regime: 1 - p_a small, where a formulation that subtracts the action's own
    term from a full sum would lose digits)."""
    from cirs_hip.learner import DeviceLearner, flat_policy_params, FLAT_ORDER
    rng = np.random.RandomState(5)
    S, H = 20, 64
    shapes = dict(w1=(H, S), b1=(H,), w2=(H, H), b2=(H,), wa=(I, H), ba=(I,), wc=(1, H), bc=(1,))
    scale = dict(w1=0.3, b1=0.1, w2=0.2, b2=0.1, wa=0.25 * sharp, ba=0.1, wc=0.2, bc=0.1)   # logits spread over several units
    p64 = {k: torch.as_tensor(rng.standard_normal(shapes[k]) * scale[k]).float().double() for k in FLAT_ORDER}
    obs = torch.as_tensor(rng.standard_normal((mb, S))).float().double()
    act = torch.as_tensor(rng.randint(0, I, mb))
    adv = torch.as_tensor(rng.standard_normal(mb)).float().double()
    ret = torch.as_tensor(rng.standard_normal(mb)).float().double()
    v_s = torch.as_tensor(rng.standard_normal(mb) * 0.5).float().double()
    eps_clip, vf_coef = 0.2, 0.25

    def grads_in(dtype):
        q = {k: v.to(dtype).clone().requires_grad_(True) for k, v in p64.items()}
        with torch.no_grad():
            z0 = torch.relu(torch.relu(obs.to(dtype) @ q["w1"].T + q["b1"]) @ q["w2"].T + q["b2"]) @ q["wa"].T + q["ba"]
        return q, z0

    # logp_old: the current policy's log-probabilities perturbed a little, so that some ratios leave the clip range
    with torch.no_grad():
        _, z0 = grads_in(torch.float64)
        if sharp > 1.0:      # on-policy actions: most rows take (one of) their most probable items
            act = torch.multinomial(torch.softmax(z0, -1), 1, generator=torch.Generator().manual_seed(3)).squeeze(1)
        lp0 = torch.log_softmax(z0, -1).gather(1, act.view(-1, 1)).squeeze(1)
    logp_old = (lp0 + torch.as_tensor(rng.standard_normal(mb) * 0.15)).float().double()

    def autograd(dtype):
        q, _ = grads_in(dtype)
        loss = ppo_loss(q, obs.to(dtype), act, adv.to(dtype), ret.to(dtype), v_s.to(dtype), logp_old.to(dtype), eps_clip, vf_coef, ent_coef)
        g = torch.autograd.grad(loss, [q[k] for k in FLAT_ORDER])
        return {k: t.double() for k, t in zip(FLAT_ORDER, g)}

    g64, g32 = autograd(torch.float64), autograd(torch.float32)

    flat, _ = flat_policy_params(I, init=None)
    off = 0
    for k in FLAT_ORDER:
        n = int(np.prod(shapes[k]))
        flat[off:off + n].copy_(p64[k].float().flatten())
        off += n
    ln = DeviceLearner(flat, I, mb, 1, gamma=0.99, gae_lambda=0.95, eps_clip=eps_clip, vf_coef=vf_coef, ent_coef=ent_coef,
                       max_grad_norm=0.5, lr=1e-3, norm_adv=True, value_clip=True, rew_norm=False)
    ln._alloc_batch(mb)
    ln.n_rows = mb
    ln.b_obs[:mb].copy_(obs.float()); ln.b_act[:mb].copy_(act.int()); ln.b_adv[:mb].copy_(adv.float())
    ln.b_ret[:mb].copy_(ret.float()); ln.b_vs[:mb].copy_(v_s.float()); ln.b_logp[:mb].copy_(logp_old.float())
    idx = torch.arange(mb, dtype=torch.int32, device="cuda")
    slot = torch.zeros(4, dtype=torch.float32, device="cuda")
    ln.mb_phase1(idx, idx, False, slot)          # forward + backward only: raw gradients in ln.grads
    torch.cuda.synchronize()
    gh = ln.grads.double().cpu()

    def rel(a, b):
        return float((a - b).norm() / b.norm())

    off = 0
    report = {}
    for k in FLAT_ORDER:
        n = int(np.prod(shapes[k]))
        hip = gh[off:off + n].view(shapes[k]); off += n
        e_hip, e_32 = rel(hip, g64[k]), rel(g32[k], g64[k])
        report[k] = (e_hip, e_32)
        # as accurate as fp32: within a small factor of what a float32 evaluation loses against float64
        assert e_hip <= max(4.0 * e_32, 3e-6), (k, e_hip, e_32)
    # the head layer is where the bf16 pieces are used: its error must be at the fp32 level in absolute terms too
    print("relative gradient error vs float64 (HIP, torch-fp32):", {k: (float("%.2e" % a), float("%.2e" % b)) for k, (a, b) in report.items()})
    # (sharp policies: the float32 evaluation itself loses more -- 4e-6 on ba at sharp = 6 --, so the bar follows it)
    bar = max(5e-6, 2.0 * max(report["wa"][1], report["ba"][1]))
    assert report["wa"][0] < bar and report["ba"][0] < bar, report
