"""GPU (single device, W virtual ranks as threads): the tensor-parallel learner for an ITEM-SHARDED actor head
(`DeviceLearner.learn_tp` / cirs_ppo_minibatch_tp; BASELINE configs[4], VERDICT r02 missing #1) against the single-device learner
holding the whole catalogue: same minibatches, same losses, the concatenation of the shards' head parameters == the full head, trunk /
critic bit-identical on every rank, d loss / d obs (the gradient into the tracker) equal.  Collectives: the thread stand-ins of
tests/test_gpu_engine_dp.py (all-gather of 16 B per row, all-reduce of the d h2 partials)."""
import threading

import numpy as np
import pytest
import torch
import torch.distributed as dist

import nn_oracle
import policycase
import rolloutcase
from test_gpu_engine_dp import FakeCollectives
from test_gpu_learn import POL, rollout_time_value_logp, upload_traj

pytestmark = pytest.mark.gpu


def _shard_params(pp, r, W, Is):
    lo, hi = r * Is, (r + 1) * Is
    out = dict(pp)
    out["wa"], out["ba"] = pp["wa"][lo:hi].contiguous(), pp["ba"][lo:hi].contiguous()
    return out


@pytest.mark.parametrize("W,I,B,T,bs,ent", [(2, 512, 40, 12, 64, 0.0), (4, 1024, 48, 12, 100, 0.01), (8, 8 * 1408, 256, 30, 1024, 0.0)])
def test_tp_learner_equals_single_device(monkeypatch, W, I, B, T, bs, ent):
    from cirs_hip.distributed import Collectives
    from cirs_hip.learner import DeviceLearner, flat_policy_params
    from cirs_hip.rollout import Trajectory
    Is = I // W
    rng = np.random.RandomState(W * 7 + B)
    tp = rolloutcase.tracker_param_dict(100, I, T, seed=1)
    arrs = policycase.random_weights(rng, I, head_scale=1.5)
    pp = {k: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)) for k, v in arrs.items()}
    lens = rng.randint(5, T + 1, size=B)
    users = rng.randint(0, 100, B); acts = rng.randint(0, I, (B, T)); rews = rng.uniform(0, 1, (B, T))
    dones = np.zeros((B, T), bool); dones[np.arange(B), lens - 1] = True
    with torch.no_grad():
        obs_bts = nn_oracle.tracker_states(tp, users, acts, rews).numpy()
    value, logp = rollout_time_value_logp(pp, obs_bts, acts, lens)
    n = int(lens.sum())
    perms = [rng.permutation(n) for _ in range(2)]
    traj = Trajectory(B, T, 20, "cuda")
    upload_traj(traj, acts, rews, dones, lens, obs_bts, value, logp)
    kw = dict(gamma=0.95, gae_lambda=0.95, eps_clip=0.2, vf_coef=0.25, ent_coef=ent, max_grad_norm=0.5, lr=1e-3, norm_adv=True, value_clip=True,
              rew_norm=True)

    def make(params, n_items):
        flat, views = flat_policy_params(n_items, init={POL[k]: v for k, v in params.items()})
        return DeviceLearner(flat, n_items, B, T, **kw), views

    ref, ref_views = make(pp, I)
    ref.prepare(traj, lens)
    ref_losses = ref.learn(bs, 2, perms=perms)

    fake = FakeCollectives(W)
    monkeypatch.setattr(dist, "all_reduce", fake.all_reduce)
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake.all_gather_into_tensor)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    ranks = [make(_shard_params(pp, r, W, Is), Is) for r in range(W)]
    results = [None] * W

    def run(r):
        try:
            fake.local.rank = r
            ln = ranks[r][0]
            ln.prepare(traj, lens)
            results[r] = ln.learn_tp(bs, 2, perms, r, W, r * Is, Collectives())
        except Exception as exc:  # noqa: BLE001
            fake.errors.append(exc)
            fake.bar.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not fake.errors, fake.errors
    monkeypatch.undo()
    trunk_names = [POL[k] for k in ("w1", "b1", "w2", "b2", "wc", "bc")]
    for r in range(1, W):          # replicated parts and the reported losses are bit-identical on every rank
        assert torch.equal(results[0], results[r])
        for name in trunk_names:
            assert torch.equal(ranks[0][1][name], ranks[r][1][name]), name
        assert torch.equal(ranks[0][0].dobs, ranks[r][0].dobs)
    np.testing.assert_allclose(results[0].cpu().numpy(), ref_losses.cpu().numpy(), rtol=3e-4, atol=3e-5)
    wa = torch.cat([ranks[r][1][POL["wa"]] for r in range(W)]).cpu().numpy()
    ba = torch.cat([ranks[r][1][POL["ba"]] for r in range(W)]).cpu().numpy()
    np.testing.assert_allclose(wa, ref_views[POL["wa"]].cpu().numpy(), rtol=3e-4, atol=3e-6)
    np.testing.assert_allclose(ba, ref_views[POL["ba"]].cpu().numpy(), rtol=3e-4, atol=3e-6)
    for name in trunk_names:
        np.testing.assert_allclose(ranks[0][1][name].cpu().numpy(), ref_views[name].cpu().numpy(), rtol=3e-4, atol=3e-6, err_msg=name)
    np.testing.assert_allclose(ranks[0][0].dobs.cpu().numpy(), ref.dobs.cpu().numpy(), rtol=2e-3, atol=1e-7)
