"""CPU: the C oracle of the actor forward + Gumbel-max sampler reproduces the reference's recorded outputs."""
import os

import numpy as np

import policycase


def test_actor_oracle_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "policy.npz"))
    arrs = policycase.weights_from_golden(z)
    g = (-np.log(z["q"])).astype(np.float32)  # shared-noise protocol: reference argmax(p/q) == argmax(logit - log q)
    act, logp, value, logits = policycase.oracle_sample(arrs, z["s"], gumbel=g, want_logits=True)
    probs = np.exp(logits - logits.max(1, keepdims=True)); probs /= probs.sum(1, keepdims=True)
    np.testing.assert_allclose(probs, z["probs"], atol=1e-6)
    np.testing.assert_allclose(value, z["value"], rtol=1e-5, atol=1e-5)
    assert z["margin"].min() > 1e-4
    assert np.array_equal(act, z["act"])            # action indices bit-exact under shared noise
    np.testing.assert_allclose(logp, z["logp"], rtol=1e-5, atol=1e-5)
    # masked path (remove_recommended_ids)
    bm = policycase.visited_bitmap(z["visited"], len(z["s"]), arrs["wa"].shape[0])
    act_m, _, _, _ = policycase.oracle_sample(arrs, z["s"], gumbel=g, visited=bm)
    assert np.array_equal(act_m, z["act_masked"])


def test_counter_rng_statistics():
    """Philox + fmaf-log Gumbel noise: sampling frequencies follow softmax(logits)."""
    rng = np.random.RandomState(0)
    I = 8
    arrs = policycase.random_weights(rng, I, head_scale=3.0)
    s = np.tile(rng.normal(size=(1, 20)).astype(np.float32), (4096, 1))
    counts = np.zeros(I)
    _, _, _, logits = policycase.oracle_sample(arrs, s[:1], want_logits=True)
    p = np.exp(logits[0] - logits[0].max()); p /= p.sum()
    for step in range(8):
        act, _, _, _ = policycase.oracle_sample(arrs, s, seed=1234, rng_step=step)
        counts += np.bincount(act, minlength=I)
    freq = counts / counts.sum()
    assert np.abs(freq - p).max() < 0.01, (freq, p)
