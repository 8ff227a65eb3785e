"""CPU: the torch-fp32 restatements in oracle/nn_oracle.py reproduce the reference's recorded outputs."""
import os

import numpy as np
import torch

import nn_oracle


def test_tracker_restatement_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "tracker.npz"))
    p = nn_oracle.tracker_params(z)
    states = nn_oracle.tracker_states(p, z["users"], z["acts"], z["rews"]).numpy()
    want = z["states"]
    m = np.isfinite(want)
    assert m.any() and not m.all()
    # whole-episode causal pass == the reference's per-step recompute, to fp32 round-off
    np.testing.assert_allclose(states[m], want[m], atol=1e-5, rtol=1e-5)
    # rows of envs that dropped out must not influence the others: recompute on the live subset only
    live = np.where(z["last_turn"] == z["acts"].shape[1])[0]
    sub = nn_oracle.tracker_states(p, z["users"][live], z["acts"][live], z["rews"][live]).numpy()
    np.testing.assert_allclose(sub, want[live], atol=1e-5, rtol=1e-5)


def test_policy_restatement_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "policy.npz"))
    pp = nn_oracle.policy_params(z)
    logits, value = nn_oracle.policy_forward(pp, z["s"])
    probs, logp, ent = nn_oracle.categorical_logp_entropy(logits, z["act"])
    np.testing.assert_allclose(probs.numpy(), z["probs"], atol=1e-6)
    np.testing.assert_allclose(value.numpy(), z["value"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(logp.numpy(), z["logp"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(ent.numpy(), z["ent"], atol=1e-5, rtol=1e-5)
    g = -torch.log(torch.as_tensor(z["q"]))
    assert z["margin"].min() > 1e-4  # fixture has no near-ties, so indices must be identical
    assert np.array_equal(nn_oracle.sample_with_gumbel(logits, g).numpy(), z["act"])
    assert np.array_equal(nn_oracle.sample_with_gumbel(logits, g, z["visited"]).numpy(), z["act_masked"])
