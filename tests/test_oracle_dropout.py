"""CPU: the dropout masks of the tracker's production mode.  (a) the vectorised numpy Philox of nn_oracle equals the
scalar C restatement decision by decision; (b) the keep rate is 1 - p and masks differ across (env, position, layer, site, key);
(c) with masks injected the restatement is still causal-consistent: the state at position j computed from the prefix [0..j] alone
equals the state at position j of the full episode (what the K/V-cached decode step relies on with position-keyed masks);
(d) p = 0 / dropout=None leaves the restatement untouched (the reference-pinned eval-mode path)."""
import numpy as np
import torch

import nn_oracle
import oracle_lib
import rolloutcase


def test_numpy_philox_masks_equal_c_oracle():
    lib = oracle_lib.lib()
    key = nn_oracle.dropout_key(2023, 60)
    p = 0.1
    envs, pos = np.array([0, 3, 1029]), np.arange(5)
    for layer, site, n in [(0, nn_oracle.DROP_POS, 32), (1, nn_oracle.DROP_ATTN, 4 * 7), (0, nn_oracle.DROP_FF, 128), (1, nn_oracle.DROP_RES2, 32)]:
        m = nn_oracle.dropout_scale(key, p, envs, pos, layer, site, n).numpy()
        for a, e in enumerate(envs):
            for b, q in enumerate(pos):
                for el in range(n):
                    keep = lib.oracle_dropout_keep(key, int(e), int(q), layer, site, el, p)
                    assert (m[a, b, el] != 0) == bool(keep)
                    assert m[a, b, el] in (0.0, np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    big = nn_oracle.dropout_scale(key, p, np.arange(64), np.arange(31), 1, nn_oracle.DROP_FF, 128).numpy()
    assert abs((big != 0).mean() - 0.9) < 0.005
    other = nn_oracle.dropout_scale(nn_oracle.dropout_key(2023, 61), p, np.arange(64), np.arange(31), 1, nn_oracle.DROP_FF, 128).numpy()
    assert ((big != 0) != (other != 0)).mean() > 0.1


def test_masked_forward_is_prefix_consistent_and_p0_is_identity():
    U, I, B, T = 30, 50, 5, 9
    tp = rolloutcase.tracker_param_dict(U, I, T, seed=2)
    rng = np.random.RandomState(0)
    users, acts, rews = rng.randint(0, U, B), rng.randint(0, I, (B, T)), rng.uniform(0, 1, (B, T))
    d = dict(p=0.1, key=nn_oracle.dropout_key(7, 3), envs=np.arange(B) + 100)
    with torch.no_grad():
        full = nn_oracle.tracker_states(tp, users, acts, rews, dropout=d)
        plain = nn_oracle.tracker_states(tp, users, acts, rews)
        assert float((full - plain).abs().max()) > 1e-3
        for j in (0, 3, T):
            part = nn_oracle.tracker_states(tp, users, acts[:, :j], rews[:, :j], dropout=d)
            np.testing.assert_allclose(part[:, j].numpy(), full[:, j].numpy(), rtol=1e-5, atol=1e-6)
        none = nn_oracle.tracker_states(tp, users, acts, rews, dropout=None)
        assert torch.equal(none, plain)
