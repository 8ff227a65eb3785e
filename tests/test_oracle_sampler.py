"""CPU: the two-level ("chunked") Gumbel-max draw of the counter-based sampler (csrc/rng.h, oracle/cirs_oracle.c two_level_draw) IS a
sample of Categorical(softmax(logits)) (reference core/policy/ppo.py:148-155 `dist.sample()`): chunk by argmax(log-mass + Gumbel),
item inside the chunk by argmax(logit + Gumbel).  Checked statistically against the exact probabilities (peaked and flat heads,
catalogue sizes that are not multiples of the 128-item chunk, masked ids), plus the building blocks: det_expf_neg accuracy and the
exactness of the mask (a masked item is never drawn)."""
import numpy as np

import policycase


def _freq(arrs, state, n_draws_rows, steps, visited=None, seed=5):
    I = arrs["wa"].shape[0]
    st = np.repeat(state[None, :], n_draws_rows, 0).astype(np.float32)
    counts = np.zeros(I, np.int64)
    vis = None if visited is None else np.repeat(visited[None, :], n_draws_rows, 0)
    logits = None
    for t in range(steps):
        a, lp, v, lg = policycase.oracle_sample(arrs, st, seed=seed, rng_step=t, visited=vis, want_logits=(t == 0))
        if t == 0:
            logits = lg[0].astype(np.float64)
        assert (a >= 0).all()
        counts += np.bincount(a, minlength=I)
    return counts, logits


def _check(counts, logits, mask=None):
    z = logits.copy()
    if mask is not None:
        z[mask] = -np.inf
        assert counts[mask].sum() == 0, "a masked id was drawn"
    p = np.exp(z - z.max()); p /= p.sum()
    n = counts.sum()
    sd = np.sqrt(np.maximum(n * p * (1 - p), 1e-9))
    zscore = (counts - n * p) / np.maximum(sd, 1.0)
    big = p * n > 20
    assert np.abs(zscore[big]).max() < 5.0, (np.abs(zscore[big]).max(), int(big.sum()))
    # chunk-level masses as well (the first stage of the draw)
    nch = (len(p) + 127) // 128
    pc = np.array([p[c * 128:(c + 1) * 128].sum() for c in range(nch)])
    cc = np.array([counts[c * 128:(c + 1) * 128].sum() for c in range(nch)])
    zc = (cc - n * pc) / np.sqrt(np.maximum(n * pc * (1 - pc), 1.0))
    assert np.abs(zc).max() < 5.0, np.abs(zc).max()
    chi2 = (((counts - n * p) ** 2)[big] / (n * p)[big]).sum()
    dof = big.sum() - 1
    assert chi2 < dof + 6 * np.sqrt(2 * dof) + 10, (chi2, dof)


def test_two_level_draw_matches_categorical_probabilities():
    for I, scale, seed in ((300, 6.0, 1), (1000, 2.0, 2), (129, 4.0, 3)):
        rng = np.random.RandomState(seed)
        arrs = policycase.random_weights(rng, I, head_scale=scale)
        state = rng.randn(20).astype(np.float32)
        counts, logits = _freq(arrs, state, 4096, 48, seed=seed)
        _check(counts, logits)


def test_two_level_draw_respects_masked_ids():
    I = 400
    rng = np.random.RandomState(4)
    arrs = policycase.random_weights(rng, I, head_scale=5.0)
    state = rng.randn(20).astype(np.float32)
    _, logits = _freq(arrs, state, 4, 1)
    top = np.argsort(-logits)[:40]                              # mask the most likely items, and one whole chunk
    masked = np.unique(np.r_[top, np.arange(128, 256)])
    bm = policycase.visited_bitmap([masked.tolist()], 1, I)[0]
    counts, logits = _freq(arrs, state, 4096, 32, visited=bm)
    mask = np.zeros(I, bool); mask[masked] = True
    _check(counts, logits, mask)


def test_det_exp_accuracy():
    import ctypes as C
    import oracle_lib
    lib = oracle_lib.lib()
    lib.oracle_det_expf_neg.restype = C.c_float
    lib.oracle_det_expf_neg.argtypes = [C.c_float]
    xs = -np.random.RandomState(0).uniform(0, 60, 4000)
    got = np.array([lib.oracle_det_expf_neg(float(x)) for x in xs])
    np.testing.assert_allclose(got, np.exp(np.float32(xs).astype(np.float64)), rtol=5e-6, atol=0)
    assert abs(lib.oracle_det_expf_neg(0.0) - 1.0) < 3e-7 and lib.oracle_det_expf_neg(-100.0) == 0.0
