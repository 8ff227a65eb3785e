"""Helpers for policy parity tests: run the C oracle's actor_sample on host arrays."""
import ctypes as C

import numpy as np

from cirs_hip import abi

NAMES = dict(w1="actor_preprocess.model.model.0.weight", b1="actor_preprocess.model.model.0.bias",
             w2="actor_preprocess.model.model.2.weight", b2="actor_preprocess.model.model.2.bias",
             wa="actor_last.model.0.weight", ba="actor_last.model.0.bias",
             wc="critic_last.model.0.weight", bc="critic_last.model.0.bias")


def host_weights(arrs):
    keep = {k: np.ascontiguousarray(arrs[k], dtype=np.float32) for k in NAMES}
    w = abi.PolicyWeights(**{k: keep[k].ctypes.data for k in NAMES})
    return w, keep


def weights_from_golden(z):
    return {k: z[v] for k, v in NAMES.items()}


def random_weights(rng, n_items, dim_state=20, hidden=64, head_scale=1.0):
    return dict(w1=rng.normal(0, 0.3, (hidden, dim_state)), b1=rng.normal(0, 0.1, hidden),
                w2=rng.normal(0, 0.2, (hidden, hidden)), b2=rng.normal(0, 0.1, hidden),
                wa=rng.normal(0, 0.2 * head_scale, (n_items, hidden)), ba=rng.normal(0, 0.1, n_items),
                wc=rng.normal(0, 0.2, (1, hidden)), bc=rng.normal(0, 0.1, 1))


def oracle_sample(arrs, state, *, gumbel=None, seed=0, rng_step=0, env_ids=None, visited=None, skip=None,
                  want_logits=False, want_margins=False):
    """want_margins (counter-based sampler only): the 4th return value is margins[n, 2] -- the top-2 margins of every row's two-level draw
    (chunk level, item level inside the drawn chunk) -- instead of the logits."""
    import oracle_lib
    lib = oracle_lib.lib()
    state = np.ascontiguousarray(state, dtype=np.float32)
    n, S = state.shape
    I = arrs["wa"].shape[0]
    cfg = abi.PolicyCfg(n_items=I, dim_state=S, hidden=64)
    w, keep = host_weights(arrs)
    act = np.zeros(n, np.int64); logp = np.zeros(n, np.float32); value = np.zeros(n, np.float32)
    logits = np.zeros((n, I), np.float32) if want_logits else None
    g = None if gumbel is None else np.ascontiguousarray(gumbel, dtype=np.float32)
    ids = None if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
    vis = None if visited is None else np.ascontiguousarray(visited, dtype=np.uint32)
    sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8)
    p = lambda a: None if a is None else a.ctypes.data  # noqa: E731
    if want_margins:
        assert gumbel is None and not want_logits
        margins = np.zeros((n, 2), np.float32)
        rc = lib.oracle_actor_sample_margins(C.byref(cfg), C.byref(w), state.ctypes.data, S, n, seed, rng_step, p(ids), p(vis), p(sk),
                                             act.ctypes.data, logp.ctypes.data, value.ctypes.data, margins.ctypes.data)
        assert rc == 0
        return act, logp, value, margins
    rc = lib.oracle_actor_sample(C.byref(cfg), C.byref(w), state.ctypes.data, S, n, p(g), seed, rng_step, p(ids),
                                 p(vis), p(sk), act.ctypes.data, logp.ctypes.data, value.ctypes.data, p(logits))
    assert rc == 0
    return act, logp, value, logits


def visited_bitmap(visited_ids, n_env, n_items):
    words = (n_items + 31) // 32
    bm = np.zeros((n_env, words), dtype=np.uint32)
    for e, ids in enumerate(visited_ids):
        for i in ids:
            bm[e, i >> 5] |= np.uint32(1) << np.uint32(i & 31)
    return bm


# "action indices identical wherever the top-2 margin > 1e-6 (report violations; expected 0)".  Since round 6 the device forms the chunk
# masses of the two-level sampler on the bf16 matrix pipe (fp32-accurate, ~1e-6 of the oracle's fma chains, no longer bit for bit), so a draw whose two
# best candidates are closer than that may legitimately fall the other way.
DRAW_MARGIN = 1e-6
DRAW_STATS = {"draws": 0, "differ": 0, "violations": 0}


def assert_draws_match(got_act, want_act, margins, what=""):
    """Device action ids against the C oracle's under the margin protocol: every id that differs must belong to a draw whose chunk-level or item-level
    top-2 margin is <= DRAW_MARGIN; anything else is a violation (expected: none).  Returns (draws, differing ids) and keeps running totals in DRAW_STATS."""
    got_act, want_act = np.asarray(got_act).reshape(-1), np.asarray(want_act).reshape(-1)
    margins = np.asarray(margins).reshape(-1, 2)
    diff = got_act != want_act
    # a finished / skipped row is -1 on both sides by construction: a -1 on one side only is never a margin effect
    bad = diff & ((margins.min(axis=1) > DRAW_MARGIN) | (got_act < 0) | (want_act < 0))
    DRAW_STATS["draws"] += int((want_act >= 0).sum()); DRAW_STATS["differ"] += int(diff.sum()); DRAW_STATS["violations"] += int(bad.sum())
    assert not bad.any(), (what, "ids differ outside the 1e-6 margin", np.where(bad)[0][:8], got_act[bad][:8], want_act[bad][:8], margins[bad][:8])
    return int((want_act >= 0).sum()), int(diff.sum())
