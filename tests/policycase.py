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
                  want_logits=False):
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
