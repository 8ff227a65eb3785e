"""Helpers: run the C oracle's DeepFM forward on host arrays."""
import ctypes as C

import numpy as np

from cirs_hip import abi


def weights_from_golden(z):
    return {f: np.ascontiguousarray(z[f], dtype=np.float32) for f in abi.DEEPFM_FIELDS}


def random_weights(rng, n_user, n_item, E, n_feat=32):
    K = 6 * E + 1
    w = dict(emb_user=rng.normal(0, 0.3, (n_user, E)), emb_item=rng.normal(0, 0.3, (n_item, E)), emb_feat=rng.normal(0, 0.3, (n_feat, E)),
             lin_user=rng.normal(0, 0.1, n_user), lin_item=rng.normal(0, 0.1, n_item), lin_feat=rng.normal(0, 0.1, n_feat),
             lin_dense=rng.normal(0, 0.01, 1), w1=rng.normal(0, 0.15, (64, K)), b1=rng.normal(0, 0.1, 64),
             w2=rng.normal(0, 0.15, (64, 64)), b2=rng.normal(0, 0.1, 64), last=rng.normal(0, 0.2, 64), out_bias=rng.normal(0, 0.1, 1))
    w["emb_feat"][0] = 0.0  # padding row
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}


def oracle_forward(wts, uid, pid, feats, dur):
    import oracle_lib
    lib = oracle_lib.lib()
    keep = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in wts.items()}
    E = keep["emb_user"].shape[1]
    cfg = abi.DeepFMCfg(n_user_vocab=keep["emb_user"].shape[0], n_item_vocab=keep["emb_item"].shape[0],
                        n_feat_vocab=keep["emb_feat"].shape[0], emb_dim=E, hidden=64)
    w = abi.DeepFMWeights(**{f: keep[f].ctypes.data for f in abi.DEEPFM_FIELDS})
    uid = np.ascontiguousarray(uid, np.int64); pid = np.ascontiguousarray(pid, np.int64)
    feats = np.ascontiguousarray(feats, np.int32); dur = np.ascontiguousarray(dur, np.float32)
    out = np.zeros(len(uid), np.float32)
    rc = lib.oracle_deepfm_forward(C.byref(cfg), C.byref(w), uid.ctypes.data, pid.ctypes.data, feats.ctypes.data, dur.ctypes.data,
                                   len(uid), out.ctypes.data)
    assert rc == 0
    return out


def oracle_gather_fm(wts, X):
    """K1-K2 only (C oracle, summation order of csrc/gather_fm.hip): X [n,7] float32 -> linear logit + FM term."""
    import oracle_lib
    lib = oracle_lib.lib()
    keep = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in wts.items()}
    E = keep["emb_user"].shape[1]
    cfg = abi.DeepFMCfg(n_user_vocab=keep["emb_user"].shape[0], n_item_vocab=keep["emb_item"].shape[0],
                        n_feat_vocab=keep["emb_feat"].shape[0], emb_dim=E, hidden=64)
    w = abi.DeepFMWeights(**{f: keep[f].ctypes.data for f in abi.DEEPFM_FIELDS})
    X = np.ascontiguousarray(X, np.float32)
    out = np.zeros(len(X), np.float32)
    assert lib.oracle_gather_fm(C.byref(cfg), C.byref(w), X.ctypes.data, len(X), out.ctypes.data) == 0
    return out


def dnn_part(wts, X):
    """last . relu(W2 relu(W1 [v_user, v_item, v_f0..3, dur] + b1) + b2) + out_bias in float64 (core.py:120-134,155-161)."""
    X = np.asarray(X, np.float32)
    u, p, f = X[:, 0].astype(int), X[:, 1].astype(int), X[:, 2:6].astype(int)
    x = np.concatenate([wts["emb_user"][u], wts["emb_item"][p]] + [wts["emb_feat"][f[:, q]] for q in range(4)] + [X[:, 6:7]], axis=1).astype(np.float64)
    h1 = np.maximum(x @ wts["w1"].astype(np.float64).T + wts["b1"], 0)
    h2 = np.maximum(h1 @ wts["w2"].astype(np.float64).T + wts["b2"], 0)
    return h2 @ wts["last"].astype(np.float64).reshape(-1) + float(np.asarray(wts["out_bias"]).reshape(-1)[0])


def x_rows(uid, pid, feats, dur):
    return np.concatenate([np.asarray(uid, np.float32)[:, None], np.asarray(pid, np.float32)[:, None], np.asarray(feats, np.float32),
                           np.asarray(dur, np.float32)[:, None]], axis=1).astype(np.float32)
