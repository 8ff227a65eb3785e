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
