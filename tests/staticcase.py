"""Helpers for the static-policy tests: the shipped DeepFM as arrays, catalogue scores through the C oracle, oracle select."""
import ctypes as C
import os

import numpy as np

import deepfmcase


def shipped_weights(golden_dir):
    import torch
    sd = torch.load(os.path.join(golden_dir, "DeepFM_Pair11.pt"), map_location="cpu", weights_only=False)
    g = lambda k: sd[k].numpy()
    return dict(emb_user=g("embedding_dict.user_id.weight"), emb_item=g("embedding_dict.photo_id.weight"), emb_feat=g("embedding_dict.feat.weight"),
                lin_user=g("linear.embedding_dict.user_id.weight")[:, 0], lin_item=g("linear.embedding_dict.photo_id.weight")[:, 0],
                lin_feat=g("linear.embedding_dict.feat.weight")[:, 0], lin_dense=g("linear.weight").reshape(-1),
                w1=g("dnn.linears.0.weight"), b1=g("dnn.linears.0.bias"), w2=g("dnn.linears.1.weight"), b2=g("dnn.linears.1.bias"),
                last=g("last.weight").reshape(-1), out_bias=g("out.bias").reshape(-1))


def item_side(z):
    feats = np.where(z["item_cats"] < 0, 0, z["item_cats"] + 1).astype(np.int32)
    return feats, z["duration"].astype(np.float32)


def oracle_scores(w, raw_users, raw_items, feats, dur):
    out = np.zeros((len(raw_users), len(raw_items)), np.float32)
    for r, u in enumerate(raw_users):
        out[r] = deepfmcase.oracle_forward(w, np.full(len(raw_items), u), raw_items, feats, dur)
    return out


def oracle_select(scores, *, softmax, bonus=None, visited=None, skip=None, epsilon=0.0, gumbel=None, seed=0, rng_step=0):
    import oracle_lib
    lib = oracle_lib.lib()
    scores = np.ascontiguousarray(scores, np.float32)
    n, I = scores.shape
    keep = [None if a is None else np.ascontiguousarray(a, dt) for a, dt in ((bonus, np.float32), (visited, np.uint32), (skip, np.uint8), (gumbel, np.float32))]
    p = [None if a is None else a.ctypes.data for a in keep]
    act = np.zeros(n, np.int64); val = np.zeros(n, np.float32)
    rc = lib.oracle_select_items(scores.ctypes.data, I, n, I, int(softmax), p[0], p[1], p[2], float(epsilon), p[3], int(seed), int(rng_step),
                                 act.ctypes.data, val.ctypes.data)
    assert rc == 0
    return act, val


def bitmap(ids, I):
    words = np.zeros((I + 31) // 32, np.uint32)
    ids = np.asarray(ids, np.int64)
    if len(ids):
        np.bitwise_or.at(words, ids >> 5, (np.uint32(1) << (ids & 31).astype(np.uint32)))
    return words
