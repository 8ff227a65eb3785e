"""Helpers for the dataset-preparation tests: the C oracle's exposure history and negative sampling."""
import numpy as np

from cirs_hip.synthetic import pack_item_cats


def cats_words(list_feat_padded):
    cats = np.asarray(list_feat_padded, np.int32)
    return np.ascontiguousarray(pack_item_cats(cats))


def user_start(user_id):
    user_id = np.asarray(user_id); n = len(user_id)
    change = np.r_[True, user_id[1:] != user_id[:-1]]
    return np.maximum.accumulate(np.where(change, np.arange(n), 0)).astype(np.int64)


def oracle_exposure(user_id, photo_id, timestamp, tau, words=None, dist=None):
    import oracle_lib
    lib = oracle_lib.lib()
    start = user_start(user_id)
    photo = np.ascontiguousarray(photo_id, np.int32); ts = np.ascontiguousarray(timestamp, np.float64)
    out = np.zeros(len(photo), np.float64)
    d = None if dist is None else np.ascontiguousarray(dist, np.float64)
    n_items = len(words) if words is not None else d.shape[0]
    rc = lib.oracle_exposure_history(start.ctypes.data, photo.ctypes.data, ts.ctypes.data, len(photo), None if d is None else d.ctypes.data,
                                     None if words is None else words.ctypes.data, n_items, float(tau), out.ctypes.data)
    assert rc == 0
    return out


def unpack_bits(packed_u8, n_items):
    return np.unpackbits(packed_u8, axis=1, bitorder="little")[:, :n_items].astype(bool)


def oracle_negative(users, items, small_bits, big_bits, n_items, absent=1225):
    import oracle_lib
    lib = oracle_lib.lib()
    u = np.ascontiguousarray(users, np.int64); p = np.ascontiguousarray(items, np.int64)
    a = np.ascontiguousarray(small_bits); b = np.ascontiguousarray(big_bits)
    out = np.zeros(len(u), np.int64)
    assert lib.oracle_find_negative(u.ctypes.data, p.ctypes.data, len(u), a.ctypes.data, b.ctypes.data, n_items, absent, out.ctypes.data) == 0
    return out
