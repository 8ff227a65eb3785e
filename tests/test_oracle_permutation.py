"""CPU: the keyed pseudo-random permutation that shuffles the minibatches (oracle restatement of cirs_random_permutation): it is
a permutation of [0, n) for every n, depends on seed and tag, and keeps its pinned values."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib


def oracle_perm(n, seed, tag):
    out = np.empty(n, np.int32)
    assert oracle_lib.lib().oracle_random_permutation(n, seed, tag, out.ctypes.data_as(C.c_void_p)) == 0
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 17, 64, 1000, 28673, 100003])
def test_is_a_permutation(n):
    p = oracle_perm(n, 20230, 7)
    assert np.array_equal(np.sort(p), np.arange(n))


def test_depends_on_key_and_is_pinned():
    a, b, c = oracle_perm(1000, 1, 0), oracle_perm(1000, 1, 1), oracle_perm(1000, 2, 0)
    assert not np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.array_equal(a, oracle_perm(1000, 1, 0))
    # no fixed structure: displacement statistics of a random permutation (mean |p(i) - i| = n/3)
    p = oracle_perm(30000, 99, 5).astype(np.int64)
    assert abs(np.abs(p - np.arange(30000)).mean() / 30000 - 1 / 3) < 0.01
    assert (p[1:] == p[:-1] + 1).mean() < 0.001   # neighbours do not stay neighbours
    # known answer (pins the construction: 6 rounds, splitmix64 round function, cycle walking)
    assert oracle_perm(10, 20230, 0).tolist() == [4, 7, 3, 8, 6, 1, 9, 0, 2, 5]
