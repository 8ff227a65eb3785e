"""CPU: the C oracle's _gae_return against the KNOWN ANSWERS of the reference's own test-suite
(tianshou/test/base/test_returns.py:21-92; values restated here as data) and the pure-Python definition."""
import ctypes as C

import numpy as np

import oracle_lib


def gae(v_s, v_s_, rew, end_flag, gamma, lam):
    lib = oracle_lib.lib()
    v_s = np.ascontiguousarray(v_s, np.float64); v_s_ = np.ascontiguousarray(v_s_, np.float64)
    rew = np.ascontiguousarray(rew, np.float64); end = np.ascontiguousarray(end_flag, np.uint8)
    out = np.zeros_like(rew)
    assert lib.oracle_gae_return(v_s.ctypes.data, v_s_.ctypes.data, rew.ctypes.data, end.ctypes.data, len(rew), gamma, lam, out.ctypes.data) == 0
    return out


def episodic_return(done, rew, gamma, lam, v=None, truncated=None):
    """BasePolicy.compute_episodic_return (tianshou/policy/base.py:271-313) on a full, ordered buffer."""
    done = np.asarray(done, bool)
    rew = np.asarray(rew, float)
    n = len(rew)
    if v is None:
        v_s_ = np.zeros(n)
    else:
        mask = ~done
        if truncated is not None:
            mask = mask | np.asarray(truncated, bool)
        v_s_ = np.asarray(v, float) * mask  # value_mask, base.py:246-269
    v_s = np.roll(v_s_, 1)                 # base.py:304
    end = done.copy()
    end[-1] = True                          # unfinished_index: the last stored transition (base.py:307-308)
    adv = gae(v_s, v_s_, rew, end, gamma, lam)
    return adv + v_s


def test_known_answers_from_reference_test_returns():
    # test_returns.py:24-34
    r = episodic_return([1, 0, 0, 1, 0, 1, 0, 1], [0, 1, 2, 3, 4, 5, 6, 7.], .1, 1)
    np.testing.assert_allclose(r, [0, 1.23, 2.3, 3, 4.5, 5, 6.7, 7])
    # :36-45
    r = episodic_return([0, 1, 0, 1, 0, 1, 0], [7, 6, 1, 2, 3, 4, 5.], .1, 1)
    np.testing.assert_allclose(r, [7.6, 6, 1.2, 2, 3.4, 4, 5])
    # :47-56
    r = episodic_return([0, 1, 0, 1, 0, 0, 1], [7, 6, 1, 2, 3, 4, 5.], .1, 1)
    np.testing.assert_allclose(r, [7.6, 6, 1.2, 2, 3.45, 4.5, 5])
    # :58-73  (value function given, gamma .99, lambda .95)
    done = [0, 0, 0, 1., 0, 0, 0, 1, 0, 0, 0, 1]
    rew = [101, 102, 103., 200, 104, 105, 106, 201, 107, 108, 109, 202]
    v = [2., 3., 4, -1, 5., 6., 7, -2, 8., 9., 10, -3]
    r = episodic_return(done, rew, 0.99, 0.95, v=v)
    np.testing.assert_allclose(r, [454.8344, 376.1143, 291.298, 200., 464.5610, 383.1085, 295.387, 201., 474.2876, 390.1027, 299.476, 202.], rtol=1e-6)
    # :75-92  (TimeLimit.truncated keeps V(s') at the episode end)
    trunc = [False, False, False, True, False, False, False, True, False, False, False, False]
    r = episodic_return(done, rew, 0.99, 0.95, v=v, truncated=trunc)
    np.testing.assert_allclose(r, [454.0109, 375.2386, 290.3669, 199.01, 462.9138, 381.3571, 293.5248, 199.02, 474.2876, 390.1027, 299.476, 202.], rtol=1e-6)


def test_matches_python_definition_random():
    rng = np.random.RandomState(0)
    n = 500
    v_s, v_s_, rew = rng.normal(size=n), rng.normal(size=n), rng.uniform(size=n)
    end = rng.uniform(size=n) < 0.1
    got = gae(v_s, v_s_, rew, end, 0.95, 0.9)
    delta = rew + v_s_ * 0.95 - v_s
    m = (1.0 - end) * (0.95 * 0.9)
    want = np.zeros(n); g = 0.0
    for i in range(n - 1, -1, -1):
        g = delta[i] + m[i] * g
        want[i] = g
    np.testing.assert_array_equal(got, want)
