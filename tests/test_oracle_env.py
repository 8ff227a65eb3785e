"""CPU: the C oracle of the env step reproduces the reference's recorded behaviour (tests/golden/env_step.npz,
recorded from SimulatedEnv/KuaishouEnv by oracle/gen_golden.py)."""
import numpy as np
import pytest

import envcase


@pytest.fixture(scope="module")
def cases(golden_dir):
    return envcase.load_env_cases(golden_dir)


def _run(base, c, *, simulated=1, dist_mode=0):
    p = c["params"]
    U, I = base["mat"].shape
    cfg = envcase.env_cfg(U, I, simulated=simulated, dist_mode=dist_mode, **p)
    a_env, b_env = envcase.ab_env_tables(base["raw_uid"], base["raw_pid"], base["alpha_u"] if p["has_ab"] else None,
                                         base["beta_i"], U, I)
    env = envcase.HostEnv(cfg, base["mat"], base["normed_mat"], base["dist"] if dist_mode == 0 else None,
                          base["item_cats"], a_env, b_env, len(c["users"]))
    return envcase.run_teacher_forced(env, c["users"], c["acts"], p["max_turn"])


def test_simulated_env_matches_reference(cases):
    base, cs = cases
    for ci, c in enumerate(cs):
        got = _run(base, c)
        envcase.compare_env_run(got, c, rtol=1e-12, what=f"case {ci} {c['params']}")


def test_jaccard_mode_matches_table_mode(cases):
    base, cs = cases
    for ci, c in enumerate(cs):
        got = _run(base, c, dist_mode=1)
        envcase.compare_env_run(got, c, rtol=1e-12, what=f"jaccard case {ci}")


def test_bare_kuaishou_env_matches_reference(cases):
    base, cs = cases
    for ci, c in enumerate(cs):
        got = _run(base, c, simulated=0)
        assert np.array_equal(got["done"], c["done"]) and np.array_equal(got["length"], c["length"])
        m = ~np.isnan(c["real_rew"])
        np.testing.assert_array_equal(got["rew"][m], c["real_rew"][m])


def test_negative_slice_window_quirk():
    """sequence_action[t-N:t] with t < N wraps: effective start = max(0, 2t-N)."""
    for N in (1, 2, 3, 5, 10):
        for t in range(0, 25):
            seq = list(range(t))
            ref = seq[t - N:t]
            start = t - N
            if start < 0:
                start = max(0, start + t)
            assert ref == seq[start:t]
