"""CPU: the C oracle's item selection (recommend_k_item) against choices recorded from the reference with the
shipped DeepFM weights (greedy, removed ids, shared-noise softmax sampling, UCB bound)."""
import os

import numpy as np

import staticcase


def test_select_matches_reference_recommend_k_item(golden_dir):
    z = np.load(os.path.join(golden_dir, "staticpolicy.npz"))
    w = staticcase.shipped_weights(golden_dir)
    feats, dur = staticcase.item_side(z)
    I = len(z["raw_pid"])
    for ci in range(int(z["n_rec_cases"])):
        scores = staticcase.oracle_scores(w, [int(z[f"r{ci}_user"])], z["raw_pid"], feats, dur)
        bonus = None
        if int(z[f"r{ci}_ucb"]):
            bonus = ((2 * np.log(float(z[f"r{ci}_n_rec"])) / z[f"r{ci}_n_each"]) ** 0.5).astype(np.float32)   # user_model.py:309
        vis = staticcase.bitmap(z[f"r{ci}_removed"], I)[None, :] if len(z[f"r{ci}_removed"]) else None
        act, val = staticcase.oracle_select(scores, softmax=bool(z[f"r{ci}_softmax"]), bonus=bonus, visited=vis, gumbel=z[f"r{ci}_gumbel"][None, :])
        assert int(act[0]) == int(z[f"r{ci}_out"][0]), f"case {ci}"
        assert int(z["raw_pid"][act[0]]) == int(z[f"r{ci}_out"][1])
        np.testing.assert_allclose(val[0], z[f"r{ci}_val"], rtol=1e-5, atol=2e-6)


def test_select_semantics():
    rng = np.random.RandomState(0)
    sc = rng.normal(size=(5, 70)).astype(np.float32)
    sc[1, 10] = sc[1, 40] = 9.0                                    # tie -> lowest id
    act, val = staticcase.oracle_select(sc, softmax=False)
    assert act.tolist() == [int(np.argmax(r)) for r in sc] and act[1] == 10
    vis = np.stack([staticcase.bitmap(np.arange(70)[np.arange(70) != 3], 70)] * 5)
    act, _ = staticcase.oracle_select(sc, softmax=True, visited=vis, seed=5)
    assert (act == 3).all()                                        # one free item left
    act, _ = staticcase.oracle_select(sc, softmax=False, visited=np.stack([staticcase.bitmap(np.arange(70), 70)] * 5))
    assert (act == -1).all()                                       # nothing left
    act, _ = staticcase.oracle_select(sc, softmax=False, skip=np.array([0, 1, 0, 1, 1]))
    assert act[1] == act[3] == act[4] == -1 and act[0] >= 0
    # epsilon = 1: always a uniform random free item, different across rng steps
    picks = np.stack([staticcase.oracle_select(sc, softmax=False, epsilon=1.0, seed=3, rng_step=s)[0] for s in range(40)])
    assert picks.min() >= 0 and picks.max() < 70 and len(np.unique(picks)) > 30
