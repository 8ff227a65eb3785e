"""CPU: evaluation metrics -- the C oracle's coverage counts and the host-side domination logic against
vectors recorded from the reference's Callback_Coverage_Count / get_feat_dominate_dict / get_sorted_domination_features."""
import numpy as np
import pandas as pd

import evalcase
from cirs_hip.evalmetrics import dominated_values, item_flags


def test_coverage_counts_match_reference(golden_dir):
    z, dom, cases = evalcase.load(golden_dir)
    I = int(z["n_items"])
    env_feats = z["feats_raw"][z["raw_pid"]]          # df_item_val.loc[lbe_photo.inverse_transform(env ids)]
    for c in cases:
        flags = item_flags(env_feats, dominated_values(dom, c["top_rate"]))
        for name in ("FB", "NX_0", "NX_4"):
            hit, n, fl = evalcase.oracle_counts(c[name]["acts"], I, flags)
            assert n == int(c[name]["lens"].sum())
            got = np.array([hit / I, hit / n, fl / n])
            np.testing.assert_array_equal(got, c[name]["out"])   # same integer counts -> identical float64 quotients


def test_dominated_values_rule():
    items = [(3, 0.5), (1, 0.3), (7, 0.2)]
    assert dominated_values(items, 0.05).tolist() == [3]            # never empty
    assert dominated_values(items, 0.5).tolist() == [3]             # cumulative share must EXCEED top_rate to stop
    assert dominated_values(items, 0.81).tolist() == [3, 1]
    assert dominated_values(items, 1.0).tolist() == [3, 1, 7]


def test_sorted_domination_features_match_reference(golden_dir):
    from environments.KuaishouRec.env.data_handler import get_sorted_domination_features
    z, dom, _ = evalcase.load(golden_dir)
    df_item = pd.DataFrame(z["feats_raw"], columns=["feat0", "feat1", "feat2", "feat3"])
    df_data = pd.DataFrame({"photo_id": z["log_pid"], "watch_ratio": z["log_ratio"]}).join(df_item, on=["photo_id"], how="left")
    got = get_sorted_domination_features(df_data, df_item, is_multi_hot=True, yname="watch_ratio", threshold=float(z["log_thr"]))
    assert [int(k) for k, _ in got["feat"]] == [int(k) for k, _ in dom]
    np.testing.assert_array_equal(np.array([v for _, v in got["feat"]]), np.array([v for _, v in dom]))


def test_logger_callback_policy_line():
    from util.utils import LoggerCallback_Policy
    res = {"n/ep": 4, "n/st": 40, "rew": 5.0, "CV": 0.25, "CV_turn": 0.5, "ifeat_feat": 0.75}
    for pre in ("NX_0_", "NX_10_"):
        res.update({pre + "n/st": 20, pre + "rew": 1.0, pre + "CV": 0.125, pre + "CV_turn": 1.0, pre + "ifeat_feat": 0.5})
    out = LoggerCallback_Policy("x.log", 10).on_epoch_end(3, results=res)
    assert out["ctr"] == "0.50000" and out["len_tra"] == 10.0 and out["R_tra"] == 5.0 and out["CV"] == "0.25000"
    assert out["NX_10_ctr"] == "0.20000" and out["NX_0_ifeat_feat"] == 0.5 and out["num_test"] == 4
