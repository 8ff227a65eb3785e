"""GPU: cirs_eval_coverage and the Callback_Coverage_Count mirror vs the reference's recorded outputs and the C oracle."""
from types import SimpleNamespace

import numpy as np
import pandas as pd
import pytest
import torch

import evalcase

pytestmark = pytest.mark.gpu


def test_device_counts_match_reference_golden(golden_dir):
    from cirs_hip.evalmetrics import CoverageCounter, dominated_values, item_flags
    z, dom, cases = evalcase.load(golden_dir)
    I = int(z["n_items"])
    env_feats = z["feats_raw"][z["raw_pid"]]
    cc = CoverageCounter(I)
    for c in cases:
        flags = item_flags(env_feats, dominated_values(dom, c["top_rate"]))
        for name in ("FB", "NX_0", "NX_4"):
            act = torch.as_tensor(c[name]["acts"].T.copy())       # time-major like the rollout's act tensor
            hit, n, fl = cc.count(act.cuda(), torch.as_tensor(flags))
            np.testing.assert_array_equal(np.array([hit / I, hit / n, fl / n]), c[name]["out"])


def test_callback_mirror_matches_reference_golden(golden_dir):
    from sklearn.preprocessing import LabelEncoder
    from evaluation import Callback_Coverage_Count, get_feat_dominate_dict
    z, dom, cases = evalcase.load(golden_dir)
    I = int(z["n_items"])
    df_item = pd.DataFrame(z["feats_raw"], columns=["feat0", "feat1", "feat2", "feat3"])
    lbe = LabelEncoder().fit(z["raw_pid"])
    for c in cases:
        coll = {}
        for name in ("FB", "NX_0", "NX_4"):
            traj = SimpleNamespace(act=torch.as_tensor(c[name]["acts"].T.copy()).cuda())
            coll[name] = SimpleNamespace(buffer=SimpleNamespace(_rollout=SimpleNamespace(traj=traj)))
        tcs = SimpleNamespace(collector_dict=coll, env=SimpleNamespace(mat=[np.zeros((2, I))]))
        cb = Callback_Coverage_Count(tcs, df_item, True, {"feat": dom}, lbe, c["top_rate"])
        res = cb.on_epoch_end(0, results={"n/ep": 12})
        for name in ("FB", "NX_0", "NX_4"):
            pre = "" if name == "FB" else name + "_"
            np.testing.assert_array_equal(np.array([res[pre + "CV"], res[pre + "CV_turn"], res[pre + "ifeat_feat"]]), c[name]["out"])
        acts = c["FB"]["acts"]
        d = get_feat_dominate_dict(df_item, z["raw_pid"][acts[acts >= 0]], {"feat": dom}, top_rate=c["top_rate"])
        assert d["ifeat_feat"] == c["FB"]["out"][2]


@pytest.mark.parametrize("I,T,B", [(10728, 30, 1024), (1 << 20, 30, 4096), (33, 1, 1)])
def test_device_counts_vs_oracle_at_scale(I, T, B):
    from cirs_hip.evalmetrics import CoverageCounter
    rng = np.random.RandomState(I % 97)
    act = rng.randint(0, I, (T, B)).astype(np.int64)
    if T > 1:
        lens = rng.randint(1, T + 1, B)
        act[np.arange(T)[:, None] >= lens[None, :]] = -1
    flags = (rng.uniform(size=I) < 0.3).astype(np.uint8)
    cc = CoverageCounter(I)
    got = cc.count(torch.as_tensor(act).cuda(), torch.as_tensor(flags))
    assert got == evalcase.oracle_counts(act, I, flags)
    assert cc.count(torch.as_tensor(act).cuda(), None)[:2] == got[:2]
    empty = cc.count(torch.full((T, B), -1, dtype=torch.int64).cuda(), torch.as_tensor(flags))
    assert empty == (0, 0, 0)
