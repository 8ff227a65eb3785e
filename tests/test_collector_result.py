"""CPU: Collector.collect's result dict (reference core/collector.py:343-362) — the mirror's dict builder on the transitions the
reference recorded (tests/golden/learn.npz, both rounds) must reproduce the reference's dict key for key, bit for bit:
n/ep, n/st, rews, lens, idxs (episode-completion order), rew, len, rew_std, len_std."""
import os

import numpy as np
import torch

from core.collector import result_from_trajectory
from tianshou.data import VectorReplayBuffer


class _HostTraj:
    def __init__(self, rews_bt):
        self.rew = torch.as_tensor(np.ascontiguousarray(rews_bt.T))


def test_result_dict_equals_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "learn.npz"))
    U, I, B, T = [int(v) for v in z["dims"]]
    for pre in ("", "r2_"):
        buf = VectorReplayBuffer(B * T, B)      # what gen_golden's Collector was given
        res = result_from_trajectory(_HostTraj(z[pre + "rews"]), z[pre + "lens"], buf._offset)
        n_ep, n_st, rew, ln, rew_std, len_std = z[pre + "res_scalars"]
        assert res["n/ep"] == int(n_ep) and res["n/st"] == int(n_st)
        assert np.array_equal(res["lens"], z[pre + "res_lens"])
        assert np.array_equal(res["idxs"], z[pre + "res_idxs"])
        assert np.array_equal(res["rews"], z[pre + "res_rews"]), "episode rewards must be the reference's running float64 sums"
        assert res["rew"] == rew and res["len"] == ln and res["rew_std"] == rew_std and res["len_std"] == len_std
        assert set(res) == {"n/ep", "n/st", "rews", "lens", "idxs", "rew", "len", "rew_std", "len_std"}


def test_moving_average_of_the_trainer_lets_nan_through_and_bans_inf():
    """core/trainer/onpolicy.py:_Trail == tianshou/utils/statistics.py:MovAvg on what it admits: +-inf never enters the window, a COMPUTED NaN does
    (the reference bans by membership in [inf, nan, -inf], and nan != nan) -- a diverged update must stay visible in the logged loss."""
    import numpy as np
    from core.trainer.onpolicy import _Trail
    t = _Trail(span=4)
    assert t.absorb([1.0, float("inf"), 3.0, -float("inf")]) == 2.0
    assert np.isnan(t.absorb(np.array([np.float32("nan")])))
