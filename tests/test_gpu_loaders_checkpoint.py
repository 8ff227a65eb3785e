"""GPU: SURVEY 8(f3) -- on-disk loaders (KuaishouEnv.load_mat, get_distance_mat) against outputs recorded from the reference
on the same files, and the 4-key RL checkpoint written / restored with the reference's own save code."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import loadercase

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_load_mat_matches_reference(golden_dir, tmp_path):
    import environments.KuaishouRec.env.kuaishouEnv as ke
    z = np.load(os.path.join(golden_dir, "loaders.npz"))
    root = str(tmp_path / "data")
    loadercase.write_kuairec_files(root, z["log_user"], z["log_photo"], z["log_ratio"], z["list_feat"], z["durations"])
    for attempt in ("built on the device", "read back from the cached csv"):
        mat, lbe_user, lbe_photo, list_feat, df_photo_env, df_dist = ke.KuaishouEnv.load_mat(DATAPATH=root)
        np.testing.assert_array_equal(mat, z["mat"])
        np.testing.assert_array_equal(lbe_user.classes_, z["user_classes"]); np.testing.assert_array_equal(lbe_photo.classes_, z["photo_classes"])
        assert [list(f) for f in list_feat] == [[int(c) for c in f if c >= 0] for f in z["list_feat"]]
        np.testing.assert_array_equal(df_photo_env.index.to_numpy(), z["photo_env_index"])
        assert list(df_photo_env.columns) == ["feat0", "feat1", "feat2", "feat3", "photo_duration"]
        np.testing.assert_array_equal(df_photo_env.to_numpy(dtype=np.float64), z["photo_env_values"])
        np.testing.assert_array_equal(df_dist.index.to_numpy(), z["dist_index"])
        np.testing.assert_array_equal(df_dist.columns.to_numpy().astype(np.int64), z["dist_columns"])
        want = z["dist"] if attempt.startswith("built") else z["dist_csv"]
        np.testing.assert_array_equal(df_dist.to_numpy(dtype=np.float64), want, err_msg=attempt)   # 1/Jaccard, inf when disjoint
        assert os.path.isfile(os.path.join(root, "distance_mat_photo_small.csv"))
    # the constructor's file path (mat=None) goes through the same loader
    old = ke.DATAPATH
    ke.DATAPATH = root
    try:
        env = ke.KuaishouEnv(num_leave_compute=3, leave_threshold=1, max_turn=10)
    finally:
        ke.DATAPATH = old
    assert env.mat.shape == z["mat"].shape and len(env.list_feat_small) == z["mat"].shape[1]


def _example():
    spec = importlib.util.spec_from_file_location("cirs_rl_kuaishou_synth", os.path.join(ROOT, "examples", "cirs_rl_kuaishou_synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ARGS = ["--n-users", "120", "--n-items", "300", "--training-num", "16", "--episode-per-collect", "16", "--batch-size", "64",
        "--max_turn", "12", "--tau", "10", "--leave_threshold", "0", "--num_leave_compute", "1", "--seed", "7"]


def _iterate(ex, n_iter, users, perms, start=0, state=None):
    args = ex.get_args(ARGS)
    if state is None:
        state = ex.build(args)
    tab, train_envs, st, policy, coll = state
    for it in range(start, n_iter):
        coll._collect_count = it
        res = coll.collect(n_episode=16, users=users[it])
        n = int(res["n/st"])
        policy.update(0, coll.buffer, batch_size=64, repeat=2, perms=[p[p < n] for p in perms[it]])
    return state


def test_rl_checkpoint_roundtrip_and_resume(tmp_path):
    ex = _example()
    rng = np.random.RandomState(0)
    users = [rng.randint(0, 120, 16) for _ in range(3)]
    perms = [[rng.permutation(16 * 12) for _ in range(2)] for _ in range(3)]
    # A: three iterations straight
    _, _, stA, polA, _ = _iterate(ex, 3, users, perms)
    # B: two iterations, then the reference's save code (CIRS-RL-kuaishou.py:340-347)
    stateB = _iterate(ex, 2, users, perms)
    _, _, stB, polB, _ = stateB
    optim = polB.optim
    path = str(tmp_path / "CIRS_test.pt")
    torch.save({'policy': polB.state_dict(), 'optim_RL': optim[0].state_dict(), 'optim_state': optim[1].state_dict(),
                'state_tracker': stB.state_dict()}, path)
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"policy", "optim_RL", "optim_state", "state_tracker"}
    assert "actor.last.model.0.weight" in ck["policy"] and "embedding_dict.feat_item.weight" in ck["state_tracker"]
    # the optimiser entries carry the DEVICE Adam state: moments non-zero, trunk stepped twice per optimiser step
    ln = polB._learner
    s_rl = ck["optim_RL"]["state"]
    steps = sorted({int(float(v["step"])) for v in s_rl.values()})
    assert steps == [ln.opt_step, 2 * ln.opt_step] and ln.opt_step > 0
    assert max(float(v["exp_avg_sq"].abs().max()) for v in s_rl.values()) > 0
    n_unique = len({id(p) for g in optim[0].param_groups for p in g["params"]})
    assert len(s_rl) == n_unique
    assert all(int(float(v["step"])) == stB.adam_steps for v in ck["optim_state"]["state"].values()) and stB.adam_steps == 2
    # C: fresh objects, restore, third iteration -> identical to A
    argsC = ex.get_args(ARGS)
    stateC = ex.build(argsC)
    _, _, stC, polC, collC = stateC
    polC.load_state_dict(ck["policy"]); stC.load_state_dict(ck["state_tracker"])
    polC.optim[0].load_state_dict(ck["optim_RL"]); polC.optim[1].load_state_dict(ck["optim_state"])
    assert torch.equal(polC.flat, polB.flat) and torch.equal(stC.flat, stB.flat)
    # ret_rms is a plain attribute in tianshou, not part of any state_dict: the reference loses it on resume as well.
    # Carry it over by hand to show that everything else continues bit-exactly.
    lnC = polC._get_learner(16, 12)
    lnC.rms_state.copy_(ln.rms_state)
    assert torch.equal(lnC.adam_m, ln.adam_m) and torch.equal(lnC.adam_v, ln.adam_v) and lnC.opt_step == ln.opt_step
    _iterate(ex, 3, users, perms, start=2, state=stateC)
    assert torch.equal(polC.flat, polA.flat), float((polC.flat - polA.flat).abs().max())
    assert torch.equal(stC.flat, stA.flat), float((stC.flat - stA.flat).abs().max())
