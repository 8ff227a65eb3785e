"""GPU: SURVEY 8(f4) inference side -- cirs_select_items / cirs_rollout_static and the mirrored recommend_k_item /
interactive_evaluation against the reference's recorded choices (shipped DeepFM weights) and the C oracle."""
import os
import pickle
from types import SimpleNamespace

import numpy as np
import pandas as pd
import pytest
import torch

import staticcase

pytestmark = pytest.mark.gpu


def _model(golden_dir):
    from core.user_model_pairwise import UserModel_Pairwise
    with open(os.path.join(golden_dir, "DeepFM_params_Pair11.pickle"), "rb") as fh:
        params = pickle.load(fh)
    params["device"] = "cpu"
    m = UserModel_Pairwise(**params)
    m.load_state_dict(torch.load(os.path.join(golden_dir, "DeepFM_Pair11.pt"), map_location="cpu", weights_only=False))
    return m


def _dataset(z):
    feats, dur = staticcase.item_side(z)
    df = pd.DataFrame(feats, index=pd.Index(z["raw_pid"], name="photo_id"), columns=["feat0", "feat1", "feat2", "feat3"])
    df["photo_duration"] = z["duration"]
    return SimpleNamespace(df_photo_env=df, x_columns=list(range(7)))


def test_recommend_k_item_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "staticpolicy.npz"))
    model, ds = _model(golden_dir), _dataset(z)
    for ci in range(int(z["n_rec_cases"])):
        ucb = bool(int(z[f"r{ci}_ucb"]))
        if ucb:
            model.n_rec, model.n_each = float(z[f"r{ci}_n_rec"]), z[f"r{ci}_n_each"].copy()
        t_id, raw_id, val = model.recommend_k_item(int(z[f"r{ci}_user"]), ds, k=1, is_softmax=bool(int(z[f"r{ci}_softmax"])), epsilon=0,
                                                   is_ucb=ucb, recommended_ids=z[f"r{ci}_removed"].tolist(),
                                                   gumbel=torch.as_tensor(z[f"r{ci}_gumbel"][None, :]))
        assert [int(t_id[0]), int(raw_id[0])] == z[f"r{ci}_out"].tolist(), f"case {ci}"
        np.testing.assert_allclose(val[0], z[f"r{ci}_val"], rtol=1e-5, atol=2e-6)
        if ucb:
            assert model.n_each[int(t_id[0])] == z[f"r{ci}_n_each"][int(t_id[0])] + 1 and model.n_rec == float(z[f"r{ci}_n_rec"]) + 1


def test_recommend_k_item_more_than_one(golden_dir):
    """k > 1 (reference core/user_model.py:316-332): arg-max mode = torch.topk of the predicted values among the items left, sampling
    mode with harness noise = the Gumbel top-k of log-softmax + noise (one sample without replacement), picks distinct, removed ids avoided."""
    z = np.load(os.path.join(golden_dir, "staticpolicy.npz"))
    model, ds = _model(golden_dir), _dataset(z)
    user, k = int(z["r0_user"]), 5
    removed = [3, 17, 40]
    item_index = ds.df_photo_env.index.to_numpy()
    pred, _ = model.device_model().sweep(np.asarray([user]), item_index, ds.df_photo_env[["feat0", "feat1", "feat2", "feat3"]].to_numpy(),
                                         ds.df_photo_env["photo_duration"].to_numpy())
    u = pred[0].cpu()
    keep = np.ones(len(item_index), bool); keep[removed] = False
    t_id, raw_id, val = model.recommend_k_item(user, ds, k=k, is_softmax=False, recommended_ids=removed)
    want = np.arange(len(item_index))[keep][torch.topk(u[torch.as_tensor(keep)], k).indices.numpy()]
    assert t_id.tolist() == want.tolist() and raw_id.tolist() == item_index[want].tolist()
    np.testing.assert_allclose(val, u.numpy()[want], rtol=1e-6)
    g = torch.as_tensor(z["r0_gumbel"][None, :])
    t_id, _, _ = model.recommend_k_item(user, ds, k=k, is_softmax=True, recommended_ids=removed, gumbel=g)
    noisy = torch.log_softmax(u, 0) + g[0]          # the selection kernel's softmax draw = arg-max of logit + noise
    want = np.arange(len(item_index))[keep][torch.topk(noisy[torch.as_tensor(keep)], k).indices.numpy()]
    assert t_id.tolist() == want.tolist()
    t_id, _, _ = model.recommend_k_item(user, ds, k=k, is_softmax=True, recommended_ids=removed, seed=5)   # device noise: fresh per pick
    assert len(set(t_id.tolist())) == k and not set(t_id.tolist()) & set(removed)
    model.compile_UCB(len(item_index))
    t_id, _, _ = model.recommend_k_item(user, ds, k=k, is_softmax=False, is_ucb=True)
    assert model.n_rec == len(item_index) + k and all(model.n_each[t_id] == 2)


@pytest.mark.parametrize("n,I", [(64, 10728), (5, 33), (3, 1 << 20)])
def test_select_items_bit_exact_vs_oracle(n, I):
    from cirs_hip.static_policy import select_items
    rng = np.random.RandomState(n + I % 13)
    sc = rng.normal(size=(n, I)).astype(np.float32)
    bonus = rng.uniform(0, 0.5, I).astype(np.float32)
    vis = rng.randint(0, 1 << 32, (n, (I + 31) // 32), dtype=np.uint64).astype(np.uint32) & rng.randint(0, 1 << 32, (n, (I + 31) // 32), dtype=np.uint64).astype(np.uint32)
    skip = (rng.uniform(size=n) < 0.2).astype(np.uint8)
    d = lambda a, dt=None: torch.as_tensor(a if dt is None else a.view(dt)).cuda()
    for kw in (dict(softmax=False), dict(softmax=True, seed=11, rng_step=3), dict(softmax=True, seed=11, rng_step=4, epsilon=0.5),
               dict(softmax=False, epsilon=1.0, seed=2, rng_step=9)):
        for use_vis in (False, True):
            want_a, want_v = staticcase.oracle_select(sc, bonus=bonus, visited=vis if use_vis else None, skip=skip, **kw)
            got_a, got_v = select_items(d(sc), bonus=d(bonus), visited=d(vis, np.int32) if use_vis else None, skip=d(skip), **kw)
            np.testing.assert_array_equal(got_a.cpu().numpy(), want_a, err_msg=str((kw, use_vis)))
            np.testing.assert_array_equal(got_v.cpu().numpy(), want_v)


def _env(z, max_turn=12):
    from environments.KuaishouRec.env.kuaishouEnv import KuaishouEnv
    lbe_user = SimpleNamespace(classes_=z["raw_uid"]); lbe_photo = SimpleNamespace(classes_=z["raw_pid"])
    n_raw = int(z["raw_pid"].max()) + 1
    list_feat = [[] for _ in range(n_raw)]
    for i, rp in enumerate(z["raw_pid"]):
        list_feat[int(rp)] = [int(c) for c in z["item_cats"][i] if c >= 0]
    return KuaishouEnv(mat=z["mat"], lbe_user=lbe_user, lbe_photo=lbe_photo, list_feat=list_feat, df_photo_env=None,
                       df_dist_small=z["dist"], num_leave_compute=3, leave_threshold=1, max_turn=max_turn)


def test_interactive_evaluation_matches_reference(golden_dir):
    import evaluation as ev
    z = np.load(os.path.join(golden_dir, "staticpolicy.npz"))
    model, ds, env = _model(golden_dir), _dataset(z), _env(z)
    dom = {"feat": list(zip(z["dom_values"].tolist(), z["dom_shares"].tolist()))}
    for ei in range(int(z["n_eval_cases"])):
        cfg = [int(x) for x in z[f"e{ei}_cfg"]]
        remove, fl, ucb = cfg[0], cfg[1], bool(cfg[2]) if len(cfg) > 2 else False
        users = z[f"e{ei}_users"]
        if ucb:   # fresh arm counts, as in the fixture (reference core/user_model.py:250-252, 303-306)
            for attr in ("n_rec", "n_each"):
                if hasattr(model, attr):
                    delattr(model, attr)
        res = ev.interactive_evaluation(model, env, ds, is_softmax=False, epsilon=0, is_ucb=ucb, k=1, need_transform=True,
                                        num_trajectory=len(users), item_feat_domination=dom, remove_recommended=bool(remove), force_length=fl,
                                        top_rate=0.6, users=users)
        pre = f"NX_{fl}_" if remove else ""
        got = np.array([float(res[pre + k]) for k in ("click_loss", "CV", "CV_turn", "ctr", "len_tra", "R_tra", "ifeat_feat")])
        want = z[f"e{ei}_res"]
        np.testing.assert_array_equal(got[[1, 2, 4, 6]], want[[1, 2, 4, 6]], err_msg=f"case {ei}: counts")   # integer-derived
        np.testing.assert_allclose(got[[0, 3, 5]], want[[0, 3, 5]], rtol=1e-6, err_msg=f"case {ei}")          # float sums
    # sampling + epsilon-greedy run end to end and are reproducible
    users = z["e0_users"]
    a = ev.interactive_evaluation(model, env, ds, True, 0.2, False, 1, True, len(users), dom, remove_recommended=True, force_length=5, users=users, seed=4)
    b = ev.interactive_evaluation(model, env, ds, True, 0.2, False, 1, True, len(users), dom, remove_recommended=True, force_length=5, users=users, seed=4)
    assert a == b and a["NX_5_len_tra"] == 5.0 and float(a["NX_5_CV_turn"]) > 0.5
