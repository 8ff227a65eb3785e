"""CPU: the C oracle of the DeepFM forward reproduces UserModel_Pairwise.forward with the SHIPPED trained weights
(tests/golden/deepfm.npz: sliced tables of reproduce_results_of_our_paper/results_alpha_beta/DeepFM_Pair11.pt)."""
import os

import numpy as np

import deepfmcase


def test_deepfm_oracle_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "deepfm.npz"))
    w = deepfmcase.weights_from_golden(z)
    y = deepfmcase.oracle_forward(w, z["pu"], z["pi"], z["feats"][z["pi"]], z["dur"][z["pi"]])
    np.testing.assert_allclose(y, z["y"], rtol=1e-5, atol=1e-6)
    # full block + reference normalisation (kuaishouEnv.py:139-143)
    nu, ni = z["pred"].shape
    uu, ii = np.meshgrid(np.arange(nu), np.arange(ni), indexing="ij")
    pred = deepfmcase.oracle_forward(w, uu.ravel(), ii.ravel(), z["feats"][ii.ravel()], z["dur"][ii.ravel()]).reshape(nu, ni)
    np.testing.assert_allclose(pred, z["pred"], rtol=1e-5, atol=1e-6)
    p64 = pred.astype(np.float64)
    normed = (p64 - p64.min()) / (p64.max() - p64.min())
    np.testing.assert_allclose(normed, z["normed"], atol=2e-6)
    assert np.abs(z["emb_feat"][0]).max() == 0.0  # padding row semantics: a zero vector in the trained table


def test_shipped_checkpoint_unpickles_into_the_mirror(golden_dir):
    """`UserModel_Pairwise(**pickle_params)` + `load_state_dict(torch.load(pt))` exactly as
    CIRS-RL-kuaishou.py:141-161 does, with the reference's own files (host side only: no device call)."""
    import os
    import pickle
    import torch
    from core.user_model_pairwise import UserModel_Pairwise
    with open(os.path.join(golden_dir, "DeepFM_params_Pair11.pickle"), "rb") as fh:
        params = pickle.load(fh)
    params["device"] = "cpu"
    model = UserModel_Pairwise(**params)
    sd = torch.load(os.path.join(golden_dir, "DeepFM_Pair11.pt"), map_location="cpu", weights_only=False)
    model.load_state_dict(sd)
    mine = model.state_dict()
    assert set(mine) == set(sd), (set(mine) ^ set(sd))
    for k in sd:
        assert mine[k].shape == sd[k].shape and torch.equal(mine[k].cpu(), sd[k].cpu()), k
    alpha_u = model.ab_embedding_dict["alpha_u"].weight.detach().cpu().numpy()
    beta_i = model.ab_embedding_dict["beta_i"].weight.detach().cpu().numpy()
    assert alpha_u.shape == (7176, 1) and beta_i.shape == (10729, 1)
    z = np.load(os.path.join(golden_dir, "deepfm.npz"))
    np.testing.assert_array_equal(alpha_u[z["raw_u"], 0], z["alpha_u"])
    np.testing.assert_array_equal(beta_i[z["raw_i"], 0], z["beta_i"])


def test_gather_fm_oracle_plus_dnn_is_the_reference_forward(golden_dir):
    """K1-K2 (embedding gather + Linear + FM, oracle_gather_fm) + the DNN branch == the reference's recorded
    UserModel_Pairwise.forward on the shipped weights: pins the split the HBM micro-benchmark measures."""
    z = np.load(os.path.join(golden_dir, "deepfm.npz"))
    w = deepfmcase.weights_from_golden(z)
    X = deepfmcase.x_rows(z["pu"], z["pi"], z["feats"][z["pi"]], z["dur"][z["pi"]])
    y = deepfmcase.oracle_gather_fm(w, X).astype(np.float64) + deepfmcase.dnn_part(w, X)
    np.testing.assert_allclose(y, z["y"], rtol=1e-5, atol=1e-6)
    # and against the literal restatement, at other embedding widths
    rng = np.random.RandomState(0)
    for E in (8, 16, 32, 64):
        wr = deepfmcase.random_weights(rng, 50, 70, E)
        n = 300
        uid, pid = rng.randint(0, 50, n), rng.randint(0, 70, n)
        feats = rng.randint(0, 32, (n, 4)); dur = rng.uniform(2, 60, n).astype(np.float32)
        Xr = deepfmcase.x_rows(uid, pid, feats, dur)
        full = deepfmcase.oracle_forward(wr, uid, pid, feats, dur)
        np.testing.assert_allclose(deepfmcase.oracle_gather_fm(wr, Xr) + deepfmcase.dnn_part(wr, Xr), full, rtol=2e-5, atol=2e-5)
