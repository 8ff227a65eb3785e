"""GPU: DeepFM pair scorer and the factored MFMA catalogue sweep vs the shipped-weights golden vectors and the C oracle."""
import os

import numpy as np
import pytest
import torch

import deepfmcase

pytestmark = pytest.mark.gpu


def test_deepfm_matches_reference_golden(golden_dir):
    from cirs_hip.deepfm import DeviceDeepFM
    z = np.load(os.path.join(golden_dir, "deepfm.npz"))
    m = DeviceDeepFM(deepfmcase.weights_from_golden(z))
    y = m.forward(z["pu"], z["pi"], z["feats"][z["pi"]], z["dur"][z["pi"]]).cpu().numpy()
    np.testing.assert_allclose(y, z["y"], rtol=1e-5, atol=2e-6)          # <= 1e-5 rel on real-id inputs
    nu, ni = z["pred"].shape
    pred, mm = m.sweep(np.arange(nu), np.arange(ni), z["feats"], z["dur"])
    np.testing.assert_allclose(pred.cpu().numpy(), z["pred"], rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(mm.cpu().numpy(), [z["pred"].min(), z["pred"].max()], rtol=1e-5, atol=3e-6)
    normed = m.normed_reward(np.arange(nu), np.arange(ni), z["feats"], z["dur"]).cpu().numpy()
    assert normed.dtype == np.float64 and normed.min() == 0.0 and normed.max() == 1.0
    np.testing.assert_allclose(normed, z["normed"], atol=5e-6)


@pytest.mark.parametrize("nu,ni,E", [(70, 333, 16), (130, 1000, 32), (65, 97, 64), (5, 31, 8)])
def test_sweep_vs_oracle(nu, ni, E):
    from cirs_hip.deepfm import DeviceDeepFM
    rng = np.random.RandomState(nu + ni)
    w = deepfmcase.random_weights(rng, nu + 3, ni + 5, E)
    users = rng.permutation(nu + 3)[:nu]; items = rng.permutation(ni + 5)[:ni]
    feats = rng.randint(0, 32, (ni, 4)); feats[rng.uniform(size=(ni, 4)) < 0.4] = 0
    dur = rng.uniform(2, 60, ni).astype(np.float32)
    uu, ii = np.meshgrid(np.arange(nu), np.arange(ni), indexing="ij")
    want = deepfmcase.oracle_forward(w, users[uu.ravel()], items[ii.ravel()], feats[ii.ravel()], dur[ii.ravel()]).reshape(nu, ni)
    m = DeviceDeepFM(w)
    pred, mm = m.sweep(users, items, feats, dur)
    scale = np.abs(want).max()
    np.testing.assert_allclose(pred.cpu().numpy() / scale, want / scale, atol=2e-5)
    got_pairs = m.forward(users[uu.ravel()], items[ii.ravel()], feats[ii.ravel()], dur[ii.ravel()]).cpu().numpy().reshape(nu, ni)
    np.testing.assert_allclose(got_pairs / scale, want / scale, atol=1e-5)
    np.testing.assert_allclose(mm.cpu().numpy(), [want.min(), want.max()], rtol=1e-4, atol=1e-4)


def test_sweep_full_baseline_shape_properties():
    """C3 shape (7176 x 10728, E = 16): min/max of the written block == reported minmax, spot rows vs the pair scorer."""
    from cirs_hip.deepfm import DeviceDeepFM
    rng = np.random.RandomState(0)
    nu, ni, E = 7176, 10728, 16
    w = deepfmcase.random_weights(rng, nu, ni + 1, E)
    feats = rng.randint(0, 32, (ni, 4)); dur = rng.uniform(2, 60, ni).astype(np.float32)
    m = DeviceDeepFM(w)
    pred, mm = m.sweep(np.arange(nu), np.arange(ni), feats, dur)
    assert float(pred.min()) == float(mm[0]) and float(pred.max()) == float(mm[1])
    rows = rng.randint(0, nu, 8)
    for u in rows:
        ref = m.forward(np.full(ni, u), np.arange(ni), feats, dur)
        torch.testing.assert_close(pred[u], ref, rtol=1e-4, atol=2e-5)


def test_shipped_checkpoint_forward_through_plugin_surface(golden_dir):
    """The reference's files -> UserModel_Pairwise(**params).load_state_dict(...).forward(X) on the device."""
    import pickle
    from core.user_model_pairwise import UserModel_Pairwise
    with open(os.path.join(golden_dir, "DeepFM_params_Pair11.pickle"), "rb") as fh:
        params = pickle.load(fh)
    params["device"] = "cpu"
    model = UserModel_Pairwise(**params)
    model.load_state_dict(torch.load(os.path.join(golden_dir, "DeepFM_Pair11.pt"), map_location="cpu", weights_only=False))
    z = np.load(os.path.join(golden_dir, "deepfm.npz"))
    pu, pi = z["pu"], z["pi"]
    X = np.concatenate([z["raw_u"][pu][:, None], z["raw_i"][pi][:, None], z["feats"][pi], z["dur"][pi][:, None]], axis=1).astype(np.float32)
    y = model.forward(torch.as_tensor(X)).cpu().numpy()
    assert y.shape == (len(pu), 1)
    np.testing.assert_allclose(y[:, 0], z["y"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("E,U,I,n", [(16, 7176, 10729, 100003), (32, 7176, 10729, 65536), (64, 3000, 5000, 4099), (8, 40, 50, 1), (32, 9, 11, 31)])
def test_gather_fm_bit_exact_vs_oracle(E, U, I, n):
    """K1-K2 kernel (cirs_gather_fm) == C oracle bit for bit (same summation order), ragged n incl. a single pair."""
    from cirs_hip.deepfm import DeviceDeepFM
    rng = np.random.RandomState(E + n)
    w = deepfmcase.random_weights(rng, U, I, E)
    X = deepfmcase.x_rows(rng.randint(0, U, n), rng.randint(0, I, n), rng.randint(0, 32, (n, 4)), rng.uniform(2, 60, n))
    got = DeviceDeepFM(w).gather_fm(X).cpu().numpy()
    want = deepfmcase.oracle_gather_fm(w, X)
    assert np.array_equal(got, want), f"max diff {np.abs(got - want).max()}"


def test_gather_fm_golden_and_large_feat_vocab(golden_dir):
    from cirs_hip.deepfm import DeviceDeepFM
    z = np.load(os.path.join(golden_dir, "deepfm.npz"))
    w = deepfmcase.weights_from_golden(z)
    X = deepfmcase.x_rows(z["pu"], z["pi"], z["feats"][z["pi"]], z["dur"][z["pi"]])
    m = DeviceDeepFM(w)
    y = m.gather_fm(X).cpu().numpy().astype(np.float64) + deepfmcase.dnn_part(w, X)
    np.testing.assert_allclose(y, z["y"], rtol=1e-5, atol=2e-6)            # + DNN branch == the reference's forward
    np.testing.assert_allclose(m.forward(z["pu"], z["pi"], z["feats"][z["pi"]], z["dur"][z["pi"]]).cpu().numpy(), y, rtol=1e-5, atol=2e-6)
    # a feat vocabulary too large for the LDS copy takes the global-gather variant: same bits
    rng = np.random.RandomState(3)
    wb = deepfmcase.random_weights(rng, 100, 120, 32, n_feat=600)
    Xb = deepfmcase.x_rows(rng.randint(0, 100, 5000), rng.randint(0, 120, 5000), rng.randint(0, 600, (5000, 4)), rng.uniform(2, 60, 5000))
    assert np.array_equal(DeviceDeepFM(wb).gather_fm(Xb).cpu().numpy(), deepfmcase.oracle_gather_fm(wb, Xb))
