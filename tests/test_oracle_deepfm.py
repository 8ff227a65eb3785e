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
