"""GPU: the user-model training run end to end (core.user_model_train.train_user_model: files -> datasets -> fit on the device ->
the artefacts the RL script loads).  The optimiser step itself is pinned to the reference elsewhere (test_gpu_usertrain.py), the
data sets in test_gpu_userdata.py / test_oracle_userval.py; here: the pieces compose, the loss falls, and the saved artefacts
reproduce the model."""
import os
import pickle

import numpy as np
import pytest
import torch

from test_gpu_userdata import write_files

pytestmark = pytest.mark.gpu


def test_training_run_writes_loadable_artefacts(golden_dir, tmp_path):
    from core.user_model_pairwise import UserModel_Pairwise
    from core.user_model_train import train_user_model
    from environments.KuaishouRec.env.kuaishouEnv import KuaishouEnv
    z = np.load(os.path.join(golden_dir, "userdata.npz"))
    root = str(tmp_path / "data")
    write_files(root, z)
    calls = []
    run = train_user_model(root, save_root=str(tmp_path), tau=800.0, feature_dim=8, batch_size=64, epoch=6, lr=5e-3,
                           rl_test=lambda model, epoch: calls.append(epoch) or float(epoch))
    losses = [h["loss"] for h in run.history]
    assert len(losses) == 6 and all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert calls == list(range(6)) and run.history[-1]["RL_val"] == 5.0
    for p in (run.paths.params, run.paths.normed_mat, run.paths.state_dict):
        assert os.path.isfile(p)
    # what CIRS-RL-kuaishou.py does with the artefacts: rebuild the model from the pickle + state dict, recompute the reward table
    with open(run.paths.params, "rb") as fh:
        params = pickle.load(fh)
    clone = UserModel_Pairwise(**params)
    clone.load_state_dict(torch.load(run.paths.state_dict))
    again = KuaishouEnv.compute_normed_reward(clone, run.lbe_user, run.lbe_photo, run.val_set.df_photo_env)
    with open(run.paths.normed_mat, "rb") as fh:
        saved = pickle.load(fh)
    assert saved.shape == (len(run.lbe_user.classes_), len(run.lbe_photo.classes_))
    np.testing.assert_allclose(again, saved, rtol=1e-6, atol=1e-9)
    assert saved.min() == 0.0 and saved.max() == 1.0
