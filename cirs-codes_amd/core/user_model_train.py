"""The user-model training run of the reference (CIRS-UserModel-kuaishou.py:152-255, `main`) as one function: KuaiRec files ->
training / validation sets (core.user_data) -> UserModel_Pairwise fitted on the device -> the three artefacts the RL script
loads (`<name>_params_<msg>.pickle`, `normed_mat-<msg>.pickle`, `<name>_<msg>.pt`).  Logging, argument parsing and the upload
hook of the script are not part of it."""
import os
import pickle
from types import SimpleNamespace

import torch

from core.user_data import load_dataset_kuaishou, load_static_validate_data_kuaishou
from core.user_model_pairwise import UserModel_Pairwise, make_loss_kuaishou_pairwise
from environments.KuaishouRec.env.kuaishouEnv import KuaishouEnv

DEFAULTS = dict(env="KuaishouEnv-v0", user_model_name="DeepFM", message="UM", tau=1000.0, feature_dim=8, dnn=(64, 64),
                l2_reg_dnn=0.1, lambda_ab=10.0, is_ab=True, batch_size=2048, epoch=5, lr=1e-3, seed=2022)


def train_user_model(datapath, save_root=".", callbacks=None, rl_test=None, **overrides):
    """Returns SimpleNamespace(model, history, normed_mat, paths).  rl_test(model, epoch) (optional) is called after every epoch,
    the place of the reference's compile_RL_test hook (e.g. a partial of evaluation.test_static_model_in_RL_env)."""
    a = SimpleNamespace(**{**DEFAULTS, **overrides})
    entity_dim = a.feature_dim
    model_dir = os.path.join(save_root, "saved_models", a.env, a.user_model_name)
    os.makedirs(os.path.join(model_dir, "logs"), exist_ok=True)

    mat, lbe_user, lbe_photo, list_feat, df_photo_env, df_dist_small = KuaishouEnv.load_mat(datapath)
    train_set, x_columns, y_columns, ab_columns = load_dataset_kuaishou(a.tau, entity_dim, a.feature_dim, model_dir, datapath=datapath)
    if not a.is_ab:
        ab_columns = None
    val_set = load_static_validate_data_kuaishou(entity_dim, a.feature_dim, datapath)

    params = {"feature_columns": x_columns, "y_columns": y_columns, "task": "regression", "task_logit_dim": 1,
              "dnn_hidden_units": tuple(a.dnn), "seed": a.seed, "device": "cuda", "ab_columns": ab_columns}
    model = UserModel_Pairwise(l2_reg_dnn=a.l2_reg_dnn, **params)
    model.compile(torch.optim.Adam(model.parameters(), lr=a.lr), loss_func=make_loss_kuaishou_pairwise(a.lambda_ab))

    class _RLTest:   # epoch-end hook in the position of compile_RL_test
        def on_train_begin(self): pass
        def on_train_end(self): pass
        def on_epoch_begin(self, epoch): pass
        def on_epoch_end(self, epoch, logs):
            if rl_test is not None:
                logs["RL_val"] = rl_test(model, epoch)

    history = model.fit_data(train_set, val_set, batch_size=a.batch_size, epochs=a.epoch, callbacks=list(callbacks or []) + [_RLTest()])

    paths = SimpleNamespace(params=os.path.join(model_dir, "{}_params_{}.pickle".format(a.user_model_name, a.message)),
                            normed_mat=os.path.join(model_dir, "normed_mat-{}.pickle".format(a.message)),
                            state_dict=os.path.join(model_dir, "{}_{}.pt".format(a.user_model_name, a.message)))
    with open(paths.params, "wb") as fh:
        pickle.dump(dict(params, device="cpu"), fh)
    normed_mat = KuaishouEnv.compute_normed_reward(model, lbe_user, lbe_photo, val_set.df_photo_env)
    with open(paths.normed_mat, "wb") as fh:
        pickle.dump(normed_mat, fh)
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, paths.state_dict)
    return SimpleNamespace(model=model, history=history, normed_mat=normed_mat, paths=paths, val_set=val_set,
                           lbe_user=lbe_user, lbe_photo=lbe_photo)
