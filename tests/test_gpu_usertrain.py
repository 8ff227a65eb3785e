"""GPU: cirs_deepfm_train_step against the reference's fit_data recordings and, at a larger batch / vocabulary,
against the torch oracle."""
import numpy as np
import pytest
import torch

import nn_oracle
import traincase

pytestmark = pytest.mark.gpu


def _run(c, init, x, y, score, n, steps):
    from cirs_hip.deepfm_train import DeepFMTrainer
    tr = DeepFMTrainer(init, use_ab=c["use_ab"], lambda_ab=c["lambda_ab"])
    losses, first = [], None
    for st in range(steps):
        lo = tr.step(torch.as_tensor(x[st * n:(st + 1) * n]), torch.as_tensor(y[st * n:(st + 1) * n]), torch.as_tensor(score[st * n:(st + 1) * n]))
        l5 = lo.cpu().numpy()
        losses.append([l5[0], l5[4]])
        if st == 0:
            first = {k: v.cpu().numpy() for k, v in tr.state_dict().items()}
    return np.array(losses), first, {k: v.cpu().numpy() for k, v in tr.state_dict().items()}, tr


def test_train_step_matches_reference_fit_data(golden_dir):
    for ci, c in enumerate(traincase.load(golden_dir)):
        losses, first, final, tr = _run(c, c["init"], c["x"], c["y"], c["score"], c["n"], c["steps"])
        np.testing.assert_allclose(losses, c["losses"], rtol=3e-5, err_msg=f"case {ci}")
        traincase.compare_params(first, c["first"], c["init"], f"case {ci} first step")
        traincase.compare_params(final, c["final"], c["init"], f"case {ci} final")
        assert np.all(final["embedding_dict.feat.weight"][0] == 0)
        assert set(final) == set(c["final"])


def test_train_step_vs_oracle_at_scale_and_determinism():
    U, I, F, E, n, steps = 3000, 5000, 32, 16, 2048, 2
    rng = np.random.RandomState(0)
    from cirs_hip.deepfm_train import layout
    init = {name: rng.normal(0, 0.2, shape).astype(np.float32) for name, shape in layout(U, I, F, E)}
    init["embedding_dict.feat.weight"][0] = 0
    for k in ("ab_embedding_dict.alpha_u.weight", "ab_embedding_dict.beta_i.weight"):
        init[k] = rng.normal(1, 0.2, init[k].shape).astype(np.float32)
    N = n * steps
    col = lambda v: np.asarray(v, np.float64)[:, None]
    feats = lambda: np.where(np.arange(4)[None, :] < rng.randint(1, 5, N)[:, None], rng.randint(1, F, (N, 4)), 0)
    u = rng.randint(0, U, N)
    x = np.concatenate([col(u), col(rng.zipf(1.3, N) % I), feats(), col(rng.uniform(2, 60, N)), col(u), col(rng.randint(0, I, N)), feats(),
                        col(rng.uniform(2, 60, N))], axis=1)
    y = rng.uniform(0, 5, (N, 1)); score = rng.gamma(1.0, 0.5, (N, 1))
    c = dict(use_ab=True, lambda_ab=3.0)
    want_l, want_first, want_final = nn_oracle.deepfm_train(init, x, y, score, n, steps, True, 3.0)
    got_l, got_first, got_final, tr = _run(c, init, x, y, score, n, steps)
    np.testing.assert_allclose(got_l, want_l, rtol=5e-5)
    traincase.compare_params(got_first, want_first, init, "scale first")
    traincase.compare_params(got_final, want_final, init, "scale final")
    again = _run(c, init, x, y, score, n, steps)[2]
    for k in got_final:
        assert np.array_equal(got_final[k], again[k]), k          # fixed-order reductions: identical bits


def test_mirror_fit_data_matches_reference(golden_dir):
    """UserModel_Pairwise(...).compile(...).fit_data(StaticDataset) through the mirrored plugin surface."""
    from core.inputs import SparseFeatP
    from core.static_dataset import StaticDataset
    from core.user_model_pairwise import UserModel_Pairwise, make_loss_kuaishou_pairwise
    from deepctr_torch.inputs import DenseFeat
    for ci, c in enumerate(traincase.load(golden_dir)):
        U, I, F, E = c["U"], c["I"], c["F"], c["E"]
        x_columns = [SparseFeatP("user_id", U, embedding_dim=E), SparseFeatP("photo_id", I, embedding_dim=E)] + \
                    [SparseFeatP(f"feat{i}", F, embedding_dim=E, embedding_name="feat", padding_idx=0) for i in range(4)] + [DenseFeat("photo_duration", 1)]
        ab_columns = [SparseFeatP("alpha_u", U, embedding_dim=1), SparseFeatP("beta_i", I, embedding_dim=1)] if c["use_ab"] else None
        model = UserModel_Pairwise(x_columns, [DenseFeat("y", 1)], "regression", 1, dnn_hidden_units=(64, 64), seed=2022, l2_reg_dnn=0.1,
                                   device="cpu", ab_columns=ab_columns)
        model.load_state_dict({k: torch.as_tensor(v) for k, v in c["init"].items()})
        model.compile(optimizer="adam", loss_func=make_loss_kuaishou_pairwise(c["lambda_ab"]), metric_fun={}, metrics=None)
        ds = StaticDataset(x_columns, [DenseFeat("y", 1)], num_workers=0)
        ds.compile_dataset(c["x"], c["y"], c["score"])
        hist = model.fit_data(ds, dataset_val=None, batch_size=c["n"], epochs=1, shuffle=False, callbacks=[])
        np.testing.assert_allclose(hist[0]["loss"], c["losses"].sum() / (c["n"] * c["steps"]), rtol=5e-5)
        got = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        traincase.compare_params({k: got[k] for k in c["final"]}, c["final"], c["init"], f"mirror case {ci}")
        # the forward pass now runs on the trained weights
        X = torch.as_tensor(c["x"][:8, :7], dtype=torch.float32)
        want = nn_oracle.deepfm_pair_forward({k: torch.as_tensor(v) for k, v in c["final"].items()}, X).numpy()
        np.testing.assert_allclose(model.forward(X).cpu().numpy()[:, 0], want, rtol=2e-4, atol=2e-4)
