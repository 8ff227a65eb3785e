"""CPU: the torch restatement of UserModel_Pairwise training against three optimiser steps recorded from the
reference's own fit_data (losses, regulariser, every parameter after the first and the last step)."""
import numpy as np

import nn_oracle
import traincase


def test_training_restatement_matches_reference_fit_data(golden_dir):
    for ci, c in enumerate(traincase.load(golden_dir)):
        losses, first, final = nn_oracle.deepfm_train(c["init"], c["x"], c["y"], c["score"], c["n"], c["steps"], c["use_ab"], c["lambda_ab"])
        np.testing.assert_allclose(losses, c["losses"], rtol=2e-5, err_msg=f"case {ci}")
        traincase.compare_params(first, c["first"], c["init"], f"case {ci} first step")
        traincase.compare_params(final, c["final"], c["init"], f"case {ci} final")
        assert np.all(final["embedding_dict.feat.weight"][0] == 0)      # padding row stays zero
