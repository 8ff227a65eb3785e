"""Helpers for the user-model training tests: golden cases (recorded from the reference's fit_data)."""
import os

import numpy as np


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "usertrain.npz"))
    cases = []
    for ci in range(int(z["n_cases"])):
        pre = f"c{ci}_"
        U, I, F, E, n, use_ab, steps = (int(v) for v in z[pre + "cfg"])
        c = dict(U=U, I=I, F=F, E=E, n=n, use_ab=bool(use_ab), steps=steps, lambda_ab=float(z[pre + "lambda_ab"]), x=z[pre + "x"], y=z[pre + "y"],
                 score=z[pre + "score"], losses=z[pre + "losses"])
        for tag in ("init", "first", "final"):
            c[tag] = {k[len(pre + tag + "_"):]: z[k] for k in z.files if k.startswith(pre + tag + "_")}
        cases.append(c)
    return cases


def compare_params(got, want, prev, what, lr=1e-3):
    """Adam normalises the gradient: where a gradient component is ~0 in fp32 its SIGN decides a +-lr step, so compare tightly
    where the reference's step is well away from that regime and bound the rest by the step size."""
    for k, w in want.items():
        g = np.asarray(got[k], np.float64).reshape(w.shape)
        np.testing.assert_allclose(g, w, rtol=0, atol=2.5 * lr * 3, err_msg=f"{what}: {k}")
        close = np.abs(g - w) <= 2e-6 + 2e-5 * np.abs(w)
        assert close.mean() > 0.995, f"{what}: {k}: only {close.mean():.4f} of the entries match tightly"
