"""Helpers for the evaluation-metric tests: golden cases + the C oracle's coverage counts."""
import os

import numpy as np


def oracle_counts(act, n_items, flags=None):
    import oracle_lib
    lib = oracle_lib.lib()
    act = np.ascontiguousarray(act, np.int64).reshape(-1)
    fl = None if flags is None else np.ascontiguousarray(flags, np.uint8)
    seen = np.zeros(n_items, np.uint8)
    out = np.zeros(3, np.int64)
    rc = lib.oracle_eval_coverage(act.ctypes.data, act.size, n_items, None if fl is None else fl.ctypes.data, seen.ctypes.data, out.ctypes.data)
    assert rc == 0
    return tuple(int(x) for x in out)


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "evalmetrics.npz"))
    dom = list(zip(z["dom_values"].tolist(), z["dom_shares"].tolist()))
    cases = []
    for ci in range(int(z["n_cases"])):
        c = {"top_rate": float(z[f"c{ci}_top_rate"])}
        for name in ("FB", "NX_0", "NX_4"):
            c[name] = dict(acts=z[f"c{ci}_{name}_acts"], lens=z[f"c{ci}_{name}_lens"], out=z[f"c{ci}_{name}_out"])
        cases.append(c)
    return z, dom, cases
