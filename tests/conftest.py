import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cirs-codes_amd"), os.path.join(ROOT, "oracle"), ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- how much of each tolerance the golden comparisons use --------------------------------------------------------------------------
# close(...) is numpy's assert_allclose plus a record of the observed error: largest |got - want|, largest relative error where
# |want| > atol / rtol, and the largest fraction of the bar (atol + rtol |want|) any element used.  The session writes the records to
# gpurun_out/parity_margins.json (profiles/ keeps the copy the DESIGN table quotes).
_MARGINS = []


def close(got, want, rtol, atol, what):
    import numpy as np
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=what)
    if got.size:
        err = np.abs(got - want)
        bar = atol + rtol * np.abs(want)
        big = np.abs(want) > (atol / rtol if rtol > 0 else np.inf)
        _MARGINS.append({"what": what, "n": int(got.size), "rtol": rtol, "atol": atol, "max_abs_err": float(err.max()),
                         "max_rel_err": float((err[big] / np.abs(want[big])).max()) if big.any() else None,
                         "bar_used": float((err / bar).max())})


def pytest_sessionfinish(session, exitstatus):
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:      # the margin protocol of the sampler's draws (tests/policycase.py): draws compared with the C oracle, ids that differ inside the 1e-6 top-2 margin,
        import policycase      # violations (ids that differ outside it; any violation also fails its test)
        if policycase.DRAW_STATS["draws"]:
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "draw_margin_stats.json"), "w") as f:
                json.dump(dict(policycase.DRAW_STATS, margin=policycase.DRAW_MARGIN), f)
    except ImportError:
        pass
    if not _MARGINS:
        return
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_margins.json"), "w") as f:
        json.dump(_MARGINS, f, indent=1)
