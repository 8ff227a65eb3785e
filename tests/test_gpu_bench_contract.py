"""GPU: bench.py keeps its contract -- one JSON line on stdout with the driver's keys, the roofline object of the dominant kernel
and (single GPU) the CPU baseline leg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_json_line():
    d = run_bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "c2")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "env-steps/s" and d["value"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1
    # the PPO configuration is part of the line (VERDICT r02 next #1a): learner, global minibatch, steps per update, dropout mode
    cfg = d["config"]
    assert cfg["learner"] == "single" and cfg["global_minibatch"] == 1024 and cfg["minibatch_steps_per_update"] >= 2
    assert cfg["rows_per_rank_per_minibatch"] == 1024 and cfg["tracker_dropout"] == 0.0 and d["dropout"] == 0.0
    assert c["envs_gpu_leg"] == 64 and c["envs_cpu_leg"] >= 1


def test_bench_dropout_mode_line():
    """--dropout 0.1: the mode the reference trains in (tracker in train mode, masks in rollout and BPTT) is measurable and labelled."""
    d = run_bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "c2", "--dropout", "0.1", "--no-probes")
    assert d["dropout"] == 0.1 and d["config"]["tracker_dropout"] == 0.1 and d["value"] > 0



def test_bench_default_workload_carries_dropout_and_c2_passes():
    """the default (C3) single-GPU line also carries a timed pass with the tracker in training mode (Dropout(0.1),
    the mode the reference trains in) and a timed pass of BASELINE configs[1] (C2), both with the headline's warm-up / step protocol."""
    d = run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert "7176x10728" in d["config"]["workload"] and d["dropout"] == 0.0
    for key, envs, p in (("dropout_on", 1024, 0.1), ("c2", 64, 0.0)):
        e = d[key]
        assert e["value"] > 0 and e["unit"] == "env-steps/s" and e["ms_per_step"] > 0 and e["steps"] == 2 and e["warmup"] == 1
        assert e["envs"] == envs and e["tracker_dropout"] == p
    assert "7176x10728" in d["dropout_on"]["workload"] and "1411x3327" in d["c2"]["workload"]
