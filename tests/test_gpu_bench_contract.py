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
    # the headline runs the reference's training mode (tracker never in eval(): Dropout(0.1) live) and says how
    assert cfg["rows_per_rank_per_minibatch"] == 1024 and cfg["tracker_dropout"] == 0.1 and d["dropout"] == 0.1
    assert cfg["dropout_mode"].startswith("position-keyed") and 1.0 <= cfg["mean_episode_len"] <= 30.0
    assert abs(cfg["env_steps_per_step"] - cfg["mean_episode_len"] * 64) < 1e-6
    assert c["envs_gpu_leg"] == 64 and c["envs_cpu_leg"] >= 1
    # the last object of the line repeats what a reader of a log tail needs
    assert list(d)[-1] == "summary" and d["summary"]["value"] == d["value"] and d["summary"]["ms_per_step"] == d["ms_per_step"]


def test_bench_dropout_mode_line():
    """--dropout 0: the eval-mode tracker of the parity fixtures is measurable and labelled."""
    d = run_bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "c2", "--dropout", "0", "--no-probes")
    assert d["dropout"] == 0.0 and d["config"]["tracker_dropout"] == 0.0 and d["config"]["dropout_mode"].startswith("off") and d["value"] > 0



def test_bench_default_workload_carries_dropout_and_c2_passes():
    """the default (C3) single-GPU line runs the tracker in training mode and also carries a timed pass with the
    eval-mode tracker (the headline's warm-up / step protocol) and a timed pass of BASELINE configs[1] (C2, at its steady state) -- and their numbers sit in
    `config.also_measured`, which the driver's record keeps whole."""
    d = run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert "7176x10728" in d["config"]["workload"] and d["dropout"] == 0.1
    for key, envs, p in (("dropout_off", 1024, 0.0), ("c2", 64, 0.1)):
        e = d[key]
        assert e["value"] > 0 and e["unit"] == "env-steps/s" and e["ms_per_step"] > 0
        # the C3 pass repeats the headline's protocol; the C2 pass (1.7 ms per step) always runs to its steady state: >= 150 warm-up, >= 100 timed steps
        assert (e["steps"], e["warmup"]) == ((2, 1) if key == "dropout_off" else (100, 150))
        assert e["envs"] == envs and e["tracker_dropout"] == p
        a = d["config"]["also_measured"][key]
        assert a["value"] == e["value"] and a["ms_per_step"] == e["ms_per_step"] and a["mean_episode_len"] == e["mean_episode_len"] and a["steps"] == e["steps"]
    assert "7176x10728" in d["dropout_off"]["workload"] and "1411x3327" in d["c2"]["workload"]
    am = d["config"]["also_measured"]
    assert am["rollout_only_env_steps_per_s"] > 0 and am["minibatch_step_us"] > 0 and am["minibatch_step_launches"] == 4


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """the driver's command shape is `python3 bench.py --gpus N` with NO launcher around it.  bench.py must start its
    own ranks (torch.distributed.run), print the one JSON line from rank 0 and return rank 0's exit code.  On this 1-GPU box the ranks
    share device 0 and talk over gloo (CIRS_BENCH_SHARE_GPU=1): everything but the transport is the driver's N = 2 run."""
    env = dict(os.environ, CIRS_BENCH_SHARE_GPU="1", CIRS_DIST_CHECK="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    z = json.loads(lines[0])
    assert z["n_gpus"] == 2 and z["config"]["envs_total"] == 2048 and z["steps"] == 2 and z["warmup"] == 1 and z["value"] > 0
    assert z["rank_parameters_bit_identical"] is True and z["scaling"] == "weak"


def test_bench_gpus_n_without_enough_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("an 8-GPU node runs the job")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CIRS_BENCH_SHARE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "GPU(s)" in out.stderr
