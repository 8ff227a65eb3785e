"""GPU: the step-for-step walk of the reference's entry point (examples/cirs_rl_kuaishou.py == CIRS-RL-kuaishou.py:119-345 on a
synthetic KuaiRec-format workspace) runs end to end through the mirrored plugin surface: gym.register / make, KuaishouEnv.load_mat
from the files, the user-model artefacts, StateTrackerTransformer / Net / Actor / Critic / PPOPolicy, Collector / CollectorSet,
SummaryWriter + BasicLogger, load_item_feat / get_training_item_domination, the two callbacks, onpolicy_trainer with save_model_fn,
and the four-key checkpoint."""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_entry_point_walk_runs_end_to_end(tmp_path):
    spec = importlib.util.spec_from_file_location("cirs_rl_kuaishou_entry", os.path.join(ROOT, "examples", "cirs_rl_kuaishou.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    ws = str(tmp_path / "ws")
    old_dp = os.environ.get("CIRS_DATAPATH")
    try:
        out = ex.run(["--workspace", ws, "--epoch", "2", "--step-per-epoch", "250", "--training-num", "16", "--test-num", "16",
                      "--episode-per-collect", "16", "--batch-size", "64", "--max_turn", "10", "--leave_threshold", "1",
                      "--num_leave_compute", "3", "--force_length", "5", "--is_save", "--save-interval", "1", "--tau", "10"])
    finally:
        if old_dp is None:
            os.environ.pop("CIRS_DATAPATH", None)
        else:
            os.environ["CIRS_DATAPATH"] = old_dp
    res = out["result"]
    for key in ("train_step", "train_episode", "test_step", "test_episode", "best_reward", "best_result", "duration"):
        assert key in res, (key, res)
    assert res["train_step"] >= 500 and res["test_episode"] == 3 * 16    # the pre-training evaluation + one per epoch (FB collector)
    ckpt_path = os.path.join(ws, out["model_save_path"])
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    assert sorted(ckpt) == ["optim_RL", "optim_state", "policy", "state_tracker"]
    assert any(k.startswith("actor.last.model.0") for k in ckpt["policy"]) and "embedding_dict.feat_item.weight" in ckpt["state_tracker"]
    for e in (1, 2):     # --is_save: one checkpoint per epoch through save_model_fn
        assert os.path.isfile(ckpt_path[:-3] + f"-e{e}.pt")
    # the log file the script attached with logzero.logfile carries the epoch lines of LoggerCallback_Policy
    log = open(os.path.join(ws, out["logger_path"])).read()
    assert "Epoch: [1], Info: [" in log and "Epoch: [2], Info: [" in log and "NX_5_ctr" in log and "CV_turn" in log and "ifeat_feat" in log
    # BasicLogger wrote through the SummaryWriter (the stand-in keeps a jsonl; with tensorboard installed there are event files)
    log_dir = os.path.join(ws, out["log_dir"])
    jl = os.path.join(log_dir, "scalars.jsonl")
    # (the reference's trainer passes global_step=None to test_episode, so test statistics are never written, and train / update
    # statistics only every 1000 steps: with this short run the writer stays empty -- what is pinned is that it was created there)
    assert os.path.isfile(jl) or any(f.startswith("events.out.tfevents") for f in os.listdir(log_dir))
    cb = out["callbacks"][1]
    assert cb.last_results is not None and np.isfinite(float(cb.last_results["CV"]))
