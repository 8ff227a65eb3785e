"""Repeat bench.py's DeepFM sweep probe a few times (run-to-run spread of the secondary metric M3)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import torch
import bench

wl = bench.WORKLOADS["c3"] if hasattr(bench, "WORKLOADS") else None
dev = torch.device("cuda:0")
for rep in range(4):
    r = bench.deepfm_sweep_probe(wl, dev, reps=10)
    print(rep, "pairs/s %.3e  s/sweep %.4f" % (r["pairs_per_s"], r["seconds_per_sweep"]), flush=True)
