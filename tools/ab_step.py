"""A/B of environment-switched variants of the PPO minibatch step on ONE box (box-to-box noise is +-2 %): trains the C3 engine for a few updates, then
times cirs_ppo_learn's loop (bench.hip_event_kernel_time) under each setting.   python tools/ab_step.py NAME=v1,v2,.. [NAME2=..] [--pre N]"""
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

args = [a for a in sys.argv[1:] if "=" in a]
pre = int(sys.argv[sys.argv.index("--pre") + 1]) if "--pre" in sys.argv else 20
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in args]
wl = bench.WORKLOADS["c3"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"), dropout=0.1)
for _ in range(pre):
    eng.collect(); eng.update(1024, 2)
eng.collect(); eng.update(1024, 2)
out = []
for rnd in range(2):
    for combo in itertools.product(*[v for _, v in axes]):
        for (k, _), val in zip(axes, combo):
            os.environ[k] = val
        t, mb, tk = bench.hip_event_kernel_time(eng, wl, reps=150)
        out.append((rnd, dict(zip([k for k, _ in axes], combo)), round(1e6 * t, 2)))
        print(out[-1], flush=True)
