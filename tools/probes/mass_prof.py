"""Stage timestamps of actor_mass_kernel inside the fused C3 rollout (probe build: tools/probes/build_prof_lib.sh).
    python tools/probes/mass_prof.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np
import torch
from cirs_hip import abi

abi.LIB_PATH = os.path.join(ROOT, "tools", "probes", "libcirs_prof.so")
import bench

wl = bench.WORKLOADS["c3"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"))
eng.rollout.force_length = wl["T"]
lib = C.CDLL(abi.LIB_PATH)
names = {0: "entry", 1: "hidden rows requested + split, 'any env alive' vote", 2: "first chunk staged (global -> LDS, barrier)",
         15: "(first chunk) -> top of the SECOND chunk", 3: "next chunk requested; 4 tiles: operand reads + 48 MFMAs", 11: "mask + chunk maximum", 12: "64 x exp2, sum",
         13: "log2, store", 14: "commit of the next chunk + barrier", 16: "(third chunk) -> kernel end"}
order = [0, 1, 2, 15, 3, 11, 12, 13, 14, 16]
acc = None
for rep in range(6):
    eng.collect()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    assert lib.cirs_debug_mass_prof(buf) == 0
    t = np.array(buf[:], dtype=np.float64)
    if rep >= 1:
        acc = t if acc is None else acc + t
t = acc / 5
prev = t[0]
print("actor_mass_kernel, workgroup (0,0) wave 0 at the LAST step of the rollout (raw s_memtime ticks; the per-chunk stamps belong to its second chunk):")
for k in order:
    print(f"  {names[k]:60s} {t[k] - prev:9.0f}   (cum {t[k] - t[0]:9.0f})")
    prev = t[k]
