"""Stage timestamps inside attn_bwd_ep (probe build: tools/probes/build_prof_lib.sh), C3 or C2 shape, the bench's steady state.
    python tools/probes/attn_prof.py [c3|c2]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
wl = bench.WORKLOADS[name]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"), dropout=0.1)
lib = C.CDLL(abi.LIB_PATH)
for warm in range(40 if name == "c3" else 150):
    eng.collect(); eng.update(1024, 2)
acc = None
for rep in range(8):
    eng.collect(); eng.update(1024, 1)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    assert lib.cirs_debug_tbwd_prof(buf) == 0
    t = np.array(buf[:], dtype=np.float64)
    if rep >= 2:
        acc = t if acc is None else acc + t
t = acc / 6
print(f"attn_bwd_ep ({name}), workgroup 300 (c2: none -> zeros), thread 0, raw s_memtime ticks")
names = {39: "entry", 38: "Q|K|V + dATT staged", 40: "keep bits + barrier", 45: "phase A loop 1 (scores, max)", 46: "phase A loop 2 (exp, sum, dot)",
         41: "phase A loop 3 (dS, dQ) + stores + barrier", 42: "phase B (keys: dK, dV)"}
prev = t[39]
for k in (39, 38, 40, 45, 46, 41, 42):
    print(f"  {names[k]:44s} {t[k] - prev:9.0f}")
    prev = t[k]
