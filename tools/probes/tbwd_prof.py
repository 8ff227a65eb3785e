"""Stage timestamps inside the tracker-BPTT row-chain kernels (probe build: tools/probes/build_prof_lib.sh), C3 shape.
    python tools/probes/tbwd_prof.py"""
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
lib = C.CDLL(abi.LIB_PATH)
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
names = {k: v for k, v in enumerate(sys.argv[1].split("|"))} if len(sys.argv) > 1 else {k: f"stamp {k}" for k in range(12)}
print("raw s_memtime ticks (100 MHz), workgroup 300, thread 0 (the LAST launch of the kernel in the update):")
prev = t[0]
for k in sorted(names):
    print(f"  {k:2d} {names[k]:40s} {t[k] - prev:9.0f}   (cum {t[k] - t[0]:9.0f})")
    prev = t[k]
