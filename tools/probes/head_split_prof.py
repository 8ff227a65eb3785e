"""Stage timestamps of head_fwd_kernel / head_dwa_kernel (probe build: tools/probes/build_prof_lib.sh), C3 minibatch of 1024 rows, trained regime.
    python tools/probes/head_split_prof.py"""
import ctypes as C
import os
import sys

os.environ["CIRS_PPO_HEAD"] = "split"      # the kernels this probe stamps

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np
import torch
from cirs_hip import abi

abi.LIB_PATH = os.path.join(ROOT, "tools", "probes", "libcirs_prof.so")
import bench

wl = bench.WORKLOADS["c3"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):     # probe in the regime the bench runs in
    eng.collect(); eng.update(1024, 2)
lib = C.CDLL(abi.LIB_PATH)
acc = None
for rep in range(8):
    bench.hip_event_kernel_time(eng, wl, reps=2)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    assert lib.cirs_debug_head_prof(buf) == 0
    t = np.array(buf[:], dtype=np.float64)
    if rep >= 2:
        acc = t if acc is None else acc + t
t = acc / 6


def show(title, names, base):
    print(title)
    prev = t[base]
    for k in sorted(names):
        print(f"  {names[k]:72s} {t[k] - prev:9.0f}   (cum {t[k] - t[base]:9.0f})")
        prev = t[k]


show("head_fwd_kernel, workgroup (0,0) thread 0 (raw s_memtime ticks):",
     {40: "entry", 41: "H2 planes requested, first Wa tile staged, barrier", 42: "all tiles", 43: "O' slab + partial stores (write-through)"}, 40)
show("head_fwd_kernel, third iteration:",
     {44: "iteration start", 45: "next tile requested; [O'_k-1 | z, t, max], reference test, exp, action test", 46: "[L_k+1 | sums, splits], C planes read, plane commit", 47: "barrier"}, 44)
show("head_dwa_kernel, workgroup (0,0) thread 0:",
     {30: "entry", 31: "Wa tile planes + bias + first H2 tile requested, mask cleared, barrier", 32: "row scalars -> LDS, action masks", 33: "wait for the first H2 tile, barrier",
      34: "d h2 fold of the slice, Wa planes of the item tile", 35: "all row tiles", 36: "dWa tile / dba / entropy stores"}, 30)
show("head_dwa_kernel, third iteration:",
     {0: "iteration start", 1: "[D_j-1 | dZ_j] [L_j+1 | split_j] (one block)", 2: "wait for the LDS-DMA, barrier"}, 0)
