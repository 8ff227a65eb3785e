"""Stage timestamps (s_memtime) of chosen workgroups of adam_next_kernel and trunk_bwd_kernel<true> inside cirs_ppo_learn's loop, C3 shape, trained
regime (probe build: bash tools/probes/build_prof_lib.sh).   python tools/probes/step_prof.py"""
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
for warm in range(20):
    eng.collect(); eng.update(1024, 2)
acc = []
for rep in range(6):
    eng.collect(); eng.update(1024, 1)        # (one pass: the stamps are those of the last step with a next step... the LAST launch overwrites)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    assert lib.cirs_debug_step_prof(buf) == 0
    acc.append(np.array(buf[:], dtype=np.float64))
t = np.mean(acc[1:], axis=0)

def show(title, base, items):
    print(title)
    for k, nm in items:
        print(f"  [{k:2d}] {nm:58s} {t[k] - t[base]:9.0f}")

print("ticks relative to the first stamp of the launch (s_memtime: 100 MHz on this part -> 1 tick = 10 ns)")
show("adam_next_kernel P / A roles (same launch)", 0,
     [(8, "P first: entry"), (9, "P first: loads + clip coefficient"), (10, "P first: Adam + stores + LDS image"), (11, "P first: planes written"),
      (12, "P last: entry"), (13, "P last: end"), (14, "A0 first: entry"), (15, "A0 first: end (arrival counted)"), (16, "A0 last: entry"), (17, "A0 last: end")])
show("adam_next_kernel T / S roles (the last launch of the update that had a next step)", 0,
     [(0, "T first: entry"), (1, "T first: gathers requested, A0 arrived"), (2, "T first: updated trunk read -> LDS"), (18, "T first: both layers (MFMA)"), (3, "T first: outputs written"),
      (4, "T last: entry"), (5, "T last: end"), (6, "S: entry"), (7, "S: end")])
show("trunk_rows_kernel", 20,
     [(20, "row wg 0: entry"), (21, "row wg 0: slabs summed, operands in LDS"), (22, "row wg 0: d a1"), (23, "row wg 0: weight-gradient slab stores issued"),
      (24, "row wg 0: slab stores drained"), (25, "row wg 0: arrival counted"), (26, "wa wg 0: entry"), (27, "wa wg 0: end"),
      (30, "F wg 0: entry"), (31, "F wg 0: all rows arrived"), (32, "F wg 0: slabs summed"),
      (33, "F wg 0: end")])
