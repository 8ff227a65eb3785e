"""Per-tile stage timestamps of head_bwd_fused_kernel (probe build: tools/probes/build_prof_lib.sh), C3 minibatch of 1024 rows.
    python tools/probes/head_prof.py"""
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
eng.collect(); eng.update(1024, 1)
lib = C.CDLL(abi.LIB_PATH)
names = {0: "tile start", 1: "LDS operand reads (za, bias)", 2: "logits MFMAs (24) + dWa sum of the previous tile", 3: "dZ (exp2, coefficients, mask)",
         4: "dZ^T -> LDS (+ branch-free clamp correction)", 5: "cb planes, split + dH2 MFMAs (24)", 6: "fence + dZ^T read back", 7: "split + dWa MFMAs (24)",
         9: "partial-tile writes + plane commit", 10: "the tile's one barrier"}
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
prev = t[0]
print("head_bwd_fused_kernel, workgroup (0,0) wave 0, third tile (raw s_memtime ticks):")
for k in sorted(names):
    print(f"  {names[k]:36s} {t[k] - prev:9.0f}   (cum {t[k] - t[0]:9.0f})")
    prev = t[k]

ks = {30: "kernel entry", 31: "H2 operands loaded + split (hz, hb), row scalars", 32: "first planes tile staged + barrier", 33: "all tiles", 34: "last tile's dWa sum",
      35: "dH2 / entropy partial stores"}
print("head_bwd_fused_kernel, workgroup (0,0) thread 0, whole kernel:")
prev = t[30]
for k in sorted(ks):
    print(f"  {ks[k]:52s} {t[k] - prev:9.0f}   (cum {t[k] - t[30]:9.0f})")
    prev = t[k]
tb = {16: "row workgroup 0: entry", 17: "W2/W1 columns requested, 31 dh2 slabs summed, da2 written", 18: "barrier", 19: "da1 tile (32 MFMAs), relu gate, stores", 20: "barrier",
      21: "dW1 tile (wave 0)"}
print("trunk_bwd_kernel (raw ticks; the stamps of thread 0 = wave 0):")
prev = t[16]
for k in sorted(tb):
    print(f"  {tb[k]:64s} {t[k] - prev:9.0f}   (cum {t[k] - t[16]:9.0f})")
    prev = t[k]
print(f"  first wa|ba slab-sum workgroup: entry at {t[24] - t[16]:+.0f} relative to row workgroup 0, its sums + reduction take {t[25] - t[24]:.0f}")

hs = {40: "head_stats workgroup (0,0): entry", 41: "H2 tile coalesced -> LDS -> split into bf16 planes", 42: "first planes tile staged, barrier",
      43: "(tiles 0, 1)", 44: "tile 2: next tile requested, LDS operand reads, bias init", 45: "tile 2: 24 MFMAs", 46: "tile 2: mask, max, 16 exp, running sums",
      47: "tile 2: commit next + barrier", 48: "remaining tiles"}
print("head_stats_kernel (raw ticks; two workgroups per CU share the SIMDs):")
prev = t[40]
for k in sorted(hs):
    print(f"  {hs[k]:64s} {t[k] - prev:9.0f}   (cum {t[k] - t[40]:9.0f})")
    prev = t[k]
