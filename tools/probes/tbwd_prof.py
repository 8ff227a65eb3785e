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
for warm in range(30):      # the regime the bench measures: a trained policy plays full-length episodes
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
FWD = ["entry", "loads arrived", "out_proj MFMAs (16)", "LDS transpose, residual", "LayerNorm1", "stores, barrier", "lin1 (64 MFMAs) + FF1 stores",
       "lin2 (64 MFMAs)", "transpose, residual", "LayerNorm2", "stores", "next in_proj (48 MFMAs) + stores"]
BWD = ["entry", "pre stage: in_proj backward (48 MFMAs), dH", "LayerNorm2 backward, dB2", "lin2 backward (64 MFMAs), relu gate", "dFF1 rows: LDS -> regs, stores",
       "lin1 backward (64 MFMAs)", "transpose, residual, dH1N", "LayerNorm1 backward, dY1", "out_proj backward (16 MFMAs), dATT"]
print("raw s_memtime ticks, workgroup 300, thread 0 (stamps of the LAST launch of each kernel in the update)")
for title, names, base in (("layer_rows_fwd", FWD, 0), ("layer_rows_bwd (the pre-stage-only launch for layer 0 overwrites stamps 20-21)", BWD, 20)):
    print(title)
    prev = t[base]
    for k, nm in enumerate(names):
        print(f"  {k:2d} {nm:52s} {t[base + k] - prev:9.0f}   (cum {t[base + k] - t[base]:9.0f})")
        prev = t[base + k]
