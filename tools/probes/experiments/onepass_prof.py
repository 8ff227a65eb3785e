"""Phase stamps of head_onepass_kernel, workgroup (0, 0) thread 0 (probe build).   python tools/probes/onepass_prof.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np, torch
from cirs_hip import abi
abi.LIB_PATH = os.path.join(ROOT, "tools", "probes", os.environ.get("CIRS_PROF_LIB", "libcirs_prof.so"))
import bench
wl = bench.WORKLOADS["c3"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"), dropout=0.1)
lib = C.CDLL(abi.LIB_PATH)
for _ in range(20): eng.collect(); eng.update(1024, 2)
acc = []
for rep in range(5):
    eng.collect(); eng.update(1024, 1); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)(); assert lib.cirs_debug_head_prof(buf) == 0
    acc.append(np.array(buf[:], dtype=np.float64))
t = np.mean(acc[1:], axis=0)
names = [(50, "entry + setup"), (51, "phase A: logits + statistics of the workgroup's tiles"), (52, "partials written, flags, other chunks arrived"), (53, "merge + first C tile staged"),
         (33, "phase B: all tiles"), (54, "last dWa sum, dH2 / entropy stores")]
prev = t[50]
for k, nm in names:
    print(f"  [{k}] {nm:60s} {t[k] - prev:9.0f}  (cum {t[k] - t[50]:9.0f})"); prev = t[k]
print("  third tile of phase A:")
tn = [(41, "plane + H2 operand reads from LDS, bias"), (42, "24 MFMAs + accumulator sum"), (43, "action logit, max / exp / sums"), (44, "stash + commit of the next tile"), (45, "barrier")]
prev = t[40]
for k, nm in tn:
    print(f"  [{k}] {nm:60s} {t[k] - prev:9.0f}"); prev = t[k]
