"""Entry / exit times of every workgroup of one layer_rows_fwd launch (probe build): how many run at once, on which CUs."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np, torch
from cirs_hip import abi
abi.LIB_PATH = os.path.join(ROOT, "tools", "probes", "libcirs_prof.so")
import bench
wl = bench.WORKLOADS["c3"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"))
lib = C.CDLL(abi.LIB_PATH)
for warm in range(30):      # the regime the bench measures: a trained policy plays full-length episodes
    eng.collect(); eng.update(1024, 2)
for rep in range(4):
    eng.collect(); eng.update(1024, 1)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2048 * 3))()
assert lib.cirs_debug_tbwd_wg(buf) == 0
t = np.array(buf[:], dtype=np.uint64).reshape(2048, 3)
n = int((t[:, 0] > 0).sum())
t = t[:n]
e, x = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
t0 = e.min()
print(f"{n} workgroups; entries span {e.max() - t0:.0f} ticks, last exit at {x.max() - t0:.0f}; busy per workgroup: median {np.median(x - e):.0f}, max {(x - e).max():.0f}")
hw = t[:, 2]
xcc = (hw >> np.uint64(32)).astype(np.int64); hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7; simd = (hwid >> 4) & 3
key = xcc * 100000 + se * 1000 + sh * 100 + cu
u, cnt = np.unique(key, return_counts=True)
print(f"distinct CUs used: {len(u)}; workgroups per CU: min {cnt.min()} max {cnt.max()}; histogram {np.bincount(cnt)}")
ks = key * 10 + simd
u2, c2 = np.unique(ks, return_counts=True)
print(f"distinct SIMDs used: {len(u2)}; per SIMD histogram {np.bincount(c2)}")
order = np.argsort(e)
pb = (C.c_ulonglong * 64)()
assert lib.cirs_debug_tbwd_prof(pb) == 0
print(f"workgroup 300 (stamps 40-42): staging done -> phase A {pb[41] - pb[40]} ticks, phase B {pb[42] - pb[41]} ticks")
for xc in np.unique(xcc):
    m = xcc == xc
    ee, xx = e[m], x[m]
    print(f"XCD {xc}: {m.sum()} workgroups, entries spread over {ee.max() - ee.min():.0f} ticks, first entry -> last exit {xx.max() - ee.min():.0f} ticks, busy median {np.median(xx - ee):.0f}")
bz = x - e
print("busy percentiles 0/10/50/90/99/100:", np.percentile(bz, [0, 10, 50, 90, 99, 100]).round(0))
slow = np.argsort(-bz)[:12]
print("slowest workgroups (block, busy, xcc, se, cu, simd):", [(int(i), int(bz[i]), int(xcc[i]), int(se[i]), int(cu[i]), int(simd[i])) for i in slow])
print("lens of the slowest:", [int(eng.lengths[i]) for i in slow], "median len", float(eng.lengths.float().median()))
print("entry deciles:", np.percentile(e - t0, [0, 10, 25, 50, 75, 90, 100]).round(0))
print("exit deciles:", np.percentile(x - t0, [0, 10, 25, 50, 75, 90, 100]).round(0))
