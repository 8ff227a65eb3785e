"""Stage timestamps of tracker_step_kernel inside the fused C3 rollout (probe build: tools/probes/build_prof_lib.sh).
    python tools/probes/trk_prof.py [c3|c2]"""
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

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"))
eng.rollout.force_length = wl["T"]     # every env runs all T steps: the stamps of the last launch belong to position T of env 0
lib = C.CDLL(abi.LIB_PATH)
names = {0: "start", 1: "lds image", 2: "merge (tail)", 3: "env step (tail)", 4: "slot+gate+pe", 5: "L0 in_proj+kv write", 6: "L0 scores+softmax", 7: "L0 V sum",
         8: "L0 out_proj+LN1", 9: "L0 FF1", 11: "L1 in_proj (+L0 lin2+LN2)", 12: "L1 scores+softmax", 13: "L1 V sum", 14: "L1 out_proj+LN1", 15: "L1 FF1",
         17: "L1 lin2+LN2", 18: "decoder", 19: "trunk"}
acc = None
for rep in range(6):
    eng.collect()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    assert lib.cirs_debug_trk_prof(buf) == 0
    t = np.array(buf[:], dtype=np.float64)
    if rep >= 1:
        acc = t if acc is None else acc + t
t = acc / 5
prev = t[0]
pick = {1: "kernel entry", 32: "prologue (pointers, lambda set-up)", 33: "env prefetch issued", 34: "sampler prefetch + position issued", 35: "gate row issued", 26: "chunk Gumbels (2 Philox, 2 det_logf pairs)", 20: "chunk masses arrive, fold", 21: "chunk arg-max + log-sum-exp reduce", 36: "bias / visited / 32 row loads issued", 38: "2 item Gumbels", 22: "log-sum-exp reduce of the masses",
        23: "rows 0-63: LDS transpose, 64 fma", 37: "rows 64-127 to LDS", 24: "caller's prefetch issued; 64 fma", 25: "item arg-max reduce", 2: "action / visited store"}
if t[20] > 0:
    print("inside 'merge (tail)' = the sampler's pick (raw ticks):")
    order = [1, 32, 33, 34, 35, 26, 20, 21, 36, 38, 22, 23, 37, 24, 25, 2]
    for a, b in zip(order[:-1], order[1:]):
        print(f"  {pick[b]:60s} {t[b] - t[a]:9.0f}")
if t[27] > 0:
    print("inside layer 0's feed-forward (raw ticks):")
    for a, b, nm in [(8, 27, "tmp write, lin2 half-row loads issued (16 x float4)"), (27, 28, "two 32-term dots from prefetched rows (waits for them)"),
                     (28, 9, "ffs write, next layer's in_proj rows issued (16 x float4)"), (9, 29, "64-term dot from the prefetched half row (waits for it)"),
                     (29, 30, "half-wave add + LayerNorm-2")]:
        print(f"  {nm:60s} {t[b] - t[a]:9.0f}")
print("stage deltas of workgroup 0 / wave 0 at the LAST step of the rollout (raw s_memtime ticks):")
for k in sorted(names):
    print(f"  {names[k]:32s} {t[k] - prev:9.0f}   (cum {t[k] - t[0]:9.0f})")
    prev = t[k]
