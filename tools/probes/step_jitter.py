"""Per-step wall times of the timed loop of bench.py (collect + update, one host sync per step) -> outliers, and whether Python's GC ran in them.
    python tools/probes/step_jitter.py [workload] [steps]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np
import torch
import bench

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"))
gc_events = []
gc.callbacks.append(lambda phase, info: gc_events.append((time.perf_counter(), phase, info.get("generation"))))
for _ in range(5):
    eng.collect(); eng.update(1024, 2)
torch.cuda.synchronize()
ts = []
for k in range(steps):
    t0 = time.perf_counter()
    eng.collect(); eng.update(1024, 2)
    torch.cuda.synchronize()
    ts.append((t0, time.perf_counter()))
d = np.array([b - a for a, b in ts]) * 1e3
print(f"steps {steps}: median {np.median(d):.3f} ms, mean {d.mean():.3f}, p99 {np.percentile(d, 99):.3f}, max {d.max():.3f}")
for k in np.argsort(-d)[:8]:
    a, b = ts[k]
    g = [(ph, gen) for (t, ph, gen) in gc_events if a <= t <= b]
    print(f"  step {k}: {d[k]:.3f} ms  gc events inside: {g}")
