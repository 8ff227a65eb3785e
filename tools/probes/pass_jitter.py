"""bench.py's pass sequence (C3, C3 with dropout, C2; fresh engine, 5 warm-up + 20 timed steps each, NO host sync inside a pass) with a GPU event and a
host timestamp per step: where does a slow pass lose its time?      python tools/probes/pass_jitter.py [repeats]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np
import torch
import bench

dev = torch.device("cuda:0")
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for name, wl, p in (("c3", "c3", 0.0), ("c3+dropout", "c3", 0.1), ("c2", "c2", 0.0)):
        eng, _ = bench.build_engine(bench.WORKLOADS[wl], 0, 1, dev, dropout=p)
        idle = float(os.environ.get("PJ_IDLE", "0"))
        if idle > 0:
            torch.cuda.synchronize(); time.sleep(idle)             # let the GPU fall idle (clocks drop) before the short warm-up
        if os.environ.get("PJ_SPIN"):
            bench.spin_up(dev)
        for _ in range(5):
            eng.collect(); eng.update(1024, 2)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        host = [time.perf_counter()]
        ev[0].record()
        for k in range(20):
            eng.collect(); eng.update(1024, 2)
            ev[k + 1].record(); host.append(time.perf_counter())
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        g = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(20)])
        h = np.diff(np.array(host)) * 1e3
        print(f"rep {rep} {name:11s}: total {1e3 * (t_end - host[0]) / 20:.3f} ms/step | gpu per step median {np.median(g):.3f} max {g.max():.3f} (step {int(g.argmax())}) | "
              f"host enqueue per step median {np.median(h):.3f} max {h.max():.3f} (step {int(h.argmax())})", flush=True)
        del eng
        torch.cuda.empty_cache()
