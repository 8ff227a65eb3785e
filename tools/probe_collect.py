"""Wall time of consecutive CirsEngine.collect() calls (C3 workload), with and without an update in between."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import torch
import bench
wl = bench.WORKLOADS["c3"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"))
for k in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.collect()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    n = int(eng.lengths.sum())
    print("collect %d: host %.2f ms, total %.2f ms, env-steps %d" % (k, (t1 - t0) * 1e3, (t2 - t0) * 1e3, n), flush=True)
    if k >= 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.update(1024, 2)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("   update: host %.2f ms, total %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
