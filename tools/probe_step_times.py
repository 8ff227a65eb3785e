import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "cirs-codes_amd"))
import torch, bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"))
ts = []
for i in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.collect(); t1 = time.perf_counter()
    losses, n = eng.update(batch_size=1024, repeat=2)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((1e3 * (t2 - t0), 1e3 * (t1 - t0), n, losses.shape[0]))
for i, t in enumerate(ts):
    if i < 12 or i % 8 == 0: print(i, "ms %.2f (collect enqueue %.2f) rows %d mb-steps %d" % t)
