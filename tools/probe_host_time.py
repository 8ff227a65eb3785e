import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/cirs-codes_amd")
import torch, bench
wl = bench.WORKLOADS["c3"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"))
for _ in range(5): eng.collect(); eng.update()
torch.cuda.synchronize()
tc = tu = ts1 = ts2 = 0
N = 30
for _ in range(N):
    t0 = time.perf_counter(); eng.collect(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    eng.update(); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    tc += t1 - t0; ts1 += t2 - t1; tu += t3 - t2; ts2 += t4 - t3
print("G", os.environ.get("CIRS_ROLLOUT_GROUPS"), "collect host %.3f ms, sync %.3f | update host %.3f ms, sync %.3f" % (tc/N*1e3, ts1/N*1e3, tu/N*1e3, ts2/N*1e3))
# the bench's loop: no synchronisation between collect and update except update()'s own read-back of the episode lengths
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        eng.collect(); eng.update(batch_size=1024, repeat=2)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("G", os.environ.get("CIRS_ROLLOUT_GROUPS"), "free-running step %.3f ms" % ((t1 - t0) / N * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    eng.collect(); eng.update(batch_size=1024, repeat=2); torch.cuda.synchronize()
t1 = time.perf_counter()
print("G", os.environ.get("CIRS_ROLLOUT_GROUPS"), "step with a sync after update %.3f ms" % ((t1 - t0) / N * 1e3))
