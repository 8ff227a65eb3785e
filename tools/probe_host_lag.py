"""Is the host ahead of the GPU in bench.py's free-running loop?  For every collect() / update() call: host time at return vs the time the GPU passes
an event recorded at that point.  lag = GPU event time - host return time: positive = the GPU still had queued work when the host returned (good);
about zero or negative = the GPU ran dry and waited for the host.  Also the host time spent inside each call.   python tools/probe_host_lag.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import numpy as np, torch, bench
wl = bench.WORKLOADS[os.environ.get("WL", "c3")]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"), dropout=0.1)
for _ in range(20): eng.collect(); eng.update(1024, 2)
torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()
N = 40
ev0 = torch.cuda.Event(enable_timing=True); ev0.record(); torch.cuda.synchronize(); h0 = time.perf_counter()
rec = []
for k in range(N):
    a = time.perf_counter(); eng.collect(); b = time.perf_counter()
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    c = time.perf_counter(); eng.update(1024, 2); d = time.perf_counter()
    e2 = torch.cuda.Event(enable_timing=True); e2.record()
    rec.append((a, b, e1, c, d, e2))
torch.cuda.synchronize(); h1 = time.perf_counter()
rows = []
for a, b, e1, c, d, e2 in rec[5:]:
    g1, g2 = ev0.elapsed_time(e1), ev0.elapsed_time(e2)
    rows.append(((b - a) * 1e3, (d - c) * 1e3, g1 - (b - h0) * 1e3, g2 - (d - h0) * 1e3))
r = np.array(rows)
print("step %.3f ms | host in collect() %.3f ms, in update() %.3f ms | lag of the GPU behind the host at collect() return %.3f (min %.3f) ms, at update() return %.3f (min %.3f) ms"
      % ((h1 - h0) / N * 1e3, r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), r[:, 2].min(), r[:, 3].mean(), r[:, 3].min()))
# inside update(): host time of the pieces
import cirs_hip.engine as E
ln = eng.learner
t = {}
def timed(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        s = time.perf_counter(); out = f(*a, **k); t[name] = t.get(name, 0.0) + time.perf_counter() - s; return out
    setattr(obj, name, g)
for o, n in ((ln, "prepare_async"), (ln, "finish_prepare"), (ln, "learn"), (ln, "_perms_on_device"), (eng.tracker, "backward"), (eng.tracker, "adam_update"), (eng.rollout, "collect")):
    timed(o, n)
torch.cuda.synchronize(); s = time.perf_counter()
for k in range(N): eng.collect(); eng.update(1024, 2)
torch.cuda.synchronize(); tot = time.perf_counter() - s
print("step %.3f ms; host ms per step inside: " % (tot / N * 1e3) + ", ".join(f"{k} {v / N * 1e3:.3f}" for k, v in t.items()))
