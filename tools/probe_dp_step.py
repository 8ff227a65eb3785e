"""Per-rank time of a data-parallel minibatch step at this rank's share of the rows: the C3 engine is trained for a few updates, then
DeviceLearner.learn_dp runs one update as rank 0 of W ranks with an identity all-reduce (compute only: the wire time is not in it) and the whole loop is
timed with HIP events.   python tools/probe_dp_step.py [--worlds 1 2 4 8]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import torch
import bench

worlds = [int(x) for x in sys.argv[sys.argv.index("--worlds") + 1:]] if "--worlds" in sys.argv else [1, 2, 4, 8]
wl = bench.WORKLOADS["c3"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"), dropout=0.1)
for _ in range(20):
    eng.collect(); eng.update(1024, 2)
eng.collect(); eng.update(1024, 2)
ln = eng.learner
snap = [t.clone() for t in (ln.params, ln.adam_m, ln.adam_v)]
out = []
for W in worlds:
    for rep in range(3):
        for t, s in zip((ln.params, ln.adam_m, ln.adam_v), snap):
            t.copy_(s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        if W == 1:
            losses = ln.learn(1024, 2, want_tracker_grad=False)
        else:
            losses = ln.learn_dp(1024, 2, None, 0, W, lambda t: None, want_tracker_grad=False)
        b.record(); torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / losses.shape[0]
    out.append({"world": W, "rows_per_rank": 1024 // W, "steps": int(losses.shape[0]), "us_per_step": round(us, 2)})
    print(json.dumps(out[-1]), flush=True)
for t, s in zip((ln.params, ln.adam_m, ln.adam_v), snap):
    t.copy_(s)
