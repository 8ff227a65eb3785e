"""A/B of the two actor-head paths of a PPO minibatch step (CIRS_PPO_HEAD=fused | split) on one box, trained regime, C3 catalogue:
whole cirs_ppo_minibatch call (HIP events on the launch stream) for several minibatch row counts.
    python tools/ab_head_paths.py [pre_updates] [reps]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import torch
import bench
from cirs_hip import abi

pre = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl = bench.WORKLOADS["c3"]
eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"))
for _ in range(pre):
    eng.collect(); eng.update(1024, 2)
ln = eng.learner
lib = ln._lib
snap = [t.clone() for t in (ln.params, ln.adam_m, ln.adam_v)]
out = {}
for mb in [int(x) for x in os.environ.get("AB_MB", "64,128,256,512,1024").split(",")]:
    ws = ln.workspace(mb)
    idx = torch.arange(mb, dtype=torch.int32, device=eng.device)
    losses = torch.zeros(4, dtype=torch.float32, device=eng.device)
    for mode in os.environ.get("AB_MODES", "fused,split").split(","):
        os.environ["CIRS_PPO_HEAD"] = mode

        def run(k):
            for _ in range(k):
                abi.check(lib.cirs_ppo_minibatch(C.byref(ln.cfg), ln.params.data_ptr(), ln.grads.data_ptr(), ln.adam_m.data_ptr(), ln.adam_v.data_ptr(),
                                                 ln.opt_step, C.byref(ln.batch), idx.data_ptr(), mb, None, ln.n_env, losses.data_ptr(), ws.data_ptr(),
                                                 ws.numel(), ln._stream()), "probe")
        run(5)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); run(reps); e1.record(); torch.cuda.synchronize()
        out[f"{mode}_{mb}"] = e0.elapsed_time(e1) / reps * 1e3
        for t, s in zip((ln.params, ln.adam_m, ln.adam_v), snap):
            t.copy_(s)
    print(f"mb {mb:5d}: " + "   ".join(f"{k.split('_')[0]} {v:7.1f} us" for k, v in out.items() if k.endswith(f"_{mb}")), flush=True)
print(json.dumps(out))
