"""Time the PPO minibatch kernels (bench.hip_event_kernel_time) with an alternative build of the library:
    python tools/probe_kernel_lib.py path/to/lib.so [...]
Used for ablation builds of one kernel; one collect + prepare with the FIRST library's engine per process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import torch
from cirs_hip import abi
if len(sys.argv) > 1:
    abi.LIB_PATH = os.path.abspath(sys.argv[1])
import bench

wl = bench.WORKLOADS["c3"]
dev = torch.device("cuda:0")
eng, _ = bench.build_engine(wl, 0, 1, dev)
eng.collect()
eng.learner.prepare(eng.traj, eng.lengths_host() if hasattr(eng, "lengths_host") else None, eng.lengths) if False else None
eng.update(1024, 1)
t, mb, k = bench.hip_event_kernel_time(eng, wl)
import ctypes as C
lib = eng.learner._lib
abi.check(lib.cirs_prof_start(3, 64), "cirs_prof_start")
eng.collect(); torch.cuda.synchronize()
tot, cnt = C.c_double(0.0), C.c_int32(0)
abi.check(lib.cirs_prof_stop(C.byref(tot), C.byref(cnt)), "cirs_prof_stop")
k["actor_head_kernel"] = tot.value / max(cnt.value, 1)
print(os.path.basename(abi.LIB_PATH), "minibatch %.1f us" % (t * 1e6), {n: round(v * 1e6, 2) for n, v in k.items()}, flush=True)
