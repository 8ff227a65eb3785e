"""prefix_env_kernel (the one-launch prefix pass of the exact-redraw collect): launch time with and without dropout at the C3 shape (1024 envs, prefixes
of `rows` positions) and its stage timestamps (probe build: tools/probes/build_prof_lib.sh).   python tools/probes/prefix_prof.py [rows]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from cirs_hip import abi

PROF = os.path.join(ROOT, "tools", "probes", "libcirs_prof.so")
if os.path.exists(PROF) and not os.environ.get("CIRS_HIP_LIB"):
    abi.LIB_PATH = PROF
import rolloutcase
from cirs_hip.tracker import DeviceTracker

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 30
U, I, B, T = 7176, 10728, 1024, 30
tp = rolloutcase.tracker_param_dict(U, I, T, seed=3)
trk = DeviceTracker({k: v.float().cuda().contiguous() for k, v in tp.items()}, U, I, B, T)
rng = np.random.RandomState(0)
trk.reset(); trk.init(torch.as_tensor(rng.randint(0, U, B)))
for t in range(T):
    trk.step(torch.as_tensor(rng.randint(0, I, B)), torch.as_tensor(rng.uniform(0, 1, B)))
lens = np.full(B, rows, np.int32)
offsets = (np.arange(B) * rows).astype(np.int32)
row_env = np.repeat(np.arange(B), rows).astype(np.int32); row_t = np.tile(np.arange(rows), B).astype(np.int32)
dd = lambda x: torch.as_tensor(x).cuda()  # noqa: E731
args = (dd(row_env), dd(row_t), dd(offsets), dd(lens), B * rows)
out = torch.zeros(B, 20, device="cuda")
lib = C.CDLL(abi.LIB_PATH)
for p in (0.1, 0.0):
    trk.set_dropout(p)
    if p > 0:
        trk.set_dropout_key(5, 1, 0)
    for mode in ("one launch", "launches"):
        if mode == "launches":
            os.environ["CIRS_TRACKER_PREFIX_LAUNCHES"] = "1"
        else:
            os.environ.pop("CIRS_TRACKER_PREFIX_LAUNCHES", None)
        for _ in range(20):
            trk.prefix_states(*args, out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            trk.prefix_states(*args, out)
        b.record(); torch.cuda.synchronize()
        print(f"p = {p}: {mode:10s} {a.elapsed_time(b) / 200 * 1e3:7.1f} us per pass ({rows} rows per env)")
    os.environ.pop("CIRS_TRACKER_PREFIX_LAUNCHES", None)
    if hasattr(lib, "cirs_debug_tbwd_prof"):
        trk.prefix_states(*args, out); torch.cuda.synchronize()
        buf = (C.c_ulonglong * 64)()
        assert lib.cirs_debug_tbwd_prof(buf) == 0
        t = np.array(buf[:], dtype=np.float64)
        names = {40: "entry", 41: "slot gather + in_proj (48 MFMAs)", 42: "layer 0 attention (all queries)", 43: "layer 0 chain (192 MFMAs)",
                 44: "layer 1 attention (last query)", 45: "layer 1 chain (144 MFMAs)", 50: "decoder"}
        prev = t[40]
        print("  stage stamps of workgroup 300 (s_memtime ticks, 100 MHz):")
        for k in (40, 41, 42, 43, 44, 45, 50):
            print(f"    {names[k]:40s} {t[k] - prev:8.0f}   (cum {t[k] - t[40]:8.0f})")
            prev = t[k]
