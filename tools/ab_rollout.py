"""Same-box A/B of ROLLOUT kernels between library builds (box-to-box noise is +-2 %): every library runs in its own subprocess, alternating, and
times back-to-back collects of the C3 / C2 engine with every env forced to the full episode length (the bench's steady-state regime without
training first).   python tools/ab_rollout.py [c3|c2] [--dropout P] [--rounds R] lib_a.so lib_b.so ...
A library path of '-' is the in-tree build.  Prints ms per collect and us per vector step for every (round, library)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
    import torch
    from cirs_hip import abi
    if os.environ.get("CIRS_AB_LIB", "-") != "-":
        abi.LIB_PATH = os.environ["CIRS_AB_LIB"]
        import ctypes
        import torch  # noqa: F401  (its HIP runtime first, see abi.lib)
        h = ctypes.CDLL(abi.LIB_PATH)
        for name in list(abi.SIGNATURES):      # an older build may lack entry points added since (only the rollout is timed here)
            if not hasattr(h, name):
                del abi.SIGNATURES[name]
    import bench
    wl = bench.WORKLOADS[os.environ["CIRS_AB_WL"]]
    eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"), dropout=float(os.environ.get("CIRS_AB_DROPOUT", "0")))
    eng.rollout.force_length = wl["T"]
    for _ in range(3):
        eng.collect()
    torch.cuda.synchronize()
    reps = int(os.environ.get("CIRS_AB_REPS", "20"))
    out = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            eng.collect()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    print("AB_RESULT " + json.dumps({"ms_per_collect": min(out), "all": out, "us_per_vector_step": 1e3 * min(out) / wl["T"]}), flush=True)


if __name__ == "__main__":
    if os.environ.get("CIRS_AB_WORKER"):
        worker()
        sys.exit(0)
    args = sys.argv[1:]
    wl = "c3"
    if args and args[0] in ("c2", "c3"):
        wl = args.pop(0)
    dropout, rounds = "0", 2
    while args and args[0].startswith("--"):
        k = args.pop(0)
        if k == "--dropout": dropout = args.pop(0)
        elif k == "--rounds": rounds = int(args.pop(0))
    libs = args or ["-"]
    for rnd in range(rounds):
        for lib in libs:
            env = dict(os.environ, CIRS_AB_WORKER="1", CIRS_AB_LIB=lib if lib == "-" else os.path.abspath(lib), CIRS_AB_WL=wl, CIRS_AB_DROPOUT=dropout)
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("AB_RESULT ")]
            print(wl, "dropout", dropout, "round", rnd, os.path.basename(lib), line[0][10:] if line else ("FAILED " + r.stderr[-400:]), flush=True)
