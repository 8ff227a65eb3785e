"""Same-box A/B of the PPO MINIBATCH STEP between library builds (box-to-box noise is +-2 %): every library runs in its own subprocess (CIRS_HIP_LIB), alternating;
each trains the C3 engine for a few updates (the trained regime: sharp rows, full-length episodes), then times cirs_ppo_learn's loop with bench.hip_event_kernel_time.
    python tools/ab_step_libs.py [--pre N] [--rounds R] lib_a.so lib_b.so ...      ('-' = the in-tree build)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if os.environ.get("CIRS_AB_WORKER"):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
    import torch
    import bench
    wl = bench.WORKLOADS["c3"]
    eng, _ = bench.build_engine(wl, 0, 1, torch.device("cuda:0"), dropout=0.1)
    for _ in range(int(os.environ.get("CIRS_AB_PRE", "20"))):
        eng.collect(); eng.update(1024, 2)
    eng.collect(); losses, n = eng.update(1024, 2)
    t, mb, tk = bench.hip_event_kernel_time(eng, wl, reps=150)
    print("AB_RESULT " + json.dumps({"step_us": round(1e6 * t, 2), "mb": mb, "kernels_us": {k: round(1e6 * v, 2) for k, v in tk.items()},
                                     "last_losses": [round(float(x), 6) for x in losses[-1].cpu().numpy()], "rows": n}), flush=True)
    sys.exit(0)

args = sys.argv[1:]
pre, rounds = "20", 2
while args and args[0].startswith("--"):
    k = args.pop(0)
    if k == "--pre": pre = args.pop(0)
    elif k == "--rounds": rounds = int(args.pop(0))
for rnd in range(rounds):
    for lib in (args or ["-"]):
        env = dict(os.environ, CIRS_AB_WORKER="1", CIRS_AB_PRE=pre)
        if lib != "-":
            env["CIRS_HIP_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB_RESULT ")]
        print("round", rnd, os.path.basename(lib), line[0][10:] if line else ("FAILED " + r.stderr[-600:]), flush=True)
