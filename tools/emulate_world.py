"""Per-rank compute time of a W-rank job on ONE GPU: this rank's W - 1 peers are copies of itself (cirs_hip.distributed.EmulatedPeers).

    python tools/emulate_world.py [--worlds 2 4 8] [--learners replicated dp dp_sharded tp] [--steps 20] [--warmup 8]

What is real: the shapes, the launch sequence and the kernel work of rank 0 of a W-rank bench.py job (1024 envs per rank, the gathered
buffer of W x 1024 envs, global minibatch 1024, the learner's per-rank share).  What is not: the collectives move no bytes over xGMI --
their calls and bytes are counted, their wire time has to be added from a link model.  A single-rank engine is
trained first (the bench's regime: a trained policy plays full-length episodes) and its parameters seed every emulated engine."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
import torch

import bench
from cirs_hip.distributed import EmulatedPeers


def timed(eng, steps, warmup, batch=1024):
    for _ in range(warmup):
        eng.collect(); eng.update(batch_size=batch, repeat=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = 0
    for _ in range(steps):
        eng.collect()
        _, n = eng.update(batch_size=batch, repeat=2)
        rows += n
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, rows / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--learners", nargs="+", default=["replicated", "dp", "dp_sharded", "tp"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS["c3"]
    base, _ = bench.build_engine(wl, 0, 1, dev, learner="replicated")
    t1, rows1 = timed(base, args.steps, 30)
    out = {"workload": wl["name"], "single_rank": {"ms_per_step": 1e3 * t1, "env_steps_per_s": rows1 / t1, "rows_per_update": rows1}, "emulated": []}
    print(json.dumps(out["single_rank"]), flush=True)
    pol, trk = base.policy_flat.clone(), base.tracker_flat.clone()
    for W in args.worlds:
        for learner in args.learners:
            coll = EmulatedPeers(W)
            eng, _ = bench.build_engine(wl, 0, W, dev, learner=learner, coll=coll,
                                        tracker_backward="sharded" if learner == "replicated" else None)
            eng.policy_flat[:pol.numel()].copy_(pol); eng.tracker_flat.copy_(trk)
            if eng.tp_learner is not None:       # the head shard of rank 0 from the trained policy; the rollout policy keeps the trained head
                for k, v in eng.tp_views.items():
                    src = eng.policy_views[k]
                    v.copy_(src[eng.tp_base:eng.tp_base + eng.tp_Il] if k.startswith("actor.last") else src)
                eng._publish_tp = lambda: None   # (its all-gather would tile rank 0's shard over the whole head)
            # the emulated reductions only carry rank 0's share, which would let the dp / tp policies drift (shorter episodes, other row counts):
            # the timing runs keep the trained policy (learning rate 0: same kernels, same launches)
            for ln in (eng.learner, eng.tp_learner):
                if ln is not None:
                    ln.cfg.lr = 0.0
            eng.tracker.lr = 0.0
            t, rows = timed(eng, args.steps, args.warmup)
            c0 = {k: v for k, v in coll.calls.items()}; b0 = {k: v for k, v in coll.bytes.items()}
            n_upd = args.steps + args.warmup
            rec = {"world": W, "learner": learner, "tracker_backward": eng.tracker_backward, "per_rank_ms_per_step": 1e3 * t,
                   "rows_in_gathered_buffer": rows, "env_steps_per_s_if_collectives_were_free": W * (rows / W) / t,
                   "collective_calls_per_update": {k: v / n_upd for k, v in c0.items()},
                   "collective_MB_per_update": {k: v / n_upd / 1e6 for k, v in b0.items()}}
            out["emulated"].append(rec)
            print(json.dumps(rec), flush=True)
            del eng
            torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "emulate_world.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
