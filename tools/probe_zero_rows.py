"""which fraction of a PPO minibatch's rows has a policy-gradient coefficient of EXACTLY zero (ratio clipped on the
side where the clip is active, or the probability clamped; ent_coef = 0) in the regime bench.py trains into?  Those rows add nothing to
dWa / dH2, so the head backward could leave them out.

Runs the C3 engine for `--steps` collect + update rounds, then one more update minibatch by minibatch with CIRS_PPO_MERGE_KERNEL=1 (the
merge launch stores c_logp of every row in the minibatch workspace) and reads the coefficients back after every optimiser step.
Prints one JSON line: the zero fraction per minibatch step of both repeats + the mean."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--dropout", type=float, default=0.0)
    args = ap.parse_args()
    import bench
    from cirs_hip import abi
    from cirs_hip.learner import minibatch_slices
    wl = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    eng, _ = bench.build_engine(wl, 0, 1, dev, dropout=args.dropout)
    out = {"workload": wl["name"], "steps_before": args.steps, "probes": []}
    for probe_at in (args.steps, 4 * args.steps):
        while eng.collect_count < probe_at:
            eng.collect(); eng.update(1024, 2)
        eng.collect()
        # process_fn exactly as update() does, then learn() by hand
        traj, x_hist, lens_d, users = eng._gather()
        ln = eng.learner
        lens = lens_d.cpu().numpy().astype(np.int32)
        ln.prepare(traj, lens, lens_dev=lens_d)
        n = ln.n_rows
        slices = minibatch_slices(n, 1024)
        ws = ln.workspace(max(e - s for s, e in slices))
        perm = ln._perms_on_device(n, 2, None)
        losses = torch.zeros((2 * len(slices), 4), device=dev)
        os.environ["CIRS_PPO_MERGE_KERNEL"] = "1"
        fr, k = [], 0
        for rep in range(2):
            for s0, e0 in slices:
                mb = e0 - s0
                n_pad = (mb + 31) // 32 * 32
                abi.check(ln._lib.cirs_ppo_minibatch(C.byref(ln.cfg), ln.params.data_ptr(), ln.grads.data_ptr(), ln.adam_m.data_ptr(), ln.adam_v.data_ptr(),
                                                     ln.opt_step, C.byref(ln.batch), perm[rep].data_ptr() + 4 * s0, mb, None, ln.n_env,
                                                     losses.data_ptr() + 16 * k, ws.data_ptr(), ws.numel(), ln._stream()), "mb")
                ln.opt_step += 1; k += 1
                torch.cuda.synchronize()
                # carve() of csrc/ppo.hip: obs[n_pad*S] adv ret v_s logp_old act h1[n_pad*64] h2[n_pad*64] value lse ez za c_logp ...
                off = n_pad * (ln.S + 4 + 1 + 128 + 1 + 3)
                c = ws.view(torch.float32)[off:off + mb]
                fr.append(float((c == 0).float().mean()))
        os.environ.pop("CIRS_PPO_MERGE_KERNEL")
        eng.tracker.backward(users, traj, ln.b_env, ln.b_t, ln.offsets_dev, ln.lens_dev, n, ln.dobs, x_hist=None)
        eng.tracker.adam_update()
        h = len(slices)
        out["probes"].append({"after_updates": probe_at, "rows": n, "mean_episode_len": n / wl["B"], "minibatch_steps": 2 * h,
                              "zero_frac_repeat0": [round(x, 3) for x in fr[:h]], "zero_frac_repeat1": [round(x, 3) for x in fr[h:]],
                              "mean_repeat0": float(np.mean(fr[:h])), "mean_repeat1": float(np.mean(fr[h:])), "mean": float(np.mean(fr))})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
