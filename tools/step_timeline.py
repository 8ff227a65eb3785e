"""Phase timeline of the timed step from a rocprofv3 kernel trace (rocpd sqlite .db): python tools/step_timeline.py <db> [md_out]

Splits every steady-state step (collect + update) of `bench.py` into phases by kernel name and prints, per phase, the wall
span on the GPU, the kernel busy time inside it, the idle gaps between consecutive kernels and the launch count — averaged
over the steps of the trace (the first `SKIP` steps are dropped as warm-up).

phases:  rollout      first tracker_step_kernel of a collect  -> last rollout kernel
         prepare      end of rollout -> first trunk_adv_kernel (GAE, returns, permutations, tracker forward over the buffer)
         minibatches  first trunk_adv_kernel -> last adam_next_kernel / adam2_kernel of the update
         tracker_bwd  after the last minibatch Adam -> next collect's first kernel (BPTT through the tracker, its Adam)
"""
import sqlite3
import sys
from collections import defaultdict

SKIP = 10


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("cirs::", "")
    return n[:48]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end from kernels order by start"))
    ks = [(short(n), s, e) for n, s, e in rows]
    is_roll = lambda n: n.startswith("tracker_step_kernel") or n.startswith("actor_mass_kernel") or n.startswith("actor_mass_small_kernel") or n.startswith("actor_pick") \
        or n.startswith("actor_head_kernel") or n.startswith("actor_merge")
    # step boundaries: a rollout kernel whose predecessor is not a rollout kernel
    starts = [i for i, k in enumerate(ks) if is_roll(k[0]) and (i == 0 or not is_roll(ks[i - 1][0]))]
    steps = []
    for a, b in zip(starts[:-1], starts[1:]):
        seg = ks[a:b]
        names = [k[0] for k in seg]
        if not any(n.startswith("trunk_adv_kernel") for n in names):
            continue    # a collect without an update (probe legs)
        steps.append(seg)
    steps = steps[SKIP:]
    if not steps:
        print("no steady-state steps found"); return
    agg = defaultdict(lambda: defaultdict(float))
    kern = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    gaps = defaultdict(lambda: [0, 0.0])
    for seg in steps:
        names = [k[0] for k in seg]
        r_end = max(i for i, n in enumerate(names) if is_roll(n))
        mb0 = min(i for i, n in enumerate(names) if n.startswith("trunk_adv_kernel"))
        is_adam = lambda n: n.startswith("adam2_kernel") or n.startswith("adam_next_kernel")
        mb1 = max(i for i, n in enumerate(names) if is_adam(n) and i > mb0)
        bounds = {"rollout": (0, r_end + 1), "prepare": (r_end + 1, mb0), "minibatches": (mb0, mb1 + 1), "tracker_bwd": (mb1 + 1, len(seg))}
        t_next = seg[-1][2]
        for ph, (a, b) in bounds.items():
            if b <= a:
                continue
            part = seg[a:b]
            t0 = seg[a - 1][2] if a > 0 else part[0][1]
            t1 = part[-1][2]
            busy = sum(e - s for _, s, e in part)
            agg[ph]["span"] += t1 - t0
            agg[ph]["busy"] += busy
            agg[ph]["launches"] += len(part)
            for n, s, e in part:
                kern[ph][n][0] += 1; kern[ph][n][1] += e - s
        for i in range(1, len(seg)):       # the largest gaps between consecutive kernels, keyed by (previous kernel -> next kernel)
            gap = seg[i][1] - seg[i - 1][2]
            if gap > 3000:
                gaps[(seg[i - 1][0], seg[i][0])][0] += 1; gaps[(seg[i - 1][0], seg[i][0])][1] += gap
        agg["step"]["span"] += seg[-1][2] - seg[0][1]
        agg["step"]["busy"] += sum(e - s for _, s, e in seg)
        agg["step"]["launches"] += len(seg)
    ns = len(steps)
    out = [f"steady-state steps in the trace: {ns} (first {SKIP} dropped)", "",
           "| phase | GPU span ms | kernel busy ms | idle ms | launches |", "|---|---|---|---|---|"]
    for ph in ["rollout", "prepare", "minibatches", "tracker_bwd", "step"]:
        a = agg[ph]
        out.append(f"| {ph} | {a['span']/ns/1e6:.3f} | {a['busy']/ns/1e6:.3f} | {(a['span']-a['busy'])/ns/1e6:.3f} | {a['launches']/ns:.0f} |")
    for ph in ["prepare", "tracker_bwd"]:
        out += ["", f"kernels of `{ph}` (per step):", "", "| kernel | launches | busy us |", "|---|---|---|"]
        for n, (c, t) in sorted(kern[ph].items(), key=lambda kv: -kv[1][1])[:14]:
            out.append(f"| {n} | {c/ns:.1f} | {t/ns/1e3:.1f} |")
    out += ["", "gaps > 3 us between consecutive kernels (per step):", "", "| previous kernel -> next kernel | count | idle us |", "|---|---|---|"]
    for (pa, pb), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
        out.append(f"| {pa} -> {pb} | {c/ns:.2f} | {t/ns/1e3:.1f} |")
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
