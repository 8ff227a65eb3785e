#!/usr/bin/env python
"""Benchmark of the CIRS hot path on MI355X: one step = Collector.collect(n_episode = n_env) + policy.update(...)
(rollout of every env to the end of its episode, then the full PPO update incl. the tracker BPTT and both Adam steps).

  python bench.py --gpus 1 --steps K --warmup W                 (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  metric = simulator env-steps/s of the whole job (all ranks), with the PPO update
inside the timed region; PPO minibatch-steps/s and the rollout/update split are reported alongside.
Workloads (BASELINE.json configs): c3 (default) KuaishouEnv big_matrix-shaped 7176 x 10728, 1024 envs per GPU,
recent-N = 10 (C4 = the same with 8 ranks: 8192 envs, weak scaling); c2 = 1411 x 3327, 64 envs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cirs-codes_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    "c3": dict(U=7176, I=10728, B=1024, T=30, N=10, thr=4, tau=10.0, gamma_exposure=10.0,
               name="KuaishouEnv big_matrix-shaped synthetic 7176x10728, 1024 envs/GPU, tracker dim 32, recent-N=10, max_turn=30, PPO batch 1024 x repeat 2"),
    "c2": dict(U=1411, I=3327, B=64, T=30, N=10, thr=4, tau=10.0, gamma_exposure=10.0,
               name="KuaishouEnv small_matrix-shaped synthetic 1411x3327, 64 envs/GPU, tracker dim 32, recent-N=10, max_turn=30, PPO batch 1024 x repeat 2"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0
# measured on MI355X, see profiles/r01p_pmc_minibatch_step.md: (2*FETCH_SIZE + WRITE_SIZE) KB summed over the seven
# kernels of one 1024-row PPO minibatch step at I = 10728 (separate rocprofv3 --pmc passes)
PMC_TRAFFIC_BYTES_PER_MINIBATCH = int((2 * 35202 + 69076) * 1024)
PMC_TRAFFIC_BYTES_BWD_KERNEL = int((2 * 5284 + 30007) * 1024)   # head_bwd_fused_kernel: Wa planes / h2 in, 8 dWa + 31 dH2 partial slabs out


def build_engine(wl, rank, world, device):
    from cirs_hip.engine import CirsEngine
    from cirs_hip.env import DeviceEnvTables
    from cirs_hip.synthetic import make_tables
    tab = make_tables(wl["U"], wl["I"], seed=0, build_dist=False)
    a_env = tab.alpha_u[tab.raw_uid, 0].astype(np.float64)
    b_env = tab.beta_i[tab.raw_pid, 0].astype(np.float64)
    # table mode like the reference (df_dist_small): the I x I float64 1/Jaccard table is built on device
    dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, alpha_env=a_env, beta_env=b_env, device=device,
                         build_dist_on_device=True)
    eng = CirsEngine(dt, wl["B"], max_turn=wl["T"], num_leave_compute=wl["N"], leave_threshold=wl["thr"], tau=wl["tau"],
                     gamma_exposure=wl["gamma_exposure"], seed=2023, world_size=world, rank=rank,
                     dist_group=None)
    return eng, tab


def hip_event_kernel_time(eng, wl, reps=20):
    """HIP-event timing on the launch stream, PPO minibatch step of mb rows launched exactly as inside the timed region:
    -> (seconds per whole cirs_ppo_minibatch call, mb, {kernel name: average seconds per launch}) where the per-kernel numbers
    come from event pairs the library records around each launch of that kernel (cirs_prof_start / cirs_prof_stop): the same
    quantity as the kernel's average duration in the rocprofv3 --kernel-trace --stats summary under profiles/."""
    from cirs_hip import abi
    import ctypes as C
    ln = eng.learner
    n = ln.n_rows
    mb = min(1024, n)
    ws = ln.workspace(mb)
    idx = torch.arange(mb, dtype=torch.int32, device=eng.device)
    losses = torch.zeros(4, dtype=torch.float32, device=eng.device)
    # snapshot optimiser state so the probe does not advance training
    snap = [t.clone() for t in (ln.params, ln.adam_m, ln.adam_v)]
    lib = ln._lib

    def run(k):
        for _ in range(k):
            abi.check(lib.cirs_ppo_minibatch(C.byref(ln.cfg), ln.params.data_ptr(), ln.grads.data_ptr(), ln.adam_m.data_ptr(),
                                             ln.adam_v.data_ptr(), ln.opt_step, C.byref(ln.batch), idx.data_ptr(), mb, None,
                                             ln.n_env, losses.data_ptr(), ws.data_ptr(), ws.numel(), ln._stream()), "probe")

    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run(3)
    torch.cuda.synchronize()
    start.record()
    run(reps)
    stop.record()
    torch.cuda.synchronize()
    t_step = start.elapsed_time(stop) / reps * 1e-3
    per_kernel = {}
    for kid, name in ((1, "head_bwd_fused_kernel"), (2, "head_stats_kernel")):
        abi.check(lib.cirs_prof_start(kid, reps), "cirs_prof_start")
        run(reps)
        tot, cnt = C.c_double(0.0), C.c_int32(0)
        abi.check(lib.cirs_prof_stop(C.byref(tot), C.byref(cnt)), "cirs_prof_stop")
        per_kernel[name] = tot.value / max(cnt.value, 1)
    for t, s in zip((ln.params, ln.adam_m, ln.adam_v), snap):
        t.copy_(s)
    return t_step, mb, per_kernel


def deepfm_sweep_probe(wl, device, E=16, reps=5):
    """Secondary metric M3: the full-catalogue DeepFM sweep (compute_normed_reward) at the workload's
    U x I, synthetic weights of the shipped model's shape.  Times cirs_deepfm_sweep with HIP events on the launch stream."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import deepfmcase
    from cirs_hip.deepfm import DeviceDeepFM
    U, I = wl["U"], wl["I"]
    rng = np.random.RandomState(0)
    m = DeviceDeepFM(deepfmcase.random_weights(rng, U, I + 1, E), device=device)
    feats = rng.randint(0, 32, (I, 4)); dur = rng.uniform(2, 60, I).astype(np.float32)
    # inputs resident in HBM before the timed region (the sweep's own H2D staging of host arrays is not the metric)
    users, items = torch.arange(U, device=device), torch.arange(I, device=device)
    feats, dur = torch.as_tensor(feats).to(device, torch.int32), torch.as_tensor(dur).to(device)
    m.sweep(users, items, feats, dur, want_pred=True)  # warm-up
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        m.sweep(users, items, feats, dur, want_pred=True)
    stop.record()
    torch.cuda.synchronize()
    t = start.elapsed_time(stop) / reps * 1e-3
    pairs = float(U) * I
    executed = 2.0 * (64 * 64 + 64 + E) * pairs          # factored algorithm actually run
    algorithmic = (2.0 * ((6 * E + 1) * 64 + 64 * 64 + 64) + 18 * E) * pairs   # SURVEY §8(d) F_sweep (unfactored reference algorithm)
    return {"pairs_per_s": pairs / t, "seconds_per_sweep": t, "emb_dim": E, "users": U, "items": I,
            "roofline": {"bound": "mfma", "achieved": algorithmic / t / 1e12, "achieved_executed": executed / t / 1e12,
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": algorithmic / t / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "frac_executed": executed / t / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "note": "achieved uses SURVEY's algorithmic F_sweep (unfactored first layer); achieved_executed counts the flops the factored kernel runs"},
            "cpu_reference_pairs_per_s": 1.78e6}


def sweep_mode_probe(wl, eng, device, E=16, reps=5):
    """The north-star's catalogue-sweep formulation: per vector step every env's user is scored
    against the FULL catalogue by the DeepFM user model, one item is drawn per env from those scores (softmax sampling,
    cirs_select_items) and the env steps.  Timed with HIP events on the launch stream, B = the workload's env count."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import deepfmcase
    from cirs_hip.deepfm import DeviceDeepFM
    from cirs_hip.static_policy import select_items
    U, I, B = wl["U"], wl["I"], wl["B"]
    rng = np.random.RandomState(1)
    m = DeviceDeepFM(deepfmcase.random_weights(rng, U, I + 1, E), device=device)
    feats = torch.as_tensor(rng.randint(0, 32, (I, 4)).astype(np.int32)).to(device); dur = torch.as_tensor(rng.uniform(2, 60, I).astype(np.float32)).to(device)
    users = torch.as_tensor(rng.randint(0, U, B)); items = torch.arange(I, device=device)
    env = eng.env
    env.reset(users)
    users_d = users.to(device)

    def step(k):
        scores, _ = m.sweep(users_d, items, feats, dur, want_pred=True)
        act, _ = select_items(scores, softmax=True, seed=7, rng_step=k, skip=env.done)
        env.step(act)

    step(0)
    env.reset(users)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for k in range(reps):
        step(k + 1)
    stop.record()
    torch.cuda.synchronize()
    t = start.elapsed_time(stop) / reps * 1e-3
    a_sweep = I * (4 * E + 24) + (4 * E + 4)                       # item-side DeepFM rows streamed once per env-step
    f_sweep = I * (2.0 * ((6 * E + 1) * 64 + 64 * 64 + 64) + 18 * E)
    steps_per_s = B / t
    return {"env_steps_per_s": steps_per_s, "seconds_per_vector_step": t, "envs": B, "emb_dim": E,
            "logical_hbm": {"bytes_per_env_step": a_sweep, "achieved": steps_per_s * a_sweep / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": steps_per_s * a_sweep / 1e9 / HBM_PEAK_GBS,
                            "note": "logical rate: item rows staged once per workgroup serve every env of the tile, so it is not the physical HBM rate"},
            "mfma": {"flop_per_env_step": f_sweep, "achieved": steps_per_s * f_sweep / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": steps_per_s * f_sweep / 1e12 / PEAK_FP32_MFMA_TFLOPS}}


def cpu_baseline(wl, budget_envs=1024, threads=None):
    """The oracle port of the same step on the host cores, bounded sample (fewer envs, same tables / episode rule).
    Threads are capped: the per-step tensors are tiny and oversubscribing a 256-core host makes the port slower."""
    threads = threads or min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_path
    import policycase
    import rolloutcase
    from cirs_hip.synthetic import make_tables
    tab = make_tables(wl["U"], wl["I"], seed=0, build_dist=False)
    tp = rolloutcase.tracker_param_dict(wl["U"], wl["I"], wl["T"], seed=2, emb_scale=0.01)
    arrs = policycase.random_weights(np.random.RandomState(2), wl["I"])
    B = min(budget_envs, wl["B"])
    torch.set_num_threads(threads)
    cpu_path.run_cpu_step(tab, tp, arrs, 4, wl["T"], N=wl["N"], thr=wl["thr"], do_update=False)  # warm-up
    t0 = time.perf_counter()
    r = cpu_path.run_cpu_step(tab, tp, arrs, B, wl["T"], N=wl["N"], thr=wl["thr"], tau=wl["tau"],
                              gamma_exposure=wl["gamma_exposure"], batch_size=1024, repeat=2)
    dt = time.perf_counter() - t0
    return {"value": r["env_steps"] / dt, "unit": "env-steps/s", "cores": threads, "host_cores": os.cpu_count() or 1, "kind": "port",
            "sample": f"1 step (collect + update) with {B} envs on the same {wl['U']}x{wl['I']} tables: {r['env_steps']} env-steps, "
                      f"{r['minibatches']} PPO minibatch steps, collect {r['t_collect']:.2f}s + update {r['t_update']:.2f}s "
                      "(C oracle env/actor via OpenMP, torch-fp32 tracker/PPO restatement)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)  # RCCL over xGMI

    eng, tab = build_engine(wl, rank, world, device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        eng.collect()
        losses, n = eng.update(batch_size=1024, repeat=2)
        return losses.shape[0]

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    steps_local = 0
    mb_steps = 0
    t_collect = 0.0
    for _ in range(args.steps):
        eng.collect()
        losses, n = eng.update(batch_size=1024, repeat=2)
        steps_local += n   # rows of the update = env-steps of this collect over ALL ranks (update() reads the lengths back once)
        mb_steps += losses.shape[0]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    total_steps = float(steps_local)

    # split (untimed extra pass, same state): rollout-only and update-only rates
    barrier()
    ta = time.perf_counter(); eng.collect(); n_ro = int(eng.lengths.sum()); torch.cuda.synchronize(); tb = time.perf_counter()
    l2, n2 = eng.update(1024, 2); torch.cuda.synchronize(); tc = time.perf_counter()

    if rank == 0:
        t_mb, mb, t_k = hip_event_kernel_time(eng, wl)
        I = wl["I"]
        S, H = 20, 64
        # Dominant kernel of the timed step (profiles/*_kernel_stats.csv): head_bwd_fused_kernel, the fused actor-head backward
        # of a PPO minibatch step.  ALGORITHMIC flop per launch: the two backward products of the head layer,
        # dWa = dZ^T H2 and dH2 = dZ Wa: 2 x 2*mb*I*64 ("fwd + 2 x bwd" of 2*mb*64*I each).  EXECUTED: + one
        # recompute of the logits tile (2*mb*I*64) instead of reading a B x I probability matrix from HBM.
        t_bwd = t_k["head_bwd_fused_kernel"]
        flop_bwd = 4.0 * mb * I * H
        exec_bwd = 6.0 * mb * I * H
        # whole minibatch step (7 launches), SURVEY 8(d): 3 x 2*mb*(S*64 + 64*64 + 64*I) = 4.25 GFLOP at mb = 1024, I = 10728
        flop_step = 6.0 * mb * (S * H + H * H + H * I)
        exec_step = 8.0 * mb * I * H + 6.0 * mb * (S * H + H * H)
        out = {
            "metric": "simulator env-steps/s (collect + PPO update in the timed region), KuaishouEnv",
            "value": total_steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (policy/tracker; rollout on the fp32 MFMA, PPO head products fp32-accurate from 3 bf16 pieces per operand on the bf16 MFMA, fp32 accumulate) + f64 (env rewards, GAE)", "data": "synthetic",
            "config": {"workload": wl["name"], "envs_total": wl["B"] * world,
                       "parallelism": (f"env-sharded x{world}; one all-gather of trajectory records per update; data-parallel learner: "
                                       f"global minibatch = 1024 x {world} rows sharded by rows, one flat-gradient all-reduce per minibatch")
                       if world > 1 else "single GPU"},
            "ppo_minibatch_steps_per_s": mb_steps / elapsed,
            "rollout_only_env_steps_per_s": n_ro / (tb - ta),
            "update_only_ms": 1e3 * (tc - tb), "update_minibatch_steps": int(l2.shape[0]),
            "roofline": {"bound": "mfma", "kernel": "head_bwd_fused_kernel (PPO minibatch step: fused actor-head backward; fp32 products as 3 bf16 pieces per operand on v_mfma_f32_32x32x16_bf16, fp32 accumulate)",
                         "achieved": flop_bwd / t_bwd / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop_bwd / t_bwd / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "achieved_executed": exec_bwd / t_bwd / 1e12, "frac_executed": exec_bwd / t_bwd / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "algorithmic_flop_per_launch": flop_bwd, "traffic": None, "seconds_per_launch": t_bwd, "rows": mb,
                         "timing": "HIP events recorded by the library around each launch of this kernel on its launch stream (cirs_prof_start/stop)",
                         "peak_note": "fp32 MFMA dense peak: the kernel's results are fp32-accurate (DESIGN.md section 4); its 6x expanded bf16 flops are exec_bf16_flop_per_launch",
                         "exec_bf16_flop_per_launch": 6.0 * exec_bwd},
            "minibatch_step": {"seconds": t_mb, "launches": 7, "algorithmic_flop": flop_step, "achieved": flop_step / t_mb / 1e12,
                               "achieved_executed": exec_step / t_mb / 1e12, "frac": flop_step / t_mb / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "traffic": None,
                               "head_stats_kernel_seconds": t_k["head_stats_kernel"],
                               "note": "one whole cirs_ppo_minibatch call: head_stats_kernel + head_bwd_fused_kernel + 5 small kernels"},
        }
        # HBM traffic from the committed PMC passes (profiles/r01p_pmc_minibatch_step.md: FETCH_SIZE doubled per the gfx950
        # note, WRITE_SIZE as reported): the fused backward kernel alone, and the whole minibatch step (all eight kernels)
        out["roofline"]["traffic"] = PMC_TRAFFIC_BYTES_BWD_KERNEL if args.workload == "c3" else None
        out["minibatch_step"]["traffic"] = PMC_TRAFFIC_BYTES_PER_MINIBATCH if args.workload == "c3" else None
        out["roofline"]["traffic_source"] = "profiles/r01p_pmc_minibatch_step.md (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, C3)"
        if world == 1:  # secondary probes and the host baseline belong to the single-GPU run (task contract: rank 0 at N=1 only)
            out["deepfm_sweep"] = deepfm_sweep_probe(wl, device)
            out["sweep_mode"] = sweep_mode_probe(wl, eng, device)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
