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
import gc
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
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (no sparsity)
HBM_PEAK_GBS = 8000.0
PMC_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")
MINIBATCH_KERNELS = ("head_stats_kernel", "head_bwd_fused_kernel", "trunk_rows_kernel", "adam_next_kernel")
# the 4 launches of a minibatch step inside cirs_ppo_learn's loop (round 5; rounds 3-4: 7).  The head of step k + 1 -- trunk forward, advantage
# statistics, Wa planes -- runs in step k's optimiser launch; the chunk-slab sums of d h2, the trunk backward, the weight-gradient sums and the
# squared-norm partials are one launch; trunk_adv_kernel runs once per update


def kernel_source_hash():
    """Fingerprint of every kernel source: PMC numbers recorded for other sources are stale and are NOT reported."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "cirs-codes_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "cirs-codes_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(workload):
    """HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    runs, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md) -- only if they were taken on THIS kernel source and workload;
    otherwise None (never a stale constant).  -> ({kernel: bytes per launch}, source string) or (None, reason)."""
    if not os.path.exists(PMC_TRAFFIC_JSON):
        return None, "profiles/pmc_traffic.json absent: run tools/pmc_traffic.py on the GPU box"
    z = json.load(open(PMC_TRAFFIC_JSON))
    if z.get("source_hash") != kernel_source_hash():
        return None, f"profiles/pmc_traffic.json is stale (taken on kernel sources {z.get('source_hash')}, this build is {kernel_source_hash()})"
    if z.get("workload") != workload:
        return None, f"profiles/pmc_traffic.json was taken on workload {z.get('workload')}"
    out = {}
    for k, v in z["kernels"].items():      # template instantiations ("head_bwd_fused_kernel<false, true>") are looked up by the kernel's name
        out[k] = int(v["bytes_per_launch"])
        out.setdefault(k.split("<")[0], int(v["bytes_per_launch"]))
    return out, f"profiles/pmc_traffic.json ({z.get('taken', '?')}; 2 x FETCH_SIZE + WRITE_SIZE, separate passes)"


def pmc_json(workload):
    """The committed PMC summary (profiles/pmc_traffic.json) if it was taken on THIS kernel source and workload, else None."""
    if not os.path.exists(PMC_TRAFFIC_JSON):
        return None
    z = json.load(open(PMC_TRAFFIC_JSON))
    return z if z.get("source_hash") == kernel_source_hash() and z.get("workload") == workload else None


def build_engine(wl, rank, world, device, learner="dp", dropout=0.0, tracker_backward=None, coll=None, dropout_redraw=False):
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
                     dist_group=None, learner_mode=learner, dropout=dropout, tracker_backward=tracker_backward, coll=coll,
                     dropout_redraw=dropout_redraw)
    return eng, tab


def timed_pass(wl, device, dropout, warmup, steps, batch=1024, pretrain=0):
    """One more workload through the protocol of the headline (fresh engine, `warmup` untimed steps, `steps` timed steps = collect + update
    between synchronisations, single GPU) -> the numbers the driver's one default run would otherwise never witness: the C3 step with
    the tracker in training mode (Dropout(0.1) live in rollout and BPTT, the way the reference trains) and BASELINE
    configs[1] (C2: 1411 x 3327, 64 envs)."""
    eng, _ = build_engine(wl, 0, 1, device, dropout=dropout)
    for _ in range(pretrain + warmup):
        eng.collect(); eng.update(batch_size=batch, repeat=2)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]      # one event per step on the launch stream: the spread of the steps
    gc.collect()
    gc.disable()          # (see the headline's timed region)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    n_steps = mb_steps = 0
    for k in range(steps):
        eng.collect()
        losses, n = eng.update(batch_size=batch, repeat=2)
        n_steps += n; mb_steps += losses.shape[0]
        marks[k + 1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(steps))
    out = {"workload": wl["name"], "tracker_dropout": dropout, "value": n_steps / dt, "unit": "env-steps/s", "ms_per_step": 1e3 * dt / steps,
           "steps": steps, "warmup": warmup, "envs": wl["B"], "mean_episode_len": n_steps / steps / wl["B"],
           "minibatch_steps_per_update": mb_steps / steps,
           "gpu_ms_per_step_median_max": [per_step[len(per_step) // 2], per_step[-1]]}
    del eng
    torch.cuda.empty_cache()
    return out


def hip_event_kernel_time(eng, wl, reps=100):
    """HIP-event timing on the launch stream, PPO minibatch steps of mb rows launched exactly as inside the timed region (cirs_ppo_learn's loop):
    -> (seconds per minibatch step, mb, {kernel name: average seconds per launch}) where the per-kernel numbers
    come from event pairs the library records around each launch of that kernel (cirs_prof_start / cirs_prof_stop): the same
    quantity as the kernel's average duration in the rocprofv3 --kernel-trace --stats summary under profiles/."""
    from cirs_hip import abi
    import ctypes as C
    ln = eng.learner
    if not hasattr(ln, "n_rows"):      # learner "tp": the updates ran on the item-sharded learner; the probe times the full-catalogue kernels
        traj, lens, lens_d = eng._last_prepared
        ln.prepare(traj, lens, lens_dev=lens_d)
    n = ln.n_rows
    mb = min(1024, n)
    n_use = n // mb * mb                    # whole minibatches of mb rows: every step of the probe is a step of the timed region's size
    spc = n_use // mb                       # steps per cirs_ppo_learn call
    ws = ln.workspace(2 * mb)
    perm = torch.arange(n_use, dtype=torch.int32, device=eng.device)
    losses = torch.zeros((spc, 4), dtype=torch.float32, device=eng.device)
    # snapshot optimiser state so the probe does not advance training
    snap = [t.clone() for t in (ln.params, ln.adam_m, ln.adam_v)]
    lib = ln._lib
    reps = max(1, reps // spc)              # calls; reps * spc steps

    def run(k):
        for _ in range(k):
            abi.check(lib.cirs_ppo_learn(C.byref(ln.cfg), ln.params.data_ptr(), ln.grads.data_ptr(), ln.adam_m.data_ptr(), ln.adam_v.data_ptr(),
                                         ln.opt_step, C.byref(ln.batch), perm.data_ptr(), n_use, mb, 1, None, 0, ln.n_env, losses.data_ptr(),
                                         ws.data_ptr(), ws.numel(), ln._stream()), "probe")

    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run(3)
    torch.cuda.synchronize()
    start.record()
    run(reps)
    stop.record()
    torch.cuda.synchronize()
    t_step = start.elapsed_time(stop) / (reps * spc) * 1e-3
    per_kernel = {}
    for kid, name in ((1, "head_bwd_fused_kernel"), (2, "head_stats_kernel")):
        run(1)    # the event pairs are taken in steady state, like the kernel's average in a rocprofv3 trace of the timed loop
        abi.check(lib.cirs_prof_start(kid, reps * spc), "cirs_prof_start")
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
    # the 64x64 layer runs as fp32 products from 3 bf16 pieces per operand = 6 bf16 MFMAs per fp32 product: the pipe the kernel
    # occupies is the bf16 one, so ITS flops (6 x the 64x64 layer + the fp32 VALU rest) are priced against the bf16 peak
    bf16_pipe = 6.0 * 2.0 * 64 * 64 * pairs
    return {"pairs_per_s": pairs / t, "seconds_per_sweep": t, "emb_dim": E, "users": U, "items": I,
            "roofline": {"bound": "mfma", "achieved": bf16_pipe / t / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": bf16_pipe / t / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                         "note": "executed bf16-pipe flops (6 bf16 MFMAs per fp32 product of the 64x64 layer) vs the dense bf16 MFMA peak"},
            "logical_fp32": {"executed_tflops": executed / t / 1e12, "survey_f_sweep_tflops": algorithmic / t / 1e12,
                             "fp32_mfma_peak": PEAK_FP32_MFMA_TFLOPS,
                             "note": "fp32-equivalent rates, NOT roofline fractions: 'executed' = flops of the factored algorithm (first layer split per user / per item), "
                                     "'survey_f_sweep' = SURVEY 8(d)'s unfactored F_sweep; both can exceed the fp32 MFMA peak because the work runs on the bf16 pipe"},
            "cpu_reference_pairs_per_s": 1.78e6}


def sweep_mode_probe(wl, eng, device, E=32, reps=5):
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
            "mfma": {"bf16_pipe_flop_per_env_step": 6.0 * 2.0 * 64 * 64 * I, "achieved": steps_per_s * 6.0 * 2.0 * 64 * 64 * I / 1e12,
                     "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": steps_per_s * 6.0 * 2.0 * 64 * 64 * I / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                     "survey_f_sweep_flop_per_env_step": f_sweep, "survey_f_sweep_tflops": steps_per_s * f_sweep / 1e12,
                     "note": "frac prices the executed bf16-pipe flops against the dense bf16 peak; survey_f_sweep_tflops is the logical fp32 rate of SURVEY's unfactored count (not a roofline fraction)"}}


def gather_fm_probe(device, reps=10):
    """K1-K2 micro-benchmark (the stage the north-star's ">= 40 % of the HBM-read roofline" applies to): embedding
    gather + linear + FM bi-interaction on random (user, item) pairs, no DNN (cirs_gather_fm).  ALGORITHMIC bytes per pair =
    28 (X row) + 2 x (4E + 4) (user and item embedding rows + their linear weights) + 4 (out) = 8E + 40; the 32 x E feat table
    is LDS-resident and not counted.  Timed with HIP events on the launch stream, inputs resident in HBM.  Cases: the
    KuaishouEnv big_matrix shape (tables of 0.9 + 1.4 MB at E = 32: they live in the 4 MiB L2 of every XCD, so the physical
    HBM traffic is the X / out streams only and `achieved` is a LOGICAL rate) and the C5 shape (2^20 x 2^20, E = 64: 268 MB per
    table, past L2 and Infinity Cache, where algorithmic ~ physical).  Physical FETCH/WRITE numbers: profiles/pmc_traffic.json."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import deepfmcase
    from cirs_hip.deepfm import DeviceDeepFM
    out = []
    traffic, src = pmc_traffic("c3")
    pz = pmc_json("c3") or {}
    cases = pz.get("gather_fm_cases") or {}
    # mid_E64: 2 x 2^17 rows x 260 B = 68 MB -- past the 8 x 4 MiB L2s, inside the 256 MiB Infinity Cache: where the logical rate of the
    # C3 shape turns into the physical rate of the C5 shape
    for name, U, I, E, n in (("c3_E32", 7176, 10728, 32, 1 << 24), ("c3_E16", 7176, 10728, 16, 1 << 24), ("mid_E64", 1 << 17, 1 << 17, 64, 1 << 23),
                             ("c5_E64", 1 << 20, 1 << 20, 64, 1 << 23)):
        g = torch.Generator(device="cpu").manual_seed(E)
        # weights on the device directly (the C5 tables are 2 x 268 MB)
        w = {"emb_user": torch.randn(U, E, generator=g) * 0.3, "emb_item": torch.randn(I + 1, E, generator=g) * 0.3,
             "emb_feat": torch.randn(32, E, generator=g) * 0.3, "lin_user": torch.randn(U, generator=g) * 0.1,
             "lin_item": torch.randn(I + 1, generator=g) * 0.1, "lin_feat": torch.randn(32, generator=g) * 0.1, "lin_dense": torch.randn(1, generator=g) * 0.01,
             "w1": torch.zeros(64, 6 * E + 1), "b1": torch.zeros(64), "w2": torch.zeros(64, 64), "b2": torch.zeros(64), "last": torch.zeros(64),
             "out_bias": torch.zeros(1)}
        m = DeviceDeepFM(w, device=device)
        X = torch.empty((n, 7), dtype=torch.float32, device=device)
        gd = torch.Generator(device=device).manual_seed(E + 1)
        X[:, 0] = torch.randint(0, U, (n,), device=device, generator=gd).float()
        X[:, 1] = torch.randint(0, I, (n,), device=device, generator=gd).float()
        X[:, 2:6] = torch.randint(0, 32, (n, 4), device=device, generator=gd).float()
        X[:, 6] = torch.rand(n, device=device, generator=gd) * 58 + 2
        m.gather_fm(X)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        start.record()
        for _ in range(reps):
            y = m.gather_fm(X)
        stop.record()
        torch.cuda.synchronize()
        t = start.elapsed_time(stop) / reps * 1e-3
        alg = (8 * E + 40) * n
        rec = {"case": name, "users": U, "items": I, "emb_dim": E, "pairs": n, "seconds_per_launch": t, "pairs_per_s": n / t,
               "algorithmic_bytes_per_pair": 8 * E + 40,
               "roofline": {"bound": "hbm", "achieved": alg / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / t / 1e9 / HBM_PEAK_GBS,
                            # per CASE (dispatch group of the PMC workload), FETCH_SIZE doubled per the guide's streaming correction
                            "traffic": (cases.get(name) or {}).get("bytes_doubled_fetch", (traffic or {}).get(f"gather_fm_kernel<{E}, true>")),
                            # the same counters with the factors of the known-bytes calibration launches (random rows of this width + streams)
                            "traffic_calibrated": (cases.get(name) or {}).get("bytes_calibrated"),
                            # requests at the L2's memory-side port per second: Infinity-Cache hits are counted too (MI355X_MICROARCH.md, HBM section),
                            # so this is an UPPER bound of the HBM rate and may exceed what HBM alone delivers (6.3 TB/s streaming) for
                            # tables of which a part stays in the 256 MiB Infinity Cache
                            "pmc_fabric_GBps": ((cases.get(name) or {}).get("bytes_calibrated") or 0) / t / 1e9 or None,
                            "calibration": (cases.get(name) or {}).get("calibration"),
                            "compulsory_stream_bytes_per_pair": 32,
                            "note": "tables L2-resident: achieved is a logical gather rate, physical HBM traffic ~ 32 B/pair" if U < 100000 else
                                    ("tables past L2, inside the Infinity Cache (MALL): rows come from MALL, HBM sees the X / out streams" if U < (1 << 19) else
                                     "tables past L2 / Infinity Cache: algorithmic ~ physical")}}
        out.append(rec)
        del m, X, w, y
        torch.cuda.empty_cache()
    return out


def c5_split_probe(device, B=1024, T=10, E=64, reps=3):
    """BASELINE configs[4] shape on ONE rank through the SPLIT code path (cirs_hip.sharded: row-sharded table lookups, shard partials +
    merge of the sampler, online DeepFM reward, compact-table tracker BPTT, tensor-parallel head learner, gradient rows pushed to
    their owners) with identity collectives: U = I = 2^20, emb_dim 64, hashed-id sized tables.  What a rank of the 8-GPU job executes,
    minus the wire.  -> env-steps/s of collect, ms per update."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import deepfmcase
    import policycase
    import rolloutcase
    from cirs_hip.deepfm import DeviceDeepFM
    from cirs_hip.env import DeviceEnv, DeviceEnvTables
    from cirs_hip.learner import flat_policy_params
    from cirs_hip.rollout import Trajectory
    from cirs_hip.sharded import LocalComm, ShardedRollout, ShardedTable, ShardedTrainer
    from cirs_hip.tracker import DeviceTracker, flat_tracker_params, tracker_param_shapes
    U = I = 1 << 20
    rng = np.random.RandomState(0)
    cats = np.where(np.arange(4)[None, :] < rng.randint(1, 5, I)[:, None], rng.randint(0, 31, (I, 4)), -1).astype(np.int32)
    feats = np.where(cats >= 0, cats + 1, 0).astype(np.int32)
    dur = rng.uniform(2, 60, I).astype(np.float32)
    wfm = deepfmcase.random_weights(rng, U, I, E)
    tp = rolloutcase.tracker_param_dict(U, I, T, 4)
    arrs = policycase.random_weights(rng, I)
    names = rolloutcase.POLICY_NAMES
    comm = LocalComm()
    env = DeviceEnv(DeviceEnvTables(None, None, cats, n_users=U, n_items=I, device=device), B, num_leave_compute=10, leave_threshold=4, max_turn=T, tau=10.0,
                    gamma_exposure=10.0, dist_mode=1)
    init = {k: (v.float() if not k.startswith("embedding_dict") else torch.zeros(4, 32)) for k, v in tp.items()}
    tflat, tviews = flat_tracker_params(tracker_param_shapes(4, 4, 32, 20), device=device, init=init)
    tparams = dict(tviews); tparams["pos_encoder.pe"] = tp["pos_encoder.pe"].float().to(device).contiguous()
    trk = DeviceTracker(tparams, B, B, B, T, device=device)
    trk.enable_training(tflat, lr=1e-3)
    wl = {k: (v if k not in ("emb_user", "emb_item", "lin_user", "lin_item") else np.zeros((4,) + v.shape[1:], np.float32)) for k, v in wfm.items()}
    fm = DeviceDeepFM(wl, device=device)
    pflat, pviews = flat_policy_params(I, init={names[k]: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)) for k, v in arrs.items()}, device=device)
    shard = {k: pviews[names[k]] for k in ("w1", "b1", "w2", "b2", "wa", "ba", "wc", "bc")}
    fu = torch.zeros((U, E + 4)); fu[:, :E] = torch.as_tensor(wfm["emb_user"]); fu[:, E] = torch.as_tensor(wfm["lin_user"])
    fi = torch.zeros((I, E + 4)); fi[:, :E] = torch.as_tensor(wfm["emb_item"]); fi[:, E] = torch.as_tensor(wfm["lin_item"])
    mk = lambda full, n: ShardedTable(full.to(device).contiguous(), n, comm)  # noqa: E731
    ident = np.arange(I, dtype=np.int64)
    sr = ShardedRollout(comm, env, trk, Trajectory(B, T, 20, device), shard, 0, I, fm, mk(fu, U), mk(fi, I),
                        mk(tp["embedding_dict.feat_user.weight"].float(), U), mk(tp["embedding_dict.feat_item.weight"].float(), I), ident, ident, feats,
                        dur, (-60.0, 60.0))
    trainer = ShardedTrainer(sr, pflat, B)
    users = torch.as_tensor(rng.randint(0, U, B).astype(np.int32)).to(device)
    lens = sr.collect(users, seed=1, rng_base=0)
    trainer.update(lens, 1024, 2)
    torch.cuda.synchronize()
    t_c = t_u = 0.0
    n_steps = 0
    for k in range(reps):
        t0 = time.perf_counter()
        lens = sr.collect(users, seed=1, rng_base=(k + 1) * T)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        losses, n = trainer.update(lens, 1024, 2, perm_key=(1, 64 * k))
        torch.cuda.synchronize()
        t_c += t1 - t0; t_u += time.perf_counter() - t1; n_steps += n
    return {"users": U, "items": I, "emb_dim": E, "envs": B, "max_turn": T, "ranks": 1,
            "collect_env_steps_per_s": n_steps / t_c, "ms_per_collect": 1e3 * t_c / reps, "ms_per_update": 1e3 * t_u / reps,
            "minibatch_steps_per_update": int(losses.shape[0]),
            "note": "the split (row-sharded tables, item-sharded head, tensor-parallel learner) code path of BASELINE configs[4] with identity "
                    "collectives on one device: per-stage launches, no fused rollout; the 8-rank wire time is not in it"}


def _set_omp_threads(n):
    import ctypes
    torch.set_num_threads(n)
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))   # OMP_NUM_THREADS is only read when libgomp initialises
    except OSError:
        pass
    os.environ["OMP_NUM_THREADS"] = str(n)


def cpu_baseline(wl, eng):
    """The oracle port of the same step (oracle/cpu_path.py: C oracle env / actor, torch-fp32 tracker + PPO restatement) on the host
    cores, with the SAME tables, policy and tracker weights as the GPU leg, on a BOUNDED sample: a single-thread leg, a 16-thread
    leg and an all-core leg (the per-step tensors are small, so more threads are not always faster; the best multi-thread leg is
    `value`).  The reference's own Python numbers (SURVEY section 6 / tools/bench_reference.py, measured in the dev container:
    the reference cannot travel to the GPU box) are quoted next to it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_path
    import rolloutcase
    from cirs_hip.synthetic import make_tables
    tab = make_tables(wl["U"], wl["I"], seed=0, build_dist=False)
    tp = {k: v.detach().cpu().clone() for k, v in eng.tracker.params.items()}
    arrs = {k: eng.policy_views[name].detach().cpu().numpy().copy() for k, name in rolloutcase.POLICY_NAMES.items()}
    host = os.cpu_count() or 1
    legs = []
    # bounded samples: every leg is sized to ~10-30 s (a 256-core host does NOT make the port faster: the per-step tensors are tiny,
    # so the widest leg is capped at 64 threads)
    for threads, B in ((1, min(16, wl["B"])), (min(16, host), min(512, wl["B"])), (min(64, host), min(512, wl["B"]))):
        if any(l["cores"] == threads for l in legs):
            continue
        _set_omp_threads(threads)
        cpu_path.run_cpu_step(tab, tp, arrs, 4, wl["T"], N=wl["N"], thr=wl["thr"], do_update=False)  # warm-up
        t0 = time.perf_counter()
        r = cpu_path.run_cpu_step(tab, tp, arrs, B, wl["T"], N=wl["N"], thr=wl["thr"], tau=wl["tau"], gamma_exposure=wl["gamma_exposure"],
                                  batch_size=1024, repeat=2)
        dt = time.perf_counter() - t0
        legs.append({"cores": threads, "envs": B, "env_steps": r["env_steps"], "mean_episode_len": r["env_steps"] / B, "seconds": dt,
                     "env_steps_per_s": r["env_steps"] / dt, "rollout_env_steps_per_s": r["env_steps"] / max(r["t_collect"], 1e-9),
                     "ppo_minibatch_steps_per_s": r["minibatches"] / max(r["t_update"], 1e-9), "minibatches": r["minibatches"]})
    best = max(legs[1:] or legs, key=lambda l: l["env_steps_per_s"])
    ref = None
    ref_json = os.path.join(ROOT, "profiles", "reference_python_cpu.json")
    if os.path.exists(ref_json):
        ref = json.load(open(ref_json))
    return {"value": best["env_steps_per_s"], "unit": "env-steps/s", "cores": best["cores"], "host_cores": host, "kind": "port",
            "envs_gpu_leg": wl["B"], "envs_cpu_leg": best["envs"], "envs_cpu_single_thread_leg": legs[0]["envs"],
            "sample": f"1 step (collect + update, same tables / policy / tracker weights as the GPU leg) with {best['envs']} envs (the GPU leg runs {wl['B']}) "
                      f"on {best['cores']} of {host} host cores, {wl['U']}x{wl['I']} tables: "
                      f"{best['env_steps']} env-steps, {best['minibatches']} PPO minibatch steps in {best['seconds']:.1f}s "
                      "(C oracle env/actor via OpenMP, torch-fp32 tracker/PPO restatement); legs: "
                      + "; ".join(f"{l['cores']} threads x {l['envs']} envs = {l['env_steps_per_s']:.0f} env-steps/s" for l in legs)
                      + (" -- the widest leg is slower than the 16-thread one: the port's per-step tensors are tiny, OpenMP / torch threading does not scale past ~16 threads"
                         if len(legs) > 2 and legs[2]["env_steps_per_s"] < legs[1]["env_steps_per_s"] else ""),
            "single_thread": legs[0], "legs": legs,
            "reference_python": ref or {"note": "profiles/reference_python_cpu.json absent", "survey_section_6": {
                "c2_rollout_env_steps_per_s": 940, "c3_rollout_env_steps_per_s": 520, "c2_ppo_minibatch_steps_per_s": 1.9,
                "c3_ppo_minibatch_steps_per_s": 0.75, "deepfm_pairs_per_s": 1.78e6, "cores": 8}}}


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): re-run this command line under torch.distributed.run, one rank
    per GPU, and hand back its exit code.  Rank 0 of the child job prints the JSON line on the inherited stdout.  On a box with fewer
    than N GPUs the job only starts with CIRS_BENCH_SHARE_GPU=1 (test hook: every rank on device 0, collectives over gloo)."""
    import socket
    import subprocess
    share = os.environ.get("CIRS_BENCH_SHARE_GPU", "0") == "1"
    have = torch.cuda.device_count()
    if have < n_gpus and not share:
        raise SystemExit(f"bench.py --gpus {n_gpus}: this node exposes {have} GPU(s) (set CIRS_BENCH_SHARE_GPU=1 to run every rank on device 0 over gloo)")
    with socket.socket() as sk:      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--learner", default="replicated", choices=["dp", "dp_sharded", "replicated", "tp"],
                    help="N > 1 only: how the PPO update runs over the gathered buffer.  Every mode keeps the reference's PPO configuration "
                         "(global minibatch = --global-batch rows): dp = rows of each minibatch sharded over the ranks + one gradient all-reduce; "
                         "dp_sharded = reduce-scatter + sharded Adam + parameter all-gather; replicated = every rank runs the whole update; "
                         "tp = actor head sharded by items for the update (16 B/row all-gather + d h2 all-reduce per minibatch, head shards all-gathered once per update)")
    ap.add_argument("--tracker-backward", default="sharded", choices=["sharded", "replicated"],
                    help="N > 1, learner replicated only: tracker BPTT over the rank's own envs + one gradient all-reduce (sharded) or over all "
                         "envs on every rank (replicated: no communication, results identical to one device).  Other learners always shard it")
    ap.add_argument("--global-batch", type=int, default=1024, help="PPO minibatch size over ALL ranks (reference batch_size, CIRS-RL-kuaishou.py:89)")
    ap.add_argument("--dropout", type=float, default=0.1, help="tracker dropout probability.  Default 0.1 = the mode the reference trains in (its tracker is "
                                                               "never put in eval(), SURVEY Q7), with position-keyed masks unless --dropout-redraw; "
                                                               "0 = the eval-mode tracker of the parity fixtures (reported as config.also_measured.dropout_off)")
    ap.add_argument("--dropout-redraw", action="store_true",
                    help="with --dropout > 0: the reference's own procedure (fresh masks over the whole prefix at every build_state call, "
                         "core/state_tracker.py:170-186,243-246; cirs_hip/redraw.py) instead of position-keyed masks.  Single GPU; O(T^2) row-passes by definition")
    ap.add_argument("--scaled-batch-steps", type=int, default=-1,
                    help="N > 1: extra steps timed with the global minibatch scaled to --global-batch x N (constant optimiser steps per update; "
                         "reported under scaled_batch_variant, never as value).  -1 = min(steps, 5), 0 = skip")
    ap.add_argument("--pretrain-steps", type=int, default=40,
                    help="untimed collect + update steps BEFORE the counted warm-up, so that the timed steps run in the steady state (episodes at max_turn) whatever "
                         "--warmup is: a fresh policy plays short episodes for its first ~25 updates and the driver's 5 + 20 steps would time that transient "
                         "(reported as config.pretrain_steps; the fresh-policy number stays in config.also_measured.fresh_policy)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probes", action="store_true", help="skip the secondary probes (gather_fm / sweep / cpu baseline): contract line only")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))       # the driver's command shape: `python3 bench.py --gpus N ...` with no launcher around it
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} under a launcher needs WORLD_SIZE={args.gpus} (found {world})")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the product path)"
    # test hook (tests/test_gpu_distributed.py on a 1-GPU box): every rank on device 0, collectives through gloo -- the same
    # multi-process engine path (env sharding, packed all-gather, per-minibatch all-reduce) without a second GPU
    share_gpu = os.environ.get("CIRS_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl = RCCL over xGMI; device_id binds the communicator to this rank's GPU at creation (no lazy init on first use, no
        # ambiguity about which device a rank drives)
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    G = int(args.global_batch)
    eng, tab = build_engine(wl, rank, world, device, learner=args.learner, dropout=args.dropout,
                            tracker_backward=args.tracker_backward if args.learner == "replicated" else None,
                            dropout_redraw=bool(args.dropout_redraw and args.dropout > 0))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        eng.collect()
        losses, n = eng.update(batch_size=G, repeat=2)
        return losses.shape[0]

    for _ in range(args.pretrain_steps):     # the policy reaches its steady state (full-length episodes) before the counted warm-up
        one_step()
    for _ in range(args.warmup):
        one_step()
    # Python's cyclic collector stays out of the timed region (as timeit does): a generation-2 pass over the tables / engines built above costs
    # 3-80 ms when it happens to fire inside 20 timed steps (seen as single-step spikes: tools/probes/pass_jitter.py)
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    steps_local = 0
    mb_steps = 0
    t_collect = 0.0
    for _ in range(args.steps):
        eng.collect()
        losses, n = eng.update(batch_size=G, repeat=2)
        steps_local += n   # rows of the update = env-steps of this collect over ALL ranks (update() reads the lengths back once)
        mb_steps += losses.shape[0]
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    total_steps = float(steps_local)
    ranks_identical = None
    if world > 1:
        # every rank must hold bit-identical policy / tracker parameters after the timed updates (order-fixed reductions)
        h = torch.stack([eng.policy_flat.view(torch.int32).sum(dtype=torch.int64), eng.tracker_flat.view(torch.int32).sum(dtype=torch.int64),
                         (eng.policy_flat.view(torch.int32).to(torch.int64) * torch.arange(1, eng.policy_flat.numel() + 1, device=device) % 1000003).sum()])
        allh = [torch.zeros_like(h) for _ in range(world)]
        dist.all_gather(allh, h)
        ranks_identical = all(bool(torch.equal(allh[0], x)) for x in allh)

    # N > 1, extra: the same job with the global minibatch scaled by the number of ranks (G x N rows: the number of optimiser steps per
    # update stays that of one GPU).  A DIFFERENT PPO configuration than BASELINE's -- reported beside the headline, never as `value`.
    scaled = None
    k_scaled = (min(args.steps, 5) if args.scaled_batch_steps < 0 else args.scaled_batch_steps) if world > 1 else 0
    if k_scaled > 0:
        eng.collect(); eng.update(batch_size=G * world, repeat=2)
        barrier()
        ts = time.perf_counter()
        n_s, mb_s = 0, 0
        for _ in range(k_scaled):
            eng.collect()
            l_s, nn_s = eng.update(batch_size=G * world, repeat=2)
            n_s += nn_s; mb_s += l_s.shape[0]
        barrier()
        el_s = time.perf_counter() - ts
        t = torch.tensor([el_s], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_s = float(t)
        scaled = {"global_minibatch": G * world, "steps": k_scaled, "env_steps_per_s": n_s / el_s, "ms_per_step": 1e3 * el_s / k_scaled,
                  "minibatch_steps_per_update": mb_s / k_scaled,
                  "note": "PPO batch_size scaled by the number of ranks: NOT the reference's configuration, shown for the Amdahl comparison only"}

    # split (untimed extra pass, same state): rollout-only and update-only rates
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n_ro, reps_ro = 0, 3
    ev[0].record()
    for _ in range(reps_ro):        # rollout only: HIP events on the launch stream, no host sync / reduction inside the window
        eng.collect()
    ev[1].record()
    torch.cuda.synchronize()
    t_ro = ev[0].elapsed_time(ev[1]) * 1e-3 / reps_ro
    n_ro = int(eng.lengths.sum())    # env-steps of the LAST of the collects (same policy: the others differ by sampling noise only)
    tb = time.perf_counter()
    l2, n2 = eng.update(G, 2); torch.cuda.synchronize(); tc = time.perf_counter()

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
        # whole minibatch step (4 launches), SURVEY 8(d): 3 x 2*mb*(S*64 + 64*64 + 64*I) = 4.25 GFLOP at mb = 1024, I = 10728
        flop_step = 6.0 * mb * (S * H + H * H + H * I)
        exec_step = 8.0 * mb * I * H + 6.0 * mb * (S * H + H * H)
        out = {
            "metric": "simulator env-steps/s (collect + PPO update in the timed region), KuaishouEnv",
            "value": total_steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (policy/tracker; rollout on the fp32 MFMA, PPO head products fp32-accurate from two fp16 pieces per operand (3 f16 MFMAs per product), fp32 accumulate) + f64 (env rewards, GAE)", "data": "synthetic",
            "config": {"workload": wl["name"], "envs_total": wl["B"] * world,
                       "learner": args.learner if world > 1 else "single", "tracker_backward": eng.tracker_backward if world > 1 else "single",
                       "global_minibatch": G, "minibatch_steps_per_update": mb_steps / args.steps,
                       "rows_per_rank_per_minibatch": G if (world == 1 or args.learner in ("replicated", "tp")) else G / world,
                       "ppo_repeat": 2, "tracker_dropout": args.dropout,
                       "dropout_mode": ("off (eval-mode tracker)" if args.dropout == 0 else
                                        "exact redraw (the reference's procedure: fresh masks over the whole prefix at every build_state call)" if args.dropout_redraw
                                        else "position-keyed masks (every state has the reference's marginal distribution; a position keeps its masks for the rest of the episode)"),
                       "mean_episode_len": total_steps / args.steps / (wl["B"] * world), "env_steps_per_step": total_steps / args.steps,
                       "max_turn": wl["T"], "pretrain_steps": args.pretrain_steps,
                       "timed_region": f"{args.pretrain_steps} untimed pre-training steps (steady state: full-length episodes), then {args.warmup} warm-up + {args.steps} timed collect+update steps (gc disabled inside the timed region)",
                       "parallelism": (f"env-sharded x{world}; one all-gather of trajectory records per update; learner '{args.learner}': "
                                       + {"dp": f"global minibatch of {G} rows sharded by rows over the ranks, one flat-gradient all-reduce per minibatch",
                                          "dp_sharded": f"global minibatch of {G} rows sharded by rows, reduce-scatter + sharded Adam + parameter all-gather per minibatch",
                                          "replicated": f"every rank runs all minibatches of {G} rows on the gathered buffer (no further communication)",
                                          "tp": f"every rank runs all {G} rows of a minibatch against its 1/{world} of the catalogue (item-sharded head), "
                                                "all-gather of 16 B/row + all-reduce of the d h2 partials per minibatch, head shards all-gathered per update"}[args.learner])
                       if world > 1 else "single GPU"},
            "dropout": args.dropout, "dropout_redraw": bool(args.dropout_redraw and args.dropout > 0),
            "ppo_minibatch_steps_per_s": mb_steps / elapsed, "rank_parameters_bit_identical": ranks_identical,
            "rollout_only_env_steps_per_s": n_ro / t_ro, "rollout_only_ms_per_collect": 1e3 * t_ro,
            "update_only_ms": 1e3 * (tc - tb), "update_minibatch_steps": int(l2.shape[0]),
            "roofline": {"bound": "mfma", "kernel": "head_bwd_fused_kernel (PPO minibatch step: fused actor-head backward; fp32 products from two fp16 pieces per operand = 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulate)",
                         "achieved": flop_bwd / t_bwd / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop_bwd / t_bwd / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "achieved_executed": exec_bwd / t_bwd / 1e12, "frac_executed": exec_bwd / t_bwd / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "algorithmic_flop_per_launch": flop_bwd, "traffic": None, "seconds_per_launch": t_bwd, "rows": mb,
                         "timing": "HIP events recorded by the library around each launch of this kernel on its launch stream (cirs_prof_start/stop)",
                         "kernel_note": "since round 3 the kernel's prologue also merges the head-statistics partials of its rows and forms their loss terms / backward coefficients (~2.5 us that replace a 6.5 us launch): its duration includes that work, the algorithmic flop count does not",
                         "peak_note": "fp32 MFMA dense peak: the kernel's results are fp32-accurate (DESIGN.md section 4); its 3x expanded f16 flops are exec_f16_flop_per_launch",
                         "exec_f16_flop_per_launch": 3.0 * exec_bwd},
            "minibatch_step": {"seconds": t_mb, "launches": len(MINIBATCH_KERNELS), "algorithmic_flop": flop_step, "achieved": flop_step / t_mb / 1e12,
                               "achieved_executed": exec_step / t_mb / 1e12, "frac": flop_step / t_mb / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "traffic": None,
                               "head_stats_kernel_seconds": t_k["head_stats_kernel"],
                               "note": "one step of cirs_ppo_learn's loop: head_stats_kernel + head_bwd_fused_kernel (whose prologue also merges the statistics partials and forms the row loss terms) + trunk_rows_kernel (chunk-slab sums of d h2, trunk backward, dWa slab sums, trunk-gradient sums, squared-norm partials) + adam_next_kernel (Adam + the trunk forward / advantage statistics / Wa planes of the NEXT step)"},
        }
        # HBM traffic per launch from the committed PMC passes -- only when they were taken on exactly these kernel sources
        traffic, src = pmc_traffic(args.workload)
        out["roofline"]["traffic"] = (traffic or {}).get("head_bwd_fused_kernel")
        out["roofline"]["traffic_source"] = src
        out["minibatch_step"]["traffic"] = sum(traffic.get(k, 0) for k in MINIBATCH_KERNELS) if traffic and all(k in traffic for k in MINIBATCH_KERNELS) else None
        if traffic:
            out["hbm_traffic_per_launch"] = {k: v for k, v in traffic.items() if k.split("<")[0] in ("actor_head_kernel", "actor_mass_kernel", "tracker_step_kernel", "sweep_kernel", "gather_fm_kernel")}
        if scaled is not None:
            out["scaled_batch_variant"] = scaled
        if world > 1:
            out["collectives_per_rank"] = {"calls": dict(eng.coll.calls), "bytes": dict(eng.coll.bytes),
                                           "note": "totals over warm-up, timed and extra steps of this process"}
        if world == 1 and not args.no_probes:  # secondary probes and the host baseline belong to the single-GPU run (task contract: rank 0 at N=1 only)
            # the same warm-up / step protocol on more workloads, so that the driver's one default run carries them (never `value`):
            # the C3 step in the OTHER tracker mode (headline = Dropout(0.1) live, extra = eval-mode tracker, or the reverse) and C2
            if args.workload == "c3" and not args.dropout_redraw:
                out["dropout_off" if args.dropout > 0 else "dropout_on"] = timed_pass(WORKLOADS["c3"], device, 0.0 if args.dropout > 0 else 0.1, args.warmup, args.steps, G,
                                                                                      pretrain=args.pretrain_steps)
                if args.pretrain_steps > 0:      # the transient the driver's line used to time (rounds 1-5): same flags, policy fresh at the first warm-up step
                    out["fresh_policy"] = timed_pass(WORKLOADS["c3"], device, args.dropout, args.warmup, args.steps, G)
            if args.workload != "c2":
                # (C2 learns from 2 minibatch steps per update: after the driver's 5 + 20 steps its episodes are still 12 steps long and the number says
                #  nothing about the shape; a step is 1.7 ms, so this pass always runs >= 150 warm-up and >= 100 timed steps -- its own `steps` / `warmup` are reported)
                out["c2"] = timed_pass(WORKLOADS["c2"], device, args.dropout, max(args.warmup, 150), max(args.steps, 100), G)
            out["gather_fm"] = gather_fm_probe(device)
            out["deepfm_sweep"] = deepfm_sweep_probe(wl, device)
            out["sweep_mode"] = sweep_mode_probe(wl, eng, device)
            out["c5_split"] = c5_split_probe(device)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(wl, eng)
        # what else this run measured, in `config` (the driver's record keeps config / roofline / cpu_baseline whole and everything else by name only)
        brief = lambda e: {k: e[k] for k in ("value", "ms_per_step", "mean_episode_len", "tracker_dropout", "envs", "minibatch_steps_per_update", "steps", "warmup")}  # noqa: E731
        also = {"rollout_only_env_steps_per_s": out["rollout_only_env_steps_per_s"], "rollout_only_ms_per_collect": out["rollout_only_ms_per_collect"],
                "update_only_ms": out["update_only_ms"], "minibatch_step_us": 1e6 * t_mb, "minibatch_step_launches": out["minibatch_step"]["launches"],
                "ppo_minibatch_steps_per_s": out["ppo_minibatch_steps_per_s"]}
        for key in ("dropout_off", "dropout_on", "c2", "fresh_policy"):
            if key in out:
                also[key] = brief(out[key])
        out["config"]["also_measured"] = also
        # print order: the bulky probe objects first, the numbers a reader wants last (a log tail keeps the end of the line)
        tail_keys = ("roofline", "minibatch_step", "dropout_off", "dropout_on", "fresh_policy", "c2", "rollout_only_env_steps_per_s", "rollout_only_ms_per_collect",
                     "update_only_ms", "ppo_minibatch_steps_per_s")
        bulky = ("cpu_baseline", "gather_fm", "deepfm_sweep", "sweep_mode", "c5_split", "hbm_traffic_per_launch")
        ordered = {k: out[k] for k in bulky if k in out}
        ordered.update({k: v for k, v in out.items() if k not in bulky and k not in tail_keys})
        ordered.update({k: out[k] for k in tail_keys if k in out})
        ordered["summary"] = {"value": out["value"], "unit": out["unit"], "ms_per_step": out["ms_per_step"], "n_gpus": world,
                              "mean_episode_len": out["config"]["mean_episode_len"], "tracker_dropout": args.dropout, **also}
        print(json.dumps(ordered), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
