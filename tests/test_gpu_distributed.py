"""GPU (single device): the multi-rank code path with one rank -- NCCL process group of size 1, trajectories pushed
through the packed all-gather and the global-buffer learner/tracker-backward path; results must equal the direct path."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def build(force_gather):
    from cirs_hip.engine import CirsEngine
    from cirs_hip.env import DeviceEnvTables
    from cirs_hip.synthetic import make_tables
    tab = make_tables(120, 300, seed=0, build_dist=False)
    a_env = tab.alpha_u[tab.raw_uid, 0].astype(np.float64); b_env = tab.beta_i[tab.raw_pid, 0].astype(np.float64)
    dt = DeviceEnvTables(tab.mat, tab.normed_mat, tab.item_cats, alpha_env=a_env, beta_env=b_env)
    return CirsEngine(dt, 40, max_turn=12, num_leave_compute=3, leave_threshold=1, tau=10.0, gamma_exposure=10.0, seed=5,
                      force_gather=force_gather)


def test_gathered_path_equals_direct_path():
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        a, b = build(False), build(True)
        users = torch.as_tensor(np.random.RandomState(1).randint(0, 120, 40))
        for eng in (a, b):
            eng.collect(users)
        n = int(a.lengths.sum())
        perms = [np.random.RandomState(3 + k).permutation(n) for k in range(2)]
        la, na = a.update(16, 2, perms=perms)
        lb, nb = b.update(16, 2, perms=perms)
        assert na == nb == n
        assert torch.equal(la, lb)
        assert torch.equal(a.policy_flat, b.policy_flat)
        assert torch.equal(a.tracker_flat, b.tracker_flat)   # deterministic reductions: identical bits
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t)
        dist.barrier()
        assert float(t) == 1.0
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_data_parallel_learner_equals_single_device_large_batch():
    """Two virtual ranks on one GPU: row-sharded phase-1 gradients summed (= all-reduce) + phase 2 on both must track a
    single learner that runs the same GLOBAL minibatches (batch_size * world) on one device."""
    from cirs_hip.rollout import Trajectory
    import nn_oracle
    import policycase
    import rolloutcase
    from test_gpu_learn import make_learner, rollout_time_value_logp, upload_traj
    I, B, T, W, bs = 900, 40, 12, 2, 64
    rng = np.random.RandomState(11)
    tp = rolloutcase.tracker_param_dict(100, I, T, seed=1)
    arrs = policycase.random_weights(rng, I, head_scale=1.5)
    pp = {k: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)) for k, v in arrs.items()}
    lens = rng.randint(5, T + 1, size=B)
    users = rng.randint(0, 100, B); acts = rng.randint(0, I, (B, T)); rews = rng.uniform(0, 1, (B, T))
    dones = np.zeros((B, T), bool); dones[np.arange(B), lens - 1] = True
    with torch.no_grad():
        obs_bts = nn_oracle.tracker_states(tp, users, acts, rews).numpy()
    value, logp = rollout_time_value_logp(pp, obs_bts, acts, lens)
    n = int(lens.sum())
    perms = [rng.permutation(n) for _ in range(2)]
    hyper = [0.95, 0.95, 0.2, 0.25, 0.01, 0.5, 1e-3, bs, 2]
    traj = Trajectory(B, T, 20, "cuda")
    upload_traj(traj, acts, rews, dones, lens, obs_bts, value, logp)
    # single device, global batch
    ref, ref_views = make_learner(pp, I, B, T, hyper)
    ref.prepare(traj, lens)
    ref_losses = ref.learn(bs * W, 2, perms=perms)
    # two virtual ranks
    ranks = [make_learner(pp, I, B, T, hyper)[0] for _ in range(W)]
    for ln in ranks:
        ln.prepare(traj, lens)
    from cirs_hip.learner import minibatch_slices
    slices = minibatch_slices(n, bs * W)
    losses = torch.zeros((2 * len(slices), 4), device="cuda")
    k = 0
    for rep in range(2):
        perm_d = torch.as_tensor(perms[rep].astype(np.int32)).cuda()
        for ln in ranks:
            if rep == 1:
                ln.dobs.zero_()
        for s0, e0 in slices:
            g_idx = perm_d[s0:e0]
            shards = [g_idx[r::W].contiguous() for r in range(W)]
            for r, ln in enumerate(ranks):
                ln.mb_phase1(shards[r], g_idx, rep == 1, losses[k])
            total = sum(ln.grads for ln in ranks)   # the all-reduce
            for r, ln in enumerate(ranks):
                ln.grads.copy_(total)
                ln.mb_phase2(int(shards[r].numel()), int(g_idx.numel()), losses[k])
            k += 1
    assert torch.equal(ranks[0].params, ranks[1].params)          # ranks stay bit-identical
    np.testing.assert_allclose(losses.cpu().numpy(), ref_losses.cpu().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(ranks[0].params.cpu().numpy(), ref.params.cpu().numpy(), rtol=2e-4, atol=2e-6)
    dobs = sum(ln.dobs for ln in ranks)
    np.testing.assert_allclose(dobs.cpu().numpy(), ref.dobs.cpu().numpy(), rtol=2e-3, atol=1e-7)


def test_data_parallel_path_over_rccl_with_one_rank(monkeypatch):
    """The data-parallel learner (phase 1 -> all-reduce of the flat gradients -> phase 2, all-reduce of d loss/d obs and of the
    tracker gradients) driven through a real NCCL/RCCL group of size 1: must track the single-rank learner (the two paths sum
    gradient slabs in different launches, so float tolerance, not bits)."""
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        a = build(False)
        monkeypatch.setenv("CIRS_FORCE_DP", "1")
        b = build(True)
        assert b.force_dp
        users = torch.as_tensor(np.random.RandomState(1).randint(0, 120, 40))
        for eng in (a, b):
            eng.collect(users)
        n = int(a.lengths.sum())
        for k in range(2):
            perms = [np.random.RandomState(3 + 2 * k + q).permutation(n) for q in range(2)]
            la, na = a.update(16, 2, perms=perms)
            lb, nb = b.update(16, 2, perms=perms)
            assert na == nb == n
            np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(b.policy_flat.cpu().numpy(), a.policy_flat.cpu().numpy(), rtol=1e-3, atol=2e-5)
        # Adam turns round-off in near-zero gradients into O(lr) parameter differences: a handful of elements move by ~1e-4
        np.testing.assert_allclose(b.tracker_flat.cpu().numpy(), a.tracker_flat.cpu().numpy(), rtol=1e-3, atol=3e-4)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs on the node (RCCL over xGMI with N > 1 ranks)")
@pytest.mark.parametrize("learner", ["dp", "dp_sharded", "replicated", "tp"])
@pytest.mark.parametrize("nproc", [2])
def test_two_process_rccl_bench_ranks_stay_bit_identical(nproc, learner):
    """The driver's SCALE run must not be the first time RCCL sees N > 1 ranks: launch bench.py exactly as the driver does
    (torch.distributed.run, one process per GPU), 3 timed steps, and require (a) a valid JSON line, (b) n_gpus == nproc and
    envs_total == 1024 * nproc (env-sharded, weak scaling), (c) bit-identical policy / tracker parameters on every rank after
    the updates (the all-gather + per-minibatch all-reduce path is order-fixed)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29700 + os.getpid() % 200
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1",
           "--learner", learner, "--scaled-batch-steps", "1"]
    env["CIRS_DIST_CHECK"] = "1"     # per-collective stream-order assertions (cirs_hip.distributed.Collectives)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    z = json.loads(line)
    assert z["n_gpus"] == nproc and z["config"]["envs_total"] == 1024 * nproc and z["scaling"] == "weak"
    assert z["config"]["learner"] == learner and z["config"]["global_minibatch"] == 1024
    assert z["rank_parameters_bit_identical"] is True
    assert z["value"] > 0 and z["steps"] == 3
    assert z["scaled_batch_variant"]["global_minibatch"] == 1024 * nproc


@pytest.mark.parametrize("learner", ["dp", "dp_sharded", "tp", "replicated"])
def test_two_process_bench_on_one_gpu_ranks_stay_bit_identical(learner):
    """The same launch as above on a box with ONE GPU: both ranks on device 0 (CIRS_BENCH_SHARE_GPU=1), collectives through gloo on
    device tensors.  Everything but the transport is the N > 1 path of the driver's SCALE run: env sharding, the packed
    all-gather of the trajectory, the data-parallel learner's per-minibatch all-reduce, rank-identity of the parameters."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29500 + os.getpid() % 150
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CIRS_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-probes",
           "--learner", learner, "--scaled-batch-steps", "1"]
    env["CIRS_DIST_CHECK"] = "1"
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    z = json.loads(line)
    assert z["n_gpus"] == 2 and z["config"]["envs_total"] == 2048 and z["scaling"] == "weak"
    # the headline keeps the reference's PPO configuration: global minibatch 1024 rows = 512 per rank, ~2 x the single-GPU step count
    assert z["config"]["learner"] == learner and z["config"]["global_minibatch"] == 1024 and z["config"]["rows_per_rank_per_minibatch"] == (1024 if learner in ("tp", "replicated") else 512)
    assert z["rank_parameters_bit_identical"] is True
    assert z["value"] > 0 and z["steps"] == 2
    sv = z["scaled_batch_variant"]
    # same buffer sizes, half the minibatch size -> about twice the optimiser steps per update (episode lengths drift a little between
    # the two measurements: the policy keeps learning)
    assert sv["global_minibatch"] == 2048 and 1.2 < z["config"]["minibatch_steps_per_update"] / sv["minibatch_steps_per_update"] < 3.0
    calls = z["collectives_per_rank"]["calls"]
    assert (calls["reduce_scatter"] > 0) == (learner == "dp_sharded")
