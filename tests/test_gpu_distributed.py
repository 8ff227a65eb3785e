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
