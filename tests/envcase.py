"""Shared helpers for env parity tests: build ABI structs from tables and drive teacher-forced episodes."""
import ctypes as C

import numpy as np

from cirs_hip import abi
from cirs_hip.synthetic import pack_item_cats


def env_cfg(n_users, n_items, *, num_leave_compute, leave_threshold, max_turn, tau, gamma_exposure, version,
            r_decay, has_ab, dist_mode=0, simulated=1, use_exposure=1):
    return abi.EnvCfg(n_users=n_users, n_items=n_items, max_turn=max_turn, num_leave_compute=num_leave_compute,
                      leave_threshold=leave_threshold, version=version, use_exposure=use_exposure,
                      has_ab=int(has_ab), dist_mode=dist_mode, simulated=simulated, tau=tau,
                      gamma_exposure=gamma_exposure, r_decay=r_decay)


def ab_env_tables(raw_uid, raw_pid, alpha_u, beta_i, n_users, n_items):
    """alpha_u[lbe_user.inverse_transform(u)], beta_i[lbe_photo.inverse_transform(a)] as float64
    (simulated_env.py:158-161; inverse_transform == classes_[idx])."""
    if alpha_u is None:
        return np.ones(n_users), np.ones(n_items)
    return (np.asarray(alpha_u)[raw_uid, 0].astype(np.float64), np.asarray(beta_i)[raw_pid, 0].astype(np.float64))


class HostEnv:
    """The CPU oracle behind the same struct layout the HIP library uses (host numpy arrays)."""

    def __init__(self, cfg, mat, normed_mat, dist, item_cats, alpha_env, beta_env, n_env):
        import oracle_lib
        self.lib = oracle_lib.lib()
        self.cfg = cfg
        self.n_env = n_env
        self.keep = dict(mat=None if mat is None else np.ascontiguousarray(mat, dtype=np.float64),
                         normed=None if normed_mat is None else np.ascontiguousarray(normed_mat, dtype=np.float64),
                         dist=None if dist is None else np.ascontiguousarray(dist, dtype=np.float64),
                         cats=np.ascontiguousarray(pack_item_cats(item_cats)),
                         alpha=np.ascontiguousarray(alpha_env, dtype=np.float64),
                         beta=np.ascontiguousarray(beta_env, dtype=np.float64))
        k = self.keep
        self.tab = abi.EnvTables(mat=None if k["mat"] is None else k["mat"].ctypes.data,
                                 normed_mat=None if k["normed"] is None else k["normed"].ctypes.data,
                                 dist=None if k["dist"] is None else k["dist"].ctypes.data,
                                 item_cats=k["cats"].ctypes.data, alpha_env=k["alpha"].ctypes.data,
                                 beta_env=k["beta"].ctypes.data)
        T = cfg.max_turn
        self.s = dict(user=np.zeros(n_env, np.int32), turn=np.zeros(n_env, np.int32), done=np.zeros(n_env, np.uint8),
                      hist=np.zeros((n_env, T), np.int32), cum=np.zeros(n_env, np.float64))
        s = self.s
        self.st = abi.EnvState(user=s["user"].ctypes.data, turn=s["turn"].ctypes.data, done=s["done"].ctypes.data,
                               hist_action=s["hist"].ctypes.data, cum_reward=s["cum"].ctypes.data)

    def reset(self, users, env_ids=None):
        users = np.ascontiguousarray(users, np.int32)
        n = len(users)
        obs = np.zeros(n, np.int64)
        ids = None if env_ids is None else np.ascontiguousarray(env_ids, np.int32)
        rc = self.lib.oracle_env_reset(C.byref(self.cfg), C.byref(self.st), users.ctypes.data,
                                       None if ids is None else ids.ctypes.data, n, obs.ctypes.data)
        assert rc == 0
        return obs

    def set_online(self, pred, minmax):
        """online-reward mode: raw user-model scores of this step's rows + the global (min, max)"""
        self.keep["pred"] = np.ascontiguousarray(pred, np.float32)
        self.keep["mm"] = np.ascontiguousarray(minmax, np.float32)
        self.tab.pred_online = self.keep["pred"].ctypes.data
        self.tab.pred_minmax = self.keep["mm"].ctypes.data

    def step(self, actions, env_ids):
        actions = np.ascontiguousarray(actions, np.int64)
        ids = np.ascontiguousarray(env_ids, np.int32)
        n = len(ids)
        obs = np.zeros(n, np.int64); rew = np.zeros(n); done = np.zeros(n, np.uint8)
        ctr = np.zeros(n); expo = np.zeros(n)
        rc = self.lib.oracle_env_step(C.byref(self.cfg), C.byref(self.tab), C.byref(self.st), actions.ctypes.data,
                                      ids.ctypes.data, n, obs.ctypes.data, rew.ctypes.data, done.ctypes.data,
                                      ctr.ctypes.data, expo.ctypes.data)
        assert rc == 0
        return obs, rew, done.astype(bool), ctr, expo


def run_teacher_forced(env, users, acts, max_turn):
    """Step all envs in lock-step with recorded actions, dropping finished envs like the Collector does
    (reference core/collector.py:303-311).  Returns dense [B,T] arrays (nan / -1 beyond episode end)."""
    B = len(users)
    obs = np.full((B, max_turn), -1, np.int64); rew = np.full((B, max_turn), np.nan)
    done = np.zeros((B, max_turn), bool); ctr = np.full((B, max_turn), np.nan); expo = np.full((B, max_turn), np.nan)
    length = np.zeros(B, np.int64)
    o0 = env.reset(users)
    assert np.array_equal(o0, users)
    ready = np.arange(B)
    for t in range(max_turn):
        if len(ready) == 0:
            break
        o, r, d, c, x = env.step(acts[ready, t], ready)
        obs[ready, t] = o; rew[ready, t] = r; done[ready, t] = d; ctr[ready, t] = c; expo[ready, t] = x
        length[ready] = t + 1
        ready = ready[~d]
    return dict(obs=obs, rew=rew, done=done, ctr=ctr, expo=expo, length=length)


def load_env_cases(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "env_step.npz"))
    base = {k: z[k] for k in ("mat", "normed_mat", "dist", "item_cats", "raw_uid", "raw_pid", "alpha_u", "beta_i")}
    cases = []
    for ci in range(int(z["n_cases"])):
        pre = f"c{ci}_"
        c = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
        N, thr, T, tau, gam, ver, rdec, ab = c["cfg"]
        c["params"] = dict(num_leave_compute=int(N), leave_threshold=int(thr), max_turn=int(T), tau=float(tau),
                           gamma_exposure=float(gam), version=int(ver), r_decay=float(rdec), has_ab=bool(ab))
        cases.append(c)
    return base, cases


def compare_env_run(got, want, *, rtol=1e-12, what=""):
    """Exit decisions / observations / lengths bit-exact; float64 rewards to rtol (spec: 1e-4, oracle aims 1e-12)."""
    assert np.array_equal(got["length"], want["length"]), f"{what}: episode lengths differ"
    assert np.array_equal(got["done"], want["done"]), f"{what}: done flags differ"
    assert np.array_equal(got["obs"], want["obs"]), f"{what}: obs differ"
    for k in ("rew", "ctr", "expo"):
        a, b = got[k], want[k]
        assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: {k} validity differs"
        m = ~np.isnan(b)
        np.testing.assert_allclose(a[m], b[m], rtol=rtol, atol=1e-300, err_msg=f"{what}: {k}")
