"""Vector-env protocol (reference tianshou/tianshou/env/venvs.py:153-315).  DummyVectorEnv's serial Python loop over
env objects is replaced by ONE batched device env: the env objects the factories return are specs that share their
tables, and every vector step is a single cirs_env_step launch."""
from typing import Callable, List, Optional

import numpy as np
import torch


class BaseVectorEnv:
    is_async = False

    def __init__(self, env_fns: List[Callable]):
        self._specs = [fn() for fn in env_fns]
        self.env_num = len(self._specs)
        first = self._specs[0]
        # host mode: envs that step on the CPU with gym's reset / step protocol (VirtualTB-v0 and SimulatedEnv over it: BASELINE
        # configs[0], CPU plumbing like in the reference) are looped over like the reference's DummyVectorEnv does
        self.host_mode = getattr(first, "env_name", None) == "VirtualTB-v0" or type(first).__name__ == "VirtualTB"
        if not self.host_mode:
            for s in self._specs[1:]:
                assert type(s) is type(first) and s.batch_key() == first.batch_key(), "all envs of a vector env must share tables and parameters"
        self.workers = self._specs
        self._dev = None  # cirs_hip.env.DeviceEnv, built on first use (needs the GPU)
        self._user_rng = np.random.RandomState()
        self._want_info = True

    def __len__(self):
        return self.env_num

    @property
    def action_space(self):
        return [s.action_space for s in self._specs]

    def __getattr__(self, key):
        if key.startswith("_"):
            raise AttributeError(key)
        return [getattr(s, key) for s in self._specs]

    def device_env(self):
        if self._dev is None:
            self._dev = self._specs[0].build_device_env(self.env_num)
        return self._dev

    def seed(self, seed=None):
        if self.host_mode:       # venvs.py:263-283: worker i is seeded with seed + i (SimulatedEnv.seed seeds torch's generator)
            seeds = [None] * self.env_num if seed is None else ([seed + i for i in range(self.env_num)] if np.isscalar(seed) else list(seed))
            return [w.seed(s) for w, s in zip(self._specs, seeds)]
        if seed is not None:
            s0 = seed if np.isscalar(seed) else seed[0]
            self._user_rng = np.random.RandomState(int(s0))
        return [seed] * self.env_num

    def draw_users(self, n):
        """KuaishouEnv.__user_generator (kuaishouEnv.py:155-159): uniform over users.  The reference draws with
        Python's unseeded `random`; here the stream is seedable through env.seed()."""
        return self._user_rng.randint(0, self._specs[0].n_users, n)

    def reset(self, id=None, users=None):
        if self.host_mode:
            ids = range(self.env_num) if id is None else np.atleast_1d(id)
            return np.stack([self._specs[i].reset() for i in ids])
        dev = self.device_env()
        ids = np.arange(self.env_num) if id is None else np.atleast_1d(id)
        users = self.draw_users(len(ids)) if users is None else np.asarray(users)
        obs = dev.reset(torch.as_tensor(users), None if id is None else torch.as_tensor(ids))
        return obs.cpu().numpy().reshape(-1, 1)

    def step(self, action, id=None):
        if self.host_mode:
            from tianshou.data import Batch
            ids = range(self.env_num) if id is None else np.atleast_1d(id)
            res = [self._specs[i].step(action[j]) for j, i in enumerate(ids)]
            obs, rew, done, infos = zip(*res)
            keys = infos[0].keys() if len(infos) and isinstance(infos[0], dict) else []
            info = Batch({k: np.array([inf[k] for inf in infos]) for k in keys}, env_id=np.asarray(list(ids)))
            return np.stack(obs), np.stack(rew), np.stack(done), info
        dev = self.device_env()
        ids = np.arange(self.env_num) if id is None else np.atleast_1d(id)
        o, r, d, c, _ = dev.step(torch.as_tensor(np.asarray(action).reshape(-1)), torch.as_tensor(ids))
        ctr = c.cpu().numpy()
        key = "CTR" if self._specs[0].simulated else "cum_reward"
        info = np.array([{key: float(ctr[j]), "env_id": int(ids[j])} for j in range(len(ids))], dtype=object)
        return o.cpu().numpy().reshape(-1, 1), r.cpu().numpy(), d.cpu().numpy().astype(bool), info

    def render(self, **kwargs):
        return [None] * self.env_num

    def close(self):
        pass


class DummyVectorEnv(BaseVectorEnv):
    pass


SubprocVectorEnv = ShmemVectorEnv = DummyVectorEnv
