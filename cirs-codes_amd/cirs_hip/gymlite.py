"""Minimal stand-in for the parts of `gym` the CIRS scripts touch (Env, spaces.Box, register/make), used only when the
real package is not installed.  `install()` publishes it as `sys.modules['gym']` so `import gym` /
`from gym.envs.registration import register` in an unmodified driver script keep working."""
import importlib
import sys
import types

import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape, self.dtype = shape, dtype


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        shape = tuple(shape) if shape is not None else np.shape(low)
        super().__init__(shape, dtype)
        self.low = np.full(shape, low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype)

    def sample(self):
        if np.issubdtype(self.dtype, np.integer):      # gym: integer boxes are sampled uniformly from [low, high] inclusive
            return np.random.randint(self.low.astype(np.int64), self.high.astype(np.int64) + 1).astype(self.dtype)
        return np.random.uniform(self.low, self.high).astype(self.dtype)


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = n

    def sample(self):
        return np.random.randint(self.n)


class Env:
    metadata = {}
    observation_space = None
    action_space = None

    def seed(self, seed=None):
        return [seed]

    def close(self):
        pass


registry = {}


def register(id, entry_point=None, kwargs=None, **_):
    registry[id] = (entry_point, dict(kwargs or {}))


def make(id, **extra):
    entry_point, kwargs = registry[id]
    if isinstance(entry_point, str):
        mod, cls = entry_point.split(":")
        entry_point = getattr(importlib.import_module(mod), cls)
    kw = dict(kwargs)
    kw.update(extra)
    return entry_point(**kw)


spaces = types.SimpleNamespace(Space=Space, Box=Box, Discrete=Discrete)


def install():
    """Make `import gym` resolve to this module if the real gym is absent."""
    try:
        import gym  # noqa: F401
        return sys.modules["gym"]
    except ImportError:
        me = sys.modules[__name__]
        reg = types.ModuleType("gym.envs.registration")
        reg.register, reg.registry = register, registry
        envs = types.ModuleType("gym.envs")
        envs.registration = reg
        sp = types.ModuleType("gym.spaces")
        sp.Space, sp.Box, sp.Discrete = Space, Box, Discrete
        me.envs = envs
        sys.modules.update({"gym": me, "gym.envs": envs, "gym.envs.registration": reg, "gym.spaces": sp})
        return me
