"""Harness that imports the *reference* CIRS code (read-only, /root/reference) in the dev container.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path imports this file.  It exists so that
`oracle/gen_golden.py` can drive the reference's own Python implementation and record golden
input/output vectors under `tests/golden/`.  `/root/reference` does not exist on the GPU box, so
nothing here may be used by `-m gpu` tests, `bench.py` or `__graft_entry__.smoke()`.

What it does:
  * installs stub modules for third-party packages the container lacks (gym, numba, logzero, h5py,
    tensorboard, tensorflow) -- no reference file is modified or copied;
  * aliases `np.int` (removed in NumPy 2; used by core/env/simulatedEnv/simulated_env.py:176);
  * prepends the reference's roots to `sys.path`.
"""
import importlib
import importlib.machinery
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("CIRS_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "core"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []  # behave like a package so sub-imports resolve through sys.modules
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _install_numba():
    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    _mod("numba", njit=njit, jit=njit)


def _install_gym():
    class Space:
        def __init__(self, shape=None, dtype=None):
            self.shape = shape
            self.dtype = dtype

        def sample(self):
            raise NotImplementedError

        def seed(self, seed=None):     # gym seeds the space's own generator; the stub samples from numpy's global one
            return [seed]

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                shape = np.shape(low)
            super().__init__(tuple(shape), dtype)
            self.low = np.full(self.shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
            self.high = np.full(self.shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

    class Discrete(Space):
        def __init__(self, n):
            super().__init__((), np.int64)
            self.n = n

        def sample(self):
            return np.random.randint(self.n)

    class MultiDiscrete(Space):
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec)
            super().__init__(self.nvec.shape, np.int64)

    class MultiBinary(Space):
        def __init__(self, n):
            super().__init__((n,), np.int8)
            self.n = n

    class Dict(Space):
        def __init__(self, spaces=None, **kw):
            super().__init__()
            self.spaces = dict(spaces or {}, **kw)

    class Tuple(Space):
        def __init__(self, spaces):
            super().__init__()
            self.spaces = tuple(spaces)

    class Env:
        metadata = {}
        observation_space = None
        action_space = None

        def seed(self, seed=None):
            return [seed]

        def close(self):
            pass

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

    registry = {}

    def register(id, entry_point=None, kwargs=None, **_):
        registry[id] = (entry_point, dict(kwargs or {}))

    def make(id, **extra):
        entry_point, kwargs = registry[id]
        if isinstance(entry_point, str):
            mod_name, cls_name = entry_point.split(":")
            cls = getattr(importlib.import_module(mod_name), cls_name)
        else:
            cls = entry_point
        kw = dict(kwargs)
        kw.update(extra)
        return cls(**kw)

    spaces = _mod("gym.spaces", Space=Space, Box=Box, Discrete=Discrete, MultiDiscrete=MultiDiscrete,
                  MultiBinary=MultiBinary, Dict=Dict, Tuple=Tuple)
    registration = _mod("gym.envs.registration", register=register, registry=registry)
    envs = _mod("gym.envs", registration=registration)
    _mod("gym", Env=Env, Space=Space, Wrapper=Wrapper, spaces=spaces, envs=envs, register=register, make=make,
         __version__="0.0-stub")


def _install_misc():
    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    _mod("logzero", logger=_Logger(), logfile=lambda *a, **k: None)

    class _H5:  # h5py is only used for isinstance checks / hdf5 round trips that CIRS never calls
        pass

    _mod("h5py", Group=type("Group", (_H5,), {}), Dataset=type("Dataset", (_H5,), {}),
         File=type("File", (_H5,), {}))

    ea = _mod("tensorboard.backend.event_processing.event_accumulator", EventAccumulator=object)
    ep = _mod("tensorboard.backend.event_processing", event_accumulator=ea)
    be = _mod("tensorboard.backend", event_processing=ep)
    _mod("tensorboard", backend=be)

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None

    import torch.utils  # noqa: F401
    tb = _mod("torch.utils.tensorboard", SummaryWriter=SummaryWriter)
    import torch
    torch.utils.tensorboard = tb

    class History:
        def __init__(self):
            self.history = {}

        def on_train_begin(self, logs=None):
            self.epoch = []

        def on_epoch_end(self, epoch, logs=None):
            pass

    class _KerasCallback:
        def __init__(self, *a, **k):
            pass

    cb = _mod("tensorflow.python.keras.callbacks", History=History, CallbackList=object,
              EarlyStopping=type("EarlyStopping", (_KerasCallback,), {}),
              ModelCheckpoint=type("ModelCheckpoint", (_KerasCallback,), {}))
    ke = _mod("tensorflow.python.keras", callbacks=cb)
    py = _mod("tensorflow.python", keras=ke)
    _mod("tensorflow", python=py)


_installed = False


def install():
    """Install stubs + paths (idempotent).  Call before importing anything from the reference."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}; golden generation only runs in the dev container")
    for name in ("numba", "gym", "logzero", "h5py", "tensorflow"):
        try:
            importlib.import_module(name)
        except Exception:
            pass
    if "numba" not in sys.modules:
        _install_numba()
    if "gym" not in sys.modules:
        _install_gym()
    _install_misc() if "logzero" not in sys.modules else None
    if not hasattr(np, "int"):
        np.int = int  # noqa: NPY001  (reference uses np.int under NumPy 1.x)
    for p in ("environments/VirtualTaobao", "DeepCTR-Torch", "tianshou", ""):
        path = os.path.join(REF_ROOT, p) if p else REF_ROOT
        if path not in sys.path:
            sys.path.insert(0, path)
    _installed = True
