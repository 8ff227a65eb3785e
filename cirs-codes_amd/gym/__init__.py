"""`import gym` for the reference's entry points (CIRS-RL-kuaishou.py:10,43) when the real package is not installed: the parts the
CIRS scripts touch (Env, spaces.Box / Discrete, envs.registration.register, make) from cirs_hip.gymlite.  An installed gym found
elsewhere on sys.path replaces this package at import time (cirs_hip.compat.defer_to_real)."""
import os as _os

from cirs_hip import compat as _compat

_real = _compat.defer_to_real("gym", _os.path.dirname(_os.path.abspath(__file__)))
if _real is None:
    from cirs_hip.gymlite import Box, Discrete, Env, Space, make, register, registry  # noqa: F401
    from . import envs, spaces  # noqa: F401
    __cirs_stand_in__ = True
_compat.ensure_tensorboard()
