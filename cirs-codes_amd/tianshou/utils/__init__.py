"""tianshou.utils: the logger classes the entry points construct (`from tianshou.utils import BasicLogger`, CIRS-RL-kuaishou.py:27)."""
from cirs_hip import compat as _compat

_compat.ensure_tensorboard()
from tianshou.utils.log_tools import BaseLogger, BasicLogger  # noqa: E402,F401
