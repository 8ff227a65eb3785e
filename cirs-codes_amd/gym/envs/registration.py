"""gym.envs.registration stand-in: register(id, entry_point, kwargs) / make(id) (CIRS-RL-kuaishou.py:171-204)."""
from cirs_hip.gymlite import make, register, registry  # noqa: F401
