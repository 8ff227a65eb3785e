"""gym.envs stand-in (see gym/__init__.py)."""
from . import registration  # noqa: F401
