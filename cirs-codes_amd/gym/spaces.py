"""gym.spaces stand-in (see gym/__init__.py)."""
from cirs_hip.gymlite import Box, Discrete, Space  # noqa: F401
