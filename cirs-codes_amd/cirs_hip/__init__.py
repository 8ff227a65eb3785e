"""cirs_hip -- Python binding of libcirs_hip.so (hand-written HIP kernels for gfx950 / MI355X).

Layout of the package root `cirs-codes_amd/` (put it on sys.path to get the reference's module names):
  csrc/          HIP kernels + the C ABI declared in include/cirs_hip.h
  cirs_hip/      this binding: ctypes loader (abi.py), device engines, synthetic KuaiRec-shaped tables
  core/, tianshou/, environments/   host-side mirror of the reference's plugin surface for the hot path
"""
from . import abi  # noqa: F401
