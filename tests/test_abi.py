"""CPU: libcirs_hip.so loads without a GPU and exports exactly the entry points include/cirs_hip.h declares."""
import ctypes as C
import os
import re

from cirs_hip import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "cirs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cirs_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = abi.lib()  # raises if the .so is missing or a bound symbol is absent
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/cirs_hip.h but not exported"
        assert s in abi.SIGNATURES, f"{s} has no ctypes signature in cirs_hip/abi.py"
    for s in abi.SIGNATURES:
        assert s in syms, f"{s} bound in abi.py but not declared in the header"


def test_version_and_error_plumbing():
    lib = abi.lib()
    assert lib.cirs_version() >= 100
    # argument validation happens on the host before any launch: safe without a GPU
    rc = lib.cirs_env_step(None, None, None, None, None, 0, None, None, None, None, None, None)
    assert rc == -1 and b"cfg" in lib.cirs_last_error()
    cfg = abi.EnvCfg(n_users=3, n_items=4, max_turn=5, num_leave_compute=1, leave_threshold=0, version=3, dist_mode=0, simulated=1)
    assert lib.cirs_env_step(C.byref(cfg), None, None, None, None, 1, None, None, None, None, None, None) == -1
    assert b"version" in lib.cirs_last_error()
    cfg.version = 1
    assert lib.cirs_env_step(C.byref(cfg), None, None, None, None, 0, None, None, None, None, None, None) == 0  # empty batch


def test_struct_sizes_match_header_layout():
    import ctypes as C
    assert C.sizeof(abi.EnvCfg) == 10 * 4 + 3 * 8
    assert C.sizeof(abi.TrackerWeights) == 8 * (7 + 12 * abi.MAX_TRACKER_LAYERS + 2)
    assert C.sizeof(abi.PpoCfg) == 17 * 4
    assert C.sizeof(abi.Traj) == 7 * 8
