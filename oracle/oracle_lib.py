"""ctypes loader for the CPU oracle (oracle/_build/libcirs_oracle.so).  TEST INFRASTRUCTURE ONLY.

Importable only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product
package.  Struct layouts come from the product's ABI mirror so both sides are called with identical arguments
(host pointers here, device pointers there).
"""
import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
sys.path.insert(0, os.path.join(_ROOT, "cirs-codes_amd"))

from cirs_hip import abi  # noqa: E402

LIB_PATH = os.path.join(_HERE, "_build", "libcirs_oracle.so")
_P = C.c_void_p
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        h = C.CDLL(LIB_PATH)
        h.oracle_env_reset.restype = C.c_int
        h.oracle_env_reset.argtypes = [C.POINTER(abi.EnvCfg), C.POINTER(abi.EnvState), _P, _P, C.c_int32, _P]
        h.oracle_env_step.restype = C.c_int
        h.oracle_env_step.argtypes = [C.POINTER(abi.EnvCfg), C.POINTER(abi.EnvTables), C.POINTER(abi.EnvState),
                                      _P, _P, C.c_int32, _P, _P, _P, _P, _P]
        h.oracle_dist_jaccard.restype = C.c_int
        h.oracle_dist_jaccard.argtypes = [_P, C.c_int32, _P]
        h.oracle_gae_return.restype = C.c_int
        h.oracle_gae_return.argtypes = [_P, _P, _P, _P, C.c_long, C.c_double, C.c_double, _P]
        h.oracle_actor_gumbel.restype = C.c_float
        h.oracle_actor_gumbel.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        h.oracle_actor_sample.restype = C.c_int
        h.oracle_actor_sample.argtypes = [C.POINTER(abi.PolicyCfg), C.POINTER(abi.PolicyWeights), _P, C.c_int64,
                                          C.c_int32, _P, C.c_uint64, C.c_uint32, _P, _P, _P, _P, _P, _P, _P]
        h.oracle_actor_sample_margins.restype = C.c_int
        h.oracle_actor_sample_margins.argtypes = [C.POINTER(abi.PolicyCfg), C.POINTER(abi.PolicyWeights), _P, C.c_int64,
                                                  C.c_int32, C.c_uint64, C.c_uint32, _P, _P, _P, _P, _P, _P, _P]
        h.oracle_deepfm_forward.restype = C.c_int
        h.oracle_deepfm_forward.argtypes = [C.POINTER(abi.DeepFMCfg), C.POINTER(abi.DeepFMWeights), _P, _P, _P, _P, C.c_int32, _P]
        h.oracle_dropout_keep.restype = C.c_int
        h.oracle_dropout_keep.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float]
        h.oracle_gather_fm.restype = C.c_int
        h.oracle_gather_fm.argtypes = [C.POINTER(abi.DeepFMCfg), C.POINTER(abi.DeepFMWeights), _P, C.c_int64, _P]
        h.oracle_select_items.restype = C.c_int
        h.oracle_select_items.argtypes = [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_float, _P, C.c_uint64,
                                          C.c_uint32, _P, _P]
        h.oracle_exposure_history.restype = C.c_int
        h.oracle_exposure_history.argtypes = [_P, _P, _P, C.c_int64, _P, _P, C.c_int32, C.c_double, _P]
        h.oracle_find_negative.restype = C.c_int
        h.oracle_find_negative.argtypes = [_P, _P, C.c_int64, _P, _P, C.c_int32, C.c_int64, _P]
        h.oracle_random_permutation.restype = C.c_int
        h.oracle_random_permutation.argtypes = [C.c_int64, C.c_uint64, C.c_uint64, _P]
        h.oracle_hash_ids.restype = C.c_int
        h.oracle_hash_ids.argtypes = [_P, C.c_int64, C.c_int64, _P]
        h.oracle_eval_coverage.restype = C.c_int
        h.oracle_eval_coverage.argtypes = [_P, C.c_int64, C.c_int32, _P, _P, _P]
        _lib = h
    return _lib
