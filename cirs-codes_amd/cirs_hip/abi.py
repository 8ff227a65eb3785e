"""ctypes mirror of include/cirs_hip.h and the loader of libcirs_hip.so.

The HIP library is the product: there is NO CPU fallback.  `lib()` raises if the shared object is missing or
does not export every symbol the header declares.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CIRS_HIP_LIB: an explicit path to another build of the library (same-box A/B runs of tools/ab_step_libs.py, tests against a probe build); default: the in-tree build
LIB_PATH = os.environ.get("CIRS_HIP_LIB") or os.path.join(_HERE, "libcirs_hip.so")

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)
c_u32p = C.POINTER(C.c_uint32)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)


class EnvCfg(C.Structure):
    _fields_ = [("n_users", C.c_int32), ("n_items", C.c_int32), ("max_turn", C.c_int32),
                ("num_leave_compute", C.c_int32), ("leave_threshold", C.c_int32), ("version", C.c_int32),
                ("use_exposure", C.c_int32), ("has_ab", C.c_int32), ("dist_mode", C.c_int32),
                ("simulated", C.c_int32), ("tau", C.c_double), ("gamma_exposure", C.c_double),
                ("r_decay", C.c_double)]


class EnvTables(C.Structure):
    _fields_ = [("mat", C.c_void_p), ("normed_mat", C.c_void_p), ("dist", C.c_void_p),
                ("item_cats", C.c_void_p), ("alpha_env", C.c_void_p), ("beta_env", C.c_void_p),
                ("pred_online", C.c_void_p), ("pred_minmax", C.c_void_p)]


class EnvState(C.Structure):
    _fields_ = [("user", C.c_void_p), ("turn", C.c_void_p), ("done", C.c_void_p),
                ("hist_action", C.c_void_p), ("cum_reward", C.c_void_p)]


MAX_TRACKER_LAYERS = 4


class TrackerCfg(C.Structure):
    _fields_ = [("n_users", C.c_int32), ("n_items", C.c_int32), ("dim_model", C.c_int32), ("dim_state", C.c_int32),
                ("nhead", C.c_int32), ("d_hid", C.c_int32), ("nlayers", C.c_int32), ("max_len", C.c_int32),
                ("n_env", C.c_int32), ("dropout_p", C.c_float), ("drop_env_base", C.c_int32), ("dropout_seed", C.c_uint64)]


class TrackerLayer(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b",
                                          "lin2_w", "lin2_b", "norm1_w", "norm1_b", "norm2_w", "norm2_b")]


class TrackerWeights(C.Structure):
    _fields_ = [("emb_user", C.c_void_p), ("emb_item", C.c_void_p), ("ffn_user_w", C.c_void_p),
                ("ffn_user_b", C.c_void_p), ("gate_w", C.c_void_p), ("gate_b", C.c_void_p), ("pe", C.c_void_p),
                ("layer", TrackerLayer * MAX_TRACKER_LAYERS), ("dec_w", C.c_void_p), ("dec_b", C.c_void_p)]


class TrackerState(C.Structure):
    _fields_ = [("x_hist", C.c_void_p), ("kcache", C.c_void_p), ("vcache", C.c_void_p), ("len", C.c_void_p)]


class PolicyCfg(C.Structure):
    _fields_ = [("n_items", C.c_int32), ("dim_state", C.c_int32), ("hidden", C.c_int32)]


class PolicyWeights(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("w1", "b1", "w2", "b2", "wa", "ba", "wc", "bc")]


class PpoCfg(C.Structure):
    _fields_ = [("n_items", C.c_int32), ("dim_state", C.c_int32), ("hidden", C.c_int32), ("norm_adv", C.c_int32),
                ("value_clip", C.c_int32), ("rew_norm", C.c_int32), ("gamma", C.c_float), ("gae_lambda", C.c_float),
                ("eps_clip", C.c_float), ("vf_coef", C.c_float), ("ent_coef", C.c_float), ("max_grad_norm", C.c_float),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float), ("dual_clip", C.c_float)]


class PpoBatch(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("obs", "act", "adv", "ret", "v_s", "logp_old", "row_env", "row_t")]


class DeepFMCfg(C.Structure):
    _fields_ = [("n_user_vocab", C.c_int32), ("n_item_vocab", C.c_int32), ("n_feat_vocab", C.c_int32),
                ("emb_dim", C.c_int32), ("hidden", C.c_int32)]


DEEPFM_FIELDS = ("emb_user", "emb_item", "emb_feat", "lin_user", "lin_item", "lin_feat", "lin_dense", "w1", "b1", "w2", "b2",
                 "last", "out_bias")


class DeepFMWeights(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in DEEPFM_FIELDS]


class OnlineReward(C.Structure):
    _fields_ = [("cfg", C.POINTER(DeepFMCfg)), ("w", C.POINTER(DeepFMWeights))] + [
        (k, C.c_void_p) for k in ("raw_uid", "raw_pid", "item_feats", "item_dur", "pred_minmax", "uid_buf", "pid_buf",
                                  "feat_buf", "dur_buf", "pred_buf")]


class Traj(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("obs", "act", "rew", "done", "logp", "value", "ctr")]


class Redraw(C.Structure):      # cirs_redraw
    _fields_ = [("row_env", C.c_void_p), ("row_t", C.c_void_p), ("offsets", C.c_void_p), ("lens", C.c_void_p), ("dropout_seed", C.c_uint64),
                ("env_base0", C.c_int64), ("env_stride", C.c_int64), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]


# name -> (restype, argtypes).  Must list every symbol include/cirs_hip.h declares (tests check this).
_P = C.c_void_p
SIGNATURES = {
    "cirs_last_error": (C.c_char_p, []),
    "cirs_version": (C.c_int, []),
    "cirs_env_reset": (C.c_int, [C.POINTER(EnvCfg), C.POINTER(EnvState), _P, _P, C.c_int32, _P, _P]),
    "cirs_env_step": (C.c_int, [C.POINTER(EnvCfg), C.POINTER(EnvTables), C.POINTER(EnvState), _P, _P, C.c_int32,
                                _P, _P, _P, _P, _P, _P]),
    "cirs_dist_jaccard": (C.c_int, [_P, C.c_int32, _P, _P]),
    "cirs_tracker_init": (C.c_int, [C.POINTER(TrackerCfg), C.POINTER(TrackerWeights), C.POINTER(TrackerState), _P, _P,
                                    C.c_int32, _P, C.c_int64, _P]),
    "cirs_tracker_step": (C.c_int, [C.POINTER(TrackerCfg), C.POINTER(TrackerWeights), C.POINTER(TrackerState), _P, _P,
                                    _P, _P, C.c_int32, _P, C.c_int64, _P]),
    "cirs_policy_workspace_bytes": (C.c_int64, [C.POINTER(PolicyCfg), C.c_int32]),
    "cirs_actor_sample": (C.c_int, [C.POINTER(PolicyCfg), C.POINTER(PolicyWeights), _P, C.c_int64, C.c_int32, _P,
                                    C.c_uint64, C.c_uint32, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "cirs_critic_values": (C.c_int, [C.POINTER(PolicyCfg), C.POINTER(PolicyWeights), _P, C.c_int64, C.c_int32, _P, _P, C.c_int64, _P]),
    "cirs_rollout_steps": (C.c_int, [C.POINTER(EnvCfg), C.POINTER(EnvTables), C.POINTER(EnvState),
                                     C.POINTER(TrackerCfg), C.POINTER(TrackerWeights), C.POINTER(TrackerState),
                                     C.POINTER(PolicyCfg), C.POINTER(PolicyWeights), C.POINTER(Traj), C.c_int32,
                                     C.c_int32, C.c_int32, C.c_uint64, C.c_uint32, _P, C.c_int32, _P, C.c_int64, _P]),
    "cirs_rollout_collect": (C.c_int, [C.POINTER(EnvCfg), C.POINTER(EnvTables), C.POINTER(EnvState),
                                       C.POINTER(TrackerCfg), C.POINTER(TrackerWeights), C.POINTER(TrackerState),
                                       C.POINTER(PolicyCfg), C.POINTER(PolicyWeights), C.POINTER(Traj), C.c_int32,
                                       _P, C.c_uint64, C.c_uint32, _P, C.c_int32, _P, C.c_int64, _P]),
    "cirs_rollout_steps_redraw": (C.c_int, [C.POINTER(EnvCfg), C.POINTER(EnvTables), C.POINTER(EnvState),
                                            C.POINTER(TrackerCfg), C.POINTER(TrackerWeights), C.POINTER(TrackerState),
                                            C.POINTER(PolicyCfg), C.POINTER(PolicyWeights), C.POINTER(Traj), C.c_int32,
                                            C.c_int32, C.c_int32, C.c_uint64, C.c_uint32, C.POINTER(Redraw), _P, C.c_int64, _P]),
    "cirs_rollout_steps_noise": (C.c_int, [C.POINTER(EnvCfg), C.POINTER(EnvTables), C.POINTER(EnvState),
                                           C.POINTER(TrackerCfg), C.POINTER(TrackerWeights), C.POINTER(TrackerState),
                                           C.POINTER(PolicyCfg), C.POINTER(PolicyWeights), C.POINTER(Traj), C.c_int32,
                                           C.c_int32, C.c_int32, _P, _P, C.c_int32, _P, C.c_int64, _P]),
    "cirs_rollout_steps_online": (C.c_int, [C.POINTER(EnvCfg), C.POINTER(EnvTables), C.POINTER(EnvState),
                                            C.POINTER(TrackerCfg), C.POINTER(TrackerWeights), C.POINTER(TrackerState),
                                            C.POINTER(PolicyCfg), C.POINTER(PolicyWeights), C.POINTER(Traj), C.c_int32,
                                            C.c_int32, C.c_int32, C.c_uint64, C.c_uint32, _P, C.c_int32,
                                            C.POINTER(OnlineReward), _P, C.c_int64, _P]),
    "cirs_ppo_param_count": (C.c_int64, [C.POINTER(PpoCfg)]),
    "cirs_ppo_workspace_bytes": (C.c_int64, [C.POINTER(PpoCfg), C.c_int32]),
    "cirs_ppo_prepare_async": (C.c_int, [C.POINTER(PpoCfg), C.POINTER(Traj), _P, C.c_int32, C.c_int32, _P, _P, _P, C.POINTER(PpoBatch), _P, _P]),
    "cirs_ppo_prepare_async_perms": (C.c_int, [C.POINTER(PpoCfg), C.POINTER(Traj), _P, C.c_int32, C.c_int32, _P, _P, _P, C.POINTER(PpoBatch), _P, C.c_uint64, C.c_uint64, C.c_int32, _P, C.c_int32, _P]),
    "cirs_ppo_prepare": (C.c_int, [C.POINTER(PpoCfg), C.POINTER(Traj), _P, _P, C.c_int32, C.c_int32, C.c_int32, _P,
                                   C.POINTER(PpoBatch), _P]),
    "cirs_ppo_minibatch": (C.c_int, [C.POINTER(PpoCfg), _P, _P, _P, _P, C.c_int64, C.POINTER(PpoBatch), _P, C.c_int32,
                                     _P, C.c_int32, _P, _P, C.c_int64, _P]),
    "cirs_ppo_minibatch_dp_chain": (C.c_int, [C.POINTER(PpoCfg), _P, _P, _P, _P, C.c_int64, C.POINTER(PpoBatch), _P, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P,
                                              _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P]),
    "cirs_ppo_learn_steps": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "cirs_ppo_handoff_status": (C.c_int, [_P, C.c_int32, _P]),
    "cirs_ppo_update_readback": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "cirs_ppo_learn": (C.c_int, [C.POINTER(PpoCfg), _P, _P, _P, _P, C.c_int64, C.POINTER(PpoBatch), _P, C.c_int32, C.c_int32, C.c_int32,
                                 _P, C.c_int64, C.c_int32, _P, _P, C.c_int64, _P]),
    "cirs_tracker_backward_workspace_bytes": (C.c_int64, [C.POINTER(TrackerCfg), C.c_int32]),
    "cirs_tracker_backward": (C.c_int, [C.POINTER(TrackerCfg), C.POINTER(TrackerWeights), C.POINTER(TrackerState), _P, _P,
                                        _P, _P, _P, _P, _P, C.c_int32, _P, C.POINTER(TrackerWeights), _P, C.c_int64, _P]),
    "cirs_tracker_backward_last": (C.c_int, [C.POINTER(TrackerCfg), C.POINTER(TrackerWeights), C.POINTER(TrackerState), _P, _P,
                                             _P, _P, _P, _P, _P, C.c_int32, _P, C.POINTER(TrackerWeights), _P, C.c_int64, _P]),
    "cirs_tracker_prefix_states": (C.c_int, [C.POINTER(TrackerCfg), C.POINTER(TrackerWeights), C.POINTER(TrackerState), _P, _P, _P, _P, C.c_int32, _P,
                                            C.c_int64, _P, C.c_int64, _P]),
    "cirs_embedding_scatter_workspace_bytes": (C.c_int64, [C.c_int64]),
    "cirs_embedding_scatter": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, C.c_int64, _P]),
    "cirs_deepfm_forward": (C.c_int, [C.POINTER(DeepFMCfg), C.POINTER(DeepFMWeights), _P, _P, _P, _P, C.c_int32, _P, _P]),
    "cirs_actor_shard_partials": (C.c_int, [C.POINTER(PolicyCfg), C.POINTER(PolicyWeights), _P, C.c_int64, C.c_int32, C.c_uint64, C.c_uint32,
                                            _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, _P]),
    "cirs_actor_merge_shards": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "cirs_gather_rows": (C.c_int, [_P, C.c_int32, _P, C.c_int64, _P, _P]),
    "cirs_gather_fm": (C.c_int, [C.POINTER(DeepFMCfg), C.POINTER(DeepFMWeights), _P, C.c_int64, _P, _P]),
    "cirs_deepfm_sweep_workspace_bytes": (C.c_int64, [C.POINTER(DeepFMCfg), C.c_int32, C.c_int32]),
    "cirs_deepfm_sweep": (C.c_int, [C.POINTER(DeepFMCfg), C.POINTER(DeepFMWeights), _P, C.c_int32, _P, _P, _P, C.c_int32,
                                    _P, _P, C.c_int32, _P, C.c_int64, _P]),
    "cirs_normed_reward": (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    "cirs_ppo_minibatch_dp": (C.c_int, [C.POINTER(PpoCfg), _P, _P, _P, _P, C.c_int64, C.POINTER(PpoBatch), _P, C.c_int32, _P,
                                        C.c_int32, _P, C.c_int32, _P, _P, C.c_int64, C.c_int32, _P]),
    "cirs_ppo_tp_exchange_floats": (C.c_int64, [C.c_int32, C.c_int32]),
    "cirs_ppo_minibatch_tp": (C.c_int, [C.POINTER(PpoCfg), _P, _P, _P, _P, C.c_int64, C.POINTER(PpoBatch), _P, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, _P, _P, _P, _P, C.c_int32, _P, _P, C.c_int64, C.c_int32, _P]),
    "cirs_ppo_shard_stat_floats": (C.c_int32, []),
    "cirs_ppo_shard_norm": (C.c_int, [C.POINTER(PpoCfg), _P, C.c_int64, C.c_int64, _P, _P]),
    "cirs_ppo_shard_adam": (C.c_int, [C.POINTER(PpoCfg), _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int32, _P, _P]),
    "cirs_deepfm_train_param_count": (C.c_int64, [C.POINTER(DeepFMCfg)]),
    "cirs_deepfm_train_workspace_bytes": (C.c_int64, [C.POINTER(DeepFMCfg), C.c_int32]),
    "cirs_deepfm_train_step": (C.c_int, [C.POINTER(DeepFMCfg), _P, _P, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32,
                                         C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                         _P, _P, C.c_int64, _P]),
    "cirs_exposure_history": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, C.c_int32, C.c_double, _P, _P]),
    "cirs_find_negative": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int32, C.c_int64, _P, _P]),
    "cirs_select_items": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_float, _P, C.c_uint64, C.c_uint32,
                                    _P, _P, _P]),
    "cirs_rollout_static": (C.c_int, [C.POINTER(EnvCfg), C.POINTER(EnvTables), C.POINTER(EnvState), _P, C.c_int64, _P, C.POINTER(Traj),
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_uint32, _P, C.c_int32, _P, _P]),
    "cirs_hash_ids": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "cirs_random_permutation": (C.c_int, [C.c_int64, C.c_uint64, C.c_uint64, _P, _P]),
    "cirs_random_permutations": (C.c_int, [C.c_int64, C.c_uint64, C.c_uint64, C.c_int32, _P, _P]),
    "cirs_prof_start": (C.c_int, [C.c_int32, C.c_int32]),
    "cirs_prof_stop": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "cirs_eval_coverage": (C.c_int, [_P, C.c_int64, C.c_int32, _P, _P, _P, _P]),
    "cirs_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_float,
                                 C.c_float, _P, C.c_int32, _P]),
}

_lib = None


class CirsHipError(RuntimeError):
    pass


def lib():
    """Load libcirs_hip.so (built in-tree by `__graft_entry__.build()` / `cirs_hip.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CirsHipError(
            f"{LIB_PATH} not found: the HIP extension is the only implementation of this path (no CPU fallback). "
            "Build it with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950).")
    # torch first: its bundled HIP runtime must be the one this process uses.  Loading libcirs_hip.so before torch pulls in
    # the system libamdhip64 instead, and launches on torch's device pointers then fail ("no ROCm-capable device").
    import torch  # noqa: F401
    handle = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as exc:
            raise CirsHipError(f"libcirs_hip.so does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().cirs_last_error()
        raise CirsHipError(f"{what} failed rc={rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device (or host) pointer of a torch tensor / numpy array as an int, 0 for None."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data
