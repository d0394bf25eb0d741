"""ctypes binding of libdeepq_hip.so (include/deepq_hip.h).

The HIP library IS the product: if it cannot be loaded this module raises, and every compute
entry point raises when no GPU is visible.  There is no CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DQ_LIB_PATH: development aid (instrumented builds of the SAME library, tools/build_stamps.sh); there is still no CPU fallback
LIB_PATH = os.environ.get("DQ_LIB_PATH") or os.path.join(_HERE, "lib", "libdeepq_hip.so")

DQ_MODEL_X, DQ_MODEL_DP, DQ_MODEL_IIDXZ = 0, 1, 2
STREAM_ENV, STREAM_POLICY, STREAM_REPLAY, STREAM_DROPOUT, STREAM_INIT = range(5)


DQ_ERR_RANGE = -6


class DeepQError(RuntimeError):
    """status: the dq_status code of include/deepq_hip.h (None for errors raised on the host side)."""

    def __init__(self, message, status=None):
        super().__init__(message)
        self.status = status


class EnvCfg(ctypes.Structure):
    _fields_ = [("d", ctypes.c_int32), ("error_model", ctypes.c_int32), ("use_Y", ctypes.c_int32),
                ("volume_depth", ctypes.c_int32), ("n_envs", ctypes.c_int32), ("env_id_base", ctypes.c_uint32),
                ("seed", ctypes.c_uint32 * 2)]


class EnvInfo(ctypes.Structure):
    _fields_ = [("num_actions", ctypes.c_int32), ("n_action_layers", ctypes.c_int32), ("identity_index", ctypes.c_int32),
                ("obs_c", ctypes.c_int32), ("obs_h", ctypes.c_int32), ("obs_w", ctypes.c_int32),
                ("n_stab", ctypes.c_int32), ("state_words", ctypes.c_int32)]


class QNetCfg(ctypes.Structure):
    _fields_ = [("in_c", ctypes.c_int32), ("in_h", ctypes.c_int32), ("in_w", ctypes.c_int32), ("n_conv", ctypes.c_int32),
                ("conv", (ctypes.c_int32 * 3) * 4), ("n_ff", ctypes.c_int32), ("ff_units", ctypes.c_int32 * 4),
                ("ff_dropout", ctypes.c_float * 4), ("n_actions", ctypes.c_int32), ("dueling", ctypes.c_int32),
                ("max_batch", ctypes.c_int32)]


class QNetJob(ctypes.Structure):
    """dq_qnet_job (include/deepq_hip.h)."""
    _fields_ = [("params_dev", ctypes.c_void_p), ("obs_dev", ctypes.c_void_p), ("index_dev", ctypes.c_void_p),
                ("index_off", ctypes.c_int32), ("index_mod", ctypes.c_int32), ("batch", ctypes.c_int32), ("training", ctypes.c_int32),
                ("seed", ctypes.c_uint32 * 2), ("t", ctypes.c_uint64), ("sample_base", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("q_dev", ctypes.c_void_p), ("packed_dev", ctypes.c_void_p)]


class SampleJob(ctypes.Structure):
    """dq_sample_job (include/deepq_hip.h)."""
    _fields_ = [("terminal_ring_dev", ctypes.c_void_p), ("n_slots", ctypes.c_int32), ("head_slot", ctypes.c_int32),
                ("filled_slots", ctypes.c_int32), ("batch", ctypes.c_int32), ("seed", ctypes.c_uint32 * 2), ("t", ctypes.c_uint64),
                ("sample_base", ctypes.c_uint32), ("index_dev", ctypes.c_void_p)]


class TdJob(ctypes.Structure):
    """dq_td_job (include/deepq_hip.h)."""
    _fields_ = [("q_online_s1_dev", ctypes.c_void_p), ("q_target_s1_dev", ctypes.c_void_p), ("q_s0_dev", ctypes.c_void_p),
                ("reward_dev", ctypes.c_void_p), ("terminal_dev", ctypes.c_void_p), ("action_dev", ctypes.c_void_p),
                ("index_dev", ctypes.c_void_p), ("gamma", ctypes.c_double), ("grad_scale", ctypes.c_double), ("batch", ctypes.c_int32),
                ("n_actions", ctypes.c_int32), ("y_dev", ctypes.c_void_p), ("dq_dev", ctypes.c_void_p), ("metrics_dev", ctypes.c_void_p),
                ("done_dev", ctypes.c_void_p), ("was_reset_dev", ctypes.c_void_p), ("lifetime_dev", ctypes.c_void_p),
                ("step_reward_dev", ctypes.c_void_p), ("n", ctypes.c_int32), ("stats_dev", ctypes.c_void_p), ("auto_scale", ctypes.c_int32)]


class EnvStepJob(ctypes.Structure):
    """dq_env_step_job (include/deepq_hip.h)."""
    _fields_ = [("q_dev", ctypes.c_void_p), ("eps", ctypes.c_double), ("masked_greedy", ctypes.c_int32), ("seed", ctypes.c_uint32 * 2),
                ("t", ctypes.c_uint64), ("action_dev", ctypes.c_void_p), ("auto_reset", ctypes.c_int32), ("obs_dev", ctypes.c_void_p),
                ("reward_dev", ctypes.c_void_p), ("done_dev", ctypes.c_void_p), ("legal_dev", ctypes.c_void_p), ("lifetime_dev", ctypes.c_void_p),
                ("was_reset_dev", ctypes.c_void_p), ("sample", ctypes.POINTER(SampleJob)), ("stats_dev", ctypes.c_void_p)]


class EnvRing(ctypes.Structure):
    """dq_env_ring (include/deepq_hip.h)."""
    _fields_ = [("action_ring_dev", ctypes.c_void_p), ("reward_ring_dev", ctypes.c_void_p), ("done_ring_dev", ctypes.c_void_p), ("obs_ring_dev", ctypes.c_void_p),
                ("patch_ring_dev", ctypes.c_void_p), ("patch_stride_words", ctypes.c_int32), ("n_slots", ctypes.c_int32), ("slot0", ctypes.c_int32)]


_vp, _i, _u32, _u64, _dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_double
_sz = ctypes.c_size_t
_seedp = ctypes.POINTER(ctypes.c_uint32)

# name -> (restype, argtypes); must list every symbol include/deepq_hip.h declares
SIGNATURES = {
    "dq_version": (_i, []),
    "dq_build_digest": (ctypes.c_char_p, []),
    "dq_last_error": (ctypes.c_char_p, []),
    "dq_device_count": (_i, []),
    "dq_env_create": (_i, [ctypes.POINTER(EnvCfg), ctypes.POINTER(_vp)]),
    "dq_env_destroy": (None, [_vp]),
    "dq_env_get_info": (_i, [_vp, ctypes.POINTER(EnvInfo)]),
    "dq_env_set_rates": (_i, [_vp, _dbl, _dbl]),
    "dq_env_build_referee": (_i, [_vp, _vp]),
    "dq_env_build_referee_ml": (_i, [_vp, _dbl, _vp]),
    "dq_env_set_referee": (_i, [_vp, _vp, _vp]),
    "dq_env_set_referee_joint": (_i, [_vp, _vp]),
    "dq_env_set_referee_mlp": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_int32), _vp]),
    "dq_env_referee_classes": (_i, [_vp, _vp, _vp, _vp]),
    "dq_env_get_referee": (_i, [_vp, _vp, _vp, ctypes.c_size_t]),
    "dq_env_reset": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "dq_env_step": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dq_env_act_step": (_i, [_vp, _vp, _dbl, _i, _seedp, _u64, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dq_env_act_step_sample": (_i, [_vp, _vp, _dbl, _i, _seedp, _u64, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(SampleJob), _vp]),
    "dq_env_act_steps": (_i, [_vp, _i, _seedp, _u64, ctypes.POINTER(EnvRing), _i, _vp, _vp, _vp, _vp]),
    "dq_env_patch_output": (_i, [_vp, _vp, _i]),
    "dq_env_export_state": (_i, [_vp, _vp, _vp]),
    "dq_env_import_state": (_i, [_vp, _vp, _vp]),
    "dq_env_get_tables": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "dq_match_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "dq_match_destroy": (None, [_vp]),
    "dq_match_info": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "dq_match_get_tables": (_i, [_vp, _i, _vp, _vp, ctypes.POINTER(_i)]),
    "dq_match_decode": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "dq_envb_create": (_i, [ctypes.POINTER(EnvCfg), ctypes.POINTER(_vp)]),
    "dq_envb_destroy": (None, [_vp]),
    "dq_envb_get_info": (_i, [_vp, ctypes.POINTER(EnvInfo), ctypes.POINTER(_i)]),
    "dq_envb_set_rates": (_i, [_vp, _dbl, _dbl]),
    "dq_envb_reset": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "dq_envb_step": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dq_envb_act_step": (_i, [_vp, _vp, _dbl, _i, _seedp, _u64, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dq_envb_export_state": (_i, [_vp, _vp, _vp]),
    "dq_policy_select_wide": (_i, [_vp, _vp, _i, _i, _i, _dbl, _i, ctypes.POINTER(_u32), _u32, _u64, _vp, _vp]),
    "dq_policy_select": (_i, [_vp, _vp, _i, _i, _dbl, _i, ctypes.POINTER(_u32), _u32, _u64, _vp, _vp]),
    "dq_qnet_create": (_i, [ctypes.POINTER(QNetCfg), ctypes.POINTER(_vp)]),
    "dq_qnet_destroy": (None, [_vp]),
    "dq_qnet_param_count": (_sz, [_vp]),
    "dq_qnet_num_layers": (_i, [_vp]),
    "dq_qnet_layer_info": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                ctypes.POINTER(ctypes.c_int32 * 4), ctypes.POINTER(ctypes.c_int32)]),
    "dq_qnet_set_fused": (_i, [_vp, _i]),
    "dq_qnet_set_kernel_forms": (_i, [_vp, _i, _i, _i]),
    "dq_qnet_set_grad_scale": (_i, [_vp, _dbl]),
    "dq_struct_size": (ctypes.c_long, [_i]),
    "dq_qnet_mark_conv_backward": (_i, [_vp, _vp]),
    "dq_qnet_range_check": (_i, [_vp, _vp]),
    "dq_qnet_range_discarded": (_i, [_vp, ctypes.POINTER(ctypes.c_uint), _vp]),
    "dq_qnet_fused_supported": (_i, [_vp]),
    "dq_qnet_set_patch_input": (_i, [_vp, _i, _i]),
    "dq_qnet_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _seedp, _u64, _u32, _vp, _vp]),
    "dq_qnet_forward_multi": (_i, [_vp, _i, ctypes.POINTER(QNetJob), _vp]),
    "dq_qnet_packed_bytes": (_sz, [_vp]),
    "dq_qnet_pack": (_i, [_vp, _vp, _vp, _vp]),
    "dq_qnet_backward": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "dq_qnet_backward_phase": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "dq_qnet_conv_param_count": (_sz, [_vp]),
    "dq_qnet_backward_adam": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _u64, _vp]),
    "dq_qnet_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _u64, _vp]),
    "dq_qnet_td_backward_adam": (_i, [_vp, _vp, ctypes.POINTER(TdJob), _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _u64, _vp]),
    "dq_qnet_td_backward_adam_env": (_i, [_vp, _vp, ctypes.POINTER(TdJob), _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _u64, _vp,
                                          ctypes.POINTER(EnvStepJob), _vp]),
    "dq_qnet_td_backward_phase0": (_i, [_vp, _vp, ctypes.POINTER(TdJob), _vp, _vp]),
    "dq_qnet_td_backward_phase0_env": (_i, [_vp, _vp, ctypes.POINTER(TdJob), _vp, _vp, ctypes.POINTER(EnvStepJob), _vp]),
    "dq_replay_sample": (_i, [_vp, _i, _i, _i, _i, _i, _seedp, _u64, _u32, _vp, _vp]),
    "dq_replay_sample_multi": (_i, [_vp, _i, _i, _i, _i, _i, _seedp, _u64, _i, _u32, _vp, _vp]),
    "dq_td_target": (_i, [_vp, _vp, _vp, _vp, _vp, _dbl, _i, _i, _vp, _vp]),
    "dq_td_loss_grad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _dbl, _vp, _vp, _vp]),
    "dq_episode_stats": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "dq_test_bookkeeping": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "dq_td_update": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _i, _i, _dbl, _vp, _vp, _vp, _vp]),
    "dq_td_update_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _i, _i, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "dq_td_metrics": (_i, [_vp, _i, _vp]),
    "dq_post_step": (_i, [_vp, _i, _i, _i, _i, _i, _seedp, _u64, _u32, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "dq_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _dbl, _dbl, _dbl, _dbl, _u64, _vp]),
    "dq_prof_kernel_count": (_i, []),
    "dq_prof_kernel_name": (ctypes.c_char_p, [_i]),
    "dq_prof_kernel_symbol": (ctypes.c_char_p, [_i]),
    "dq_prof_arm": (_i, [_i, _i]),
    "dq_prof_stride": (_i, [_i]),
    "dq_prof_collect": (_i, [ctypes.POINTER(_i), ctypes.POINTER(_dbl)]),
    "dq_prof_collect_spread": (_i, [ctypes.POINTER(_i), ctypes.POINTER(_dbl), ctypes.POINTER(_dbl), ctypes.POINTER(_dbl)]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises DeepQError if the .so is absent)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DeepQError(
                f"{LIB_PATH} not found: build it with `python deepq-decoding_amd/build.py` "
                "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
        try:
            # torch's HIP runtime is initialised BEFORE this library is loaded: loaded first (e.g. build() and smoke() in one process),
            # the library's hipGetDeviceCount then saw no device on the GPU box
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise DeepQError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the build is stale
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise DeepQError(f"libdeepq_hip error {status}: {lib().dq_last_error().decode()}", status)


def require_gpu():
    if lib().dq_device_count() < 1:
        raise DeepQError("no HIP device visible: the DeepQ-Decoding hot path runs only on an MI355X (no CPU fallback)")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def current_stream(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
