"""ctypes binding of libd3il_rollout.so (include/d3il_rollout.h).

This is the reference-side binding a maintainer would add; it contains no
compute.  Loading fails loudly when the library is missing - there is no Python/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

from .model.blob import ModelBlob

_LIB = None


class Buffers(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("stride", C.c_int32), ("obs_dim", C.c_int32), ("action_dim", C.c_int32),
                ("obs", C.c_void_p), ("done", C.c_void_p), ("success", C.c_void_p), ("mode", C.c_void_p),
                ("state", C.c_void_p), ("flags", C.c_void_p), ("step_count", C.c_void_p), ("policy_des", C.c_void_p),
                ("info_f64", C.c_void_p), ("n_info_f64", C.c_int32), ("state_rows", C.c_int32), ("last_reset", C.c_void_p)]


STATE_F64 = 42
STATE_QPOS, STATE_QVEL, STATE_BIAS, STATE_TCP, STATE_IK_Q, STATE_IK_QD = 0, 9, 18, 25, 28, 35
FLAG_MODE_MASK, FLAG_TERMINATED, FLAG_SUCCESS, FLAG_ROD_CONTACT = 0x1FF, 1 << 12, 1 << 13, 1 << 14
FLAG_IK_VALID, FLAG_SOLVER_FAIL, FLAG_MULTI_CONTACT = 1 << 15, 1 << 16, 1 << 17
# Pushing (D3IL_PUSH_STATE_* / D3IL_PFLAG_* in include/d3il_rollout.h)
PUSH_STATE_BOX, PUSH_STATE_WARM, PUSH_STATE_TASK, PUSH_STATE_F64 = 42, 68, 89, 91
PFLAG_FIRST_MASK, PFLAG_MODE_MASK, PFLAG_WARM_VALID, PFLAG_CON_OVERFLOW, PFLAG_OFF_TABLE = 0x7, 0x38, 1 << 6, 1 << 18, 1 << 19
TASK_AVOIDING, TASK_PUSHING, TASK_SORTING, TASK_STACKING, TASK_ALIGNING, TASK_INSERTING = 0, 1, 2, 3, 4, 5
ALIGN_STATE_BOX, ALIGN_STATE_WARM, ALIGN_STATE_TARGET, ALIGN_STATE_F64 = 42, 55, 70, 77
STACK_STATE_BOX, STACK_STATE_WARM, STACK_STATE_F64 = 28, 67, 94
SFLAG_MODE_MASK, SFLAG_WARM_VALID, SFLAG_HAND_NEAR = 0xFF, 1 << 8, 1 << 20
TALLY_ROW, TALLY_ALL = 514, 256
ERCCL = -7
SORT_STATE_BOX, SORT_STATE_WARM, SORT_STATE_TASK, SORT_STATE_F64 = 42, 94, 127, 129
INS_STATE_BOX, INS_STATE_WARM, INS_STATE_TASK, INS_STATE_F64 = 42, 81, 108, 110

EXPORTS = ["d3il_create", "d3il_destroy", "d3il_start", "d3il_reset", "d3il_step", "d3il_get_buffers", "d3il_get_state",
           "d3il_set_state", "d3il_policy_begin", "d3il_policy_action", "d3il_attention_causal_f32", "d3il_layernorm_f32", "d3il_mlp_gelu_residual_f32", "d3il_mlp_ln_gelu_residual_f32", "d3il_linear120_f32", "d3il_mlp_ln_gelu_residual_f16x3", "d3il_linear120_f16x3", "d3il_attn_half_f16x3", "d3il_ddpm_mlp_f32", "d3il_resmlp_f32", "d3il_auto_reset", "d3il_set_tally", "d3il_count_metrics",
           "d3il_rccl_available", "d3il_comm_unique_id", "d3il_comm_init", "d3il_comm_count", "d3il_comm_destroy", "d3il_reduce_metrics", "d3il_set_timing",
           "d3il_last_step_ms", "d3il_timing_stats", "d3il_step_auto_reset", "d3il_random_rollout_step", "d3il_random_rollout_prepare", "d3il_set_option", "d3il_debug_stats", "d3il_debug_wave_stats", "d3il_debug_wave_counts", "d3il_debug_scratch", "d3il_last_error", "d3il_blob_sizeof", "d3il_version"]


class D3ilError(RuntimeError):
    pass


def lib_path() -> str:
    if os.environ.get("D3IL_LIB_PATH"):       # A/B measurements of two builds on the same GPU box
        return os.environ["D3IL_LIB_PATH"]
    name = "libd3il_rollout_stats.so" if os.environ.get("D3IL_STATS_LIB") == "1" else "libd3il_rollout.so"
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), name)


def load():
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise D3ilError("%s is missing: build it with `python -m d3il_amd.build` (hipcc, gfx950). "
                            "There is no CPU fallback for the rollout path." % path)
        L = C.CDLL(path)
        L.d3il_last_error.restype = C.c_char_p
        L.d3il_blob_sizeof.restype = C.c_size_t
        L.d3il_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.d3il_destroy.argtypes = [C.c_void_p]
        L.d3il_start.argtypes = [C.c_void_p, C.c_void_p]
        L.d3il_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.d3il_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.d3il_get_buffers.argtypes = [C.c_void_p, C.POINTER(Buffers)]
        L.d3il_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.d3il_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.d3il_policy_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.d3il_policy_action.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.d3il_count_metrics.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.d3il_attention_causal_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.d3il_layernorm_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_void_p]
        L.d3il_mlp_ln_gelu_residual_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]
        L.d3il_linear120_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
        L.d3il_mlp_ln_gelu_residual_f16x3.argtypes = L.d3il_mlp_ln_gelu_residual_f32.argtypes
        L.d3il_linear120_f16x3.argtypes = L.d3il_linear120_f32.argtypes
        L.d3il_attn_half_f16x3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.d3il_ddpm_mlp_f32.argtypes = [C.c_void_p] * 12 + [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.d3il_resmlp_f32.argtypes = [C.c_void_p] * 8 + [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.d3il_mlp_gelu_residual_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]
        L.d3il_auto_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.d3il_step_auto_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.d3il_random_rollout_step.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.d3il_random_rollout_prepare.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.d3il_timing_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.d3il_set_tally.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.d3il_comm_unique_id.argtypes = [C.c_void_p]
        L.d3il_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.d3il_comm_destroy.argtypes = [C.c_void_p]
        L.d3il_comm_count.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.d3il_rccl_available.restype = C.c_int
        L.d3il_reduce_metrics.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.d3il_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.d3il_last_step_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.d3il_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.d3il_debug_scratch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        if L.d3il_blob_sizeof() != C.sizeof(ModelBlob):
            raise D3ilError("model blob layout mismatch between include/d3il_model_blob.h and d3il_amd/model/blob.py")
        _LIB = L
    return _LIB


def check(rc: int):
    if rc != 0:
        raise D3ilError("libd3il_rollout error %d: %s" % (rc, load().d3il_last_error().decode()))
