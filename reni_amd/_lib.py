"""ctypes binding of libreni_hip.so (C ABI in include/reni_hip.h).

The product path has NO CPU fallback: if the HIP library is missing the import of any compute
entry point raises, loudly.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RENI_HIP_LIB") or os.path.join(_HERE, "lib", "libreni_hip.so")  # env override: kernel experiments

RENI_OK = 0
EQ = {"None": 0, None: 0, "SO2": 1, "SO3": 2}
ACT = {None: 0, "None": 0, "none": 0, "tanh": 1, "exp": 2}
DTYPE = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}
LOSS_MSE, LOSS_TEST = 0, 1
NEED_DW, NEED_DZ, WEIGHT_SPARSE, WEIGHT_COMPACT, WEIGHT_COS_CONSTANT = 1, 2, 4, 8, 16
COND_CONCAT, COND_FILM = 0, 1

# every symbol include/reni_hip.h declares (tests check the library exports all of them)
EXPORTS = (
    "reni_last_error", "reni_plan_create", "reni_plan_destroy", "reni_param_count", "reni_in_features",
    "reni_workspace_bytes", "reni_forward", "reni_forward_loss_backward", "reni_backward",
    "reni_forward_loss_backward_rows", "reni_train_step_rows", "reni_train_step_rows_dp", "reni_latent_step_rows", "reni_latent_step_rows_cached", "reni_weight_lists_bytes", "reni_weight_lists_build", "reni_adam_step", "reni_adam_rows_step", "reni_adam_step2", "reni_selftest_layouts", "reni_launch_info", "reni_path_info", "reni_launch_count", "reni_set_grad_ready_event", "reni_profile_enable", "reni_profile_read", "reni_profile_read_kind", "reni_profile_minmax", "reni_probe_tr",
    "reni_film_forward", "reni_film_forward_loss_backward", "reni_film_backward",
    "reni_film_map_param_count", "reni_film_model_forward", "reni_film_model_forward_loss_backward",
    "reni_film_model_backward",
    "reni_envmap_shade_workspace_bytes", "reni_envmap_shade", "reni_envmap_shade_backward",
    "reni_image_workspace_bytes", "reni_unnormalise_srgb", "reni_minmax_normalise",
    "reni_rccl_unique_id", "reni_rccl_comm_create", "reni_rccl_comm_destroy", "reni_allreduce_grads",
)


class reni_desc(Structure):
    _fields_ = [
        ("equivariance", c_int32), ("ndims", c_int32), ("hidden_features", c_int32),
        ("hidden_layers", c_int32), ("out_features", c_int32), ("last_layer_linear", c_int32),
        ("output_activation", c_int32), ("first_omega_0", c_float), ("hidden_omega_0", c_float),
        ("dtype", c_int32), ("conditioning", c_int32), ("mapping_layers", c_int32), ("mapping_features", c_int32),
    ]


class RENILibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load libreni_hip.so (built by reni_amd/csrc/build.sh or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RENILibraryError(
            f"{LIB_PATH} is missing: the RENI HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or reni_amd/csrc/build.sh). "
            "There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    i64x3 = POINTER(c_int64)
    lib.reni_last_error.restype = c_char_p
    lib.reni_plan_create.argtypes = [POINTER(reni_desc), POINTER(c_void_p)]
    lib.reni_plan_create.restype = c_int32
    lib.reni_plan_destroy.argtypes = [c_void_p]
    lib.reni_plan_destroy.restype = None
    lib.reni_param_count.argtypes = [c_void_p]
    lib.reni_param_count.restype = c_int64
    lib.reni_in_features.argtypes = [c_void_p]
    lib.reni_in_features.restype = c_int32
    lib.reni_workspace_bytes.argtypes = [c_void_p, c_int64, c_int64, c_uint32]
    lib.reni_workspace_bytes.restype = c_size_t
    lib.reni_forward.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_void_p]
    lib.reni_forward.restype = c_int32
    lib.reni_forward_loss_backward.argtypes = [
        c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, i64x3, c_void_p, i64x3,
        c_int32, c_float, c_float, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_forward_loss_backward.restype = c_int32
    lib.reni_forward_loss_backward_rows.argtypes = [
        c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, i64x3, c_void_p, i64x3,
        c_int32, c_float, c_float, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_forward_loss_backward_rows.restype = c_int32
    lib.reni_train_step_rows.argtypes = [
        c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, i64x3, c_void_p, i64x3,
        c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int64, c_float,
        POINTER(c_uint32), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_train_step_rows.restype = c_int32
    lib.reni_train_step_rows_dp.argtypes = [
        c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, i64x3, c_void_p, i64x3,
        c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int64, c_float,
        c_void_p, c_int32, POINTER(c_uint32), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_train_step_rows_dp.restype = c_int32
    lib.reni_latent_step_rows.argtypes = [
        c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, i64x3, c_void_p, i64x3,
        c_int32, c_float, c_float, c_uint32, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int64,
        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_latent_step_rows.restype = c_int32
    lib.reni_latent_step_rows_cached.argtypes = [
        c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, i64x3, c_void_p, i64x3,
        c_int32, c_float, c_float, c_uint32, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int64,
        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_latent_step_rows_cached.restype = c_int32
    lib.reni_weight_lists_bytes.argtypes = [c_int64, c_int64]
    lib.reni_weight_lists_bytes.restype = c_size_t
    lib.reni_weight_lists_build.argtypes = [c_int64, c_int64, c_void_p, i64x3, c_uint32, c_void_p, c_size_t, POINTER(c_int32), c_void_p]
    lib.reni_weight_lists_build.restype = c_int32
    lib.reni_adam_step2.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                    c_int64, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int64, c_float,
                                    c_void_p]
    lib.reni_adam_step2.restype = c_int32
    lib.reni_backward.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                  c_uint32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_backward.restype = c_int32
    lib.reni_film_forward.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_film_forward.restype = c_int32
    lib.reni_film_forward_loss_backward.argtypes = [
        c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, i64x3, c_void_p, i64x3,
        c_int32, c_float, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_film_forward_loss_backward.restype = c_int32
    lib.reni_film_backward.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_film_backward.restype = c_int32
    lib.reni_film_map_param_count.argtypes = [c_void_p]
    lib.reni_film_map_param_count.restype = c_int64
    lib.reni_film_model_forward.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_film_model_forward.restype = c_int32
    lib.reni_film_model_forward_loss_backward.argtypes = [
        c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, i64x3, c_void_p, i64x3,
        c_int32, c_float, c_float, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_film_model_forward_loss_backward.restype = c_int32
    lib.reni_film_model_backward.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                             c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_film_model_backward.restype = c_int32
    lib.reni_adam_step.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                                   c_float, c_int64, c_float, c_void_p]
    lib.reni_adam_step.restype = c_int32
    lib.reni_adam_rows_step.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_float,
                                        c_float, c_float, c_float, c_int64, c_float, c_void_p]
    lib.reni_adam_rows_step.restype = c_int32
    lib.reni_selftest_layouts.argtypes = [POINTER(c_int32), c_int32]
    lib.reni_selftest_layouts.restype = c_int32
    lib.reni_launch_info.argtypes = [c_void_p, c_int64, c_int64, POINTER(c_int32)]
    lib.reni_launch_info.restype = c_int32
    lib.reni_path_info.argtypes = [c_void_p, c_int64, c_int64, c_uint32, POINTER(c_int32)]
    lib.reni_path_info.restype = c_int32
    lib.reni_set_grad_ready_event.argtypes = [c_void_p]
    lib.reni_set_grad_ready_event.restype = c_int32
    lib.reni_launch_count.argtypes = [c_int32]
    lib.reni_launch_count.restype = c_int64
    lib.reni_profile_enable.argtypes = [c_int32]
    lib.reni_profile_enable.restype = c_int32
    lib.reni_profile_read.argtypes = [POINTER(ctypes.c_double), POINTER(c_int64), c_int32]
    lib.reni_profile_read.restype = c_int32
    lib.reni_profile_read_kind.argtypes = [c_int32, POINTER(ctypes.c_double), POINTER(c_int64), c_int32]
    lib.reni_profile_read_kind.restype = c_int32
    lib.reni_profile_minmax.argtypes = [c_int32, POINTER(ctypes.c_double), POINTER(ctypes.c_double)]
    lib.reni_profile_minmax.restype = c_int32
    lib.reni_envmap_shade_workspace_bytes.argtypes = [c_int64, c_int64, c_int64]
    lib.reni_envmap_shade_workspace_bytes.restype = c_size_t
    for fn in (lib.reni_envmap_shade, lib.reni_envmap_shade_backward):
        fn.argtypes = [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_float, c_float, c_float, c_void_p, c_int64,
                       c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p]
        fn.restype = c_int32
    lib.reni_image_workspace_bytes.argtypes = [c_int64, c_int64, c_int64]
    lib.reni_image_workspace_bytes.restype = c_size_t
    lib.reni_unnormalise_srgb.argtypes = [c_int64, c_int64, c_int64, c_void_p, POINTER(c_int64), c_int32, ctypes.c_double,
                                          ctypes.c_double, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.reni_unnormalise_srgb.restype = c_int32
    lib.reni_minmax_normalise.argtypes = [c_int64, c_void_p, ctypes.c_double, ctypes.c_double, c_void_p, c_void_p, c_size_t,
                                          c_void_p]
    lib.reni_minmax_normalise.restype = c_int32
    lib.reni_rccl_unique_id.argtypes = [c_void_p]
    lib.reni_rccl_unique_id.restype = c_int32
    lib.reni_rccl_comm_create.argtypes = [c_void_p, c_int32, c_int32, POINTER(c_void_p)]
    lib.reni_rccl_comm_create.restype = c_int32
    lib.reni_rccl_comm_destroy.argtypes = [c_void_p]
    lib.reni_rccl_comm_destroy.restype = c_int32
    lib.reni_allreduce_grads.argtypes = [c_void_p, c_void_p, c_size_t, c_float, c_void_p]
    lib.reni_allreduce_grads.restype = c_int32
    _lib = lib
    return lib


def check(rc: int):
    if rc != RENI_OK:
        msg = load().reni_last_error().decode("utf-8", "replace")
        raise RENILibraryError(f"libreni_hip error {rc}: {msg}")
