"""ctypes binding of librac_hip.so (the C ABI declared in include/rac_hip.h).

The product path has no CPU fallback: if the shared library is missing or a
call fails, this module raises.  Tensors cross the boundary as raw device
pointers (`tensor.data_ptr()`), the stream as `torch.cuda.current_stream().cuda_stream`.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAC_HIP_LIB: another build of the same library (A/B runs of kernel variants, tools/build_variant.sh)
LIB_PATH = os.environ.get("RAC_HIP_LIB") or os.path.join(_HERE, "librac_hip.so")

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvArgs(C.Structure):
    """struct rac_conv_args (include/rac_hip.h)."""
    _fields_ = [
        ("mode", i32), ("B", i32), ("H", i32), ("W", i32), ("ksize", i32), ("Cin", i32), ("Cout", i32),
        ("act", i32), ("split_k", i32), ("accumulate", i32), ("a_split", i32), ("o_split", i32),
        ("slab_stride", i64),
        ("a0", vp), ("a1", vp), ("w", vp), ("out0", vp), ("out1", vp),
        ("bias", vp), ("scale", vp), ("shift", vp), ("stats", vp),
        ("stats_rows", i64), ("a0_up", i32), ("amax_per_image", i32),
    ]


WGRAD_MAX_STEPS = 16
ABI_VERSION = 11  # RAC_ABI_VERSION of include/rac_hip.h this binding was written against


class AbsmaxJob(C.Structure):
    """struct rac_absmax_job (include/rac_hip.h)."""
    _fields_ = [("x", vp), ("n", i64), ("amax", vp), ("block_begin", i64)]


class FragJob(C.Structure):
    """struct rac_frag_job (include/rac_hip.h)."""
    _fields_ = [("w", vp), ("w_amax", vp), ("parts", vp), ("part_stride", i64), ("Cout", i32), ("Cin", i32),
                ("ksize", i32), ("transposed", i32), ("block_begin", i64)]


class AdamFragJob(C.Structure):
    """struct rac_adam_frag_job (include/rac_hip.h)."""
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("scale_slot", vp), ("amax_out", vp), ("parts_fwd", vp),
                ("parts_t", vp), ("part_stride", i64), ("Cout", i32), ("Cin", i32), ("ksize", i32), ("reserved", i32),
                ("block_begin", i64)]


class AdamRange(C.Structure):
    """struct rac_adam_range (include/rac_hip.h)."""
    _fields_ = [("begin4", i64), ("n4", i64), ("block_begin", i64)]


class GradSrc(C.Structure):
    """struct rac_grad_src (include/rac_hip.h): one addend of a gradient map, slabs read through a column window."""
    _fields_ = [("p", vp), ("slab_stride", i64), ("n_slabs", i32), ("row_stride", i32), ("col_off", i32),
                ("reserved", i32)]


class WgradArgs(C.Structure):
    """struct rac_wgrad_args (include/rac_hip.h)."""
    _fields_ = [
        ("B", i32), ("H", i32), ("W", i32), ("ksize", i32), ("Cin", i32), ("Cout", i32), ("a_split", i32),
        ("T", i32), ("nsplit", i32), ("accumulate", i32),
        ("dy", vp * WGRAD_MAX_STEPS), ("x0", vp * WGRAD_MAX_STEPS), ("x1", vp * WGRAD_MAX_STEPS),
        ("dy_amax", vp * WGRAD_MAX_STEPS), ("x0_amax", vp * WGRAD_MAX_STEPS), ("x1_amax", vp * WGRAD_MAX_STEPS),
        ("dw", vp), ("slabs", vp), ("slab_stride", i64), ("x1_zero_steps", i32), ("presplit", i32),
        ("all_ky", i32), ("col_segments", i32),
    ]


# name -> argtypes (return type is always int unless listed in _RET)
_SIGS = {
    "rac_conv2d": [C.POINTER(ConvArgs), vp],
    "rac_absmax": [vp, i64, vp, i64, vp, vp],
    "rac_absmax_rows": [vp, i64, i64, vp, vp],
    "rac_weight_frag_split": [vp, vp, vp, i32, i32, i32, i32, i64, vp],
    "rac_absmax_blocks": [i64],
    "rac_weight_frag_blocks": [i32, i32, i32],
    "rac_absmax_multi": [vp, i32, i64, vp],
    "rac_weight_frag_split_multi": [vp, i32, i64, vp],
    "rac_conv2d_split_supported": [i32, i32, i32, i32, i32, i32],
    "rac_conv2d_fwd_split": [C.POINTER(ConvArgs), vp, vp, i64, i32, vp, vp, vp],
    "rac_conv2d_fwd_split_pool_ok": [C.POINTER(ConvArgs), i32],
    "rac_convlstm_cell_fwd_split": [C.POINTER(ConvArgs), vp, vp, i64, i32, vp, vp, vp, vp, vp],
    "rac_conv2d_wgrad_split": [C.POINTER(WgradArgs), vp],
    "rac_split_steps": [C.POINTER(vp), C.POINTER(vp), i32, i64, C.POINTER(vp), i32, vp],
    "rac_slab_accumulate": [vp, i32, i64, vp, i64, vp],
    "rac_bn_finalize": [vp, i64, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, i32, i32, vp],
    "rac_affine_act": [vp, vp, vp, i32, vp, i64, i32, i32, vp, vp],
    "rac_bn_apply_act": [vp, i64, vp, vp, vp, vp, f32, f32, i32, vp, i32, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp],
    "rac_bn_small_ok": [i64, i32],
    "rac_bn_small_fwd": [vp, i32, i64, vp, vp, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, i64, i32, i32, vp, vp],
    "rac_bn_small_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp],
    "rac_bn_bwd_reduce": [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp],
    "rac_bn_bwd_apply": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp],
    "rac_maxpool2_fwd": [vp, vp, i32, i32, i32, i32, vp],
    "rac_maxpool2_bwd": [vp, vp, vp, i32, i32, i32, i32, vp],
    "rac_upsample2_fwd": [vp, vp, i32, i32, i32, i32, vp],
    "rac_upsample2_bwd": [vp, vp, i32, i32, i32, i32, vp],
    "rac_tilecat_fwd": [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, vp, i32, i32, vp, i32, vp],
    "rac_pad_rows": [vp, i32, vp, i32, i64, vp],
    "rac_unpad_add": [vp, i32, vp, i32, i64, vp],
    "rac_slice_channels": [vp, i32, i32, i32, vp, i64, vp],
    "rac_colsum_acc": [vp, vp, vp, i64, i32, vp],
    "rac_colsum_blocks": [i64, i32],
    "rac_slab_reduce": [vp, i32, i64, vp, vp, i64, i32, vp, vp],
    "rac_slab_reduce_stats": [vp, i32, i64, vp, vp, i64, i32, i32, vp, vp],
    "rac_slab_reduce2": [vp, i32, i64, vp, vp, vp, i64, i32, i32, vp, vp, vp],
    "rac_cat2_channels": [vp, i32, vp, i32, vp, i64, vp, vp],
    "rac_colsum_steps": [C.POINTER(vp), i32, vp, vp, i64, i32, vp],
    "rac_col_stats": [vp, vp, i64, i32, i32, vp],
    "rac_act_bwd": [vp, vp, i32, vp, i64, vp],
    "rac_lstm_cell_fwd": [vp, i32, i64, vp, vp, vp, vp, vp, i64, i32, vp],
    "rac_lstm_cell_bwd": [vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp],
    "rac_groupnorm_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "rac_norm_lstm_cell_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "rac_groupnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "rac_norm_lstm_cell_bwd": [vp] * 25 + [i32, i32, i32, vp],
    "rac_lstm_out_fwd": [vp, vp, vp, i64, i32, vp],
    "rac_lstm_out_bwd": [vp, vp, vp, vp, vp, i64, i32, vp],
    "rac_lstm_core_bwd": [vp, vp, vp, vp, vp, vp, i64, i32, vp],
    "rac_grad_sum": [C.POINTER(GradSrc), i32, vp, i64, i32, vp, vp],
    "rac_lstm_cell_bwd_srcs": [C.POINTER(GradSrc), i32, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp],
    "rac_reparam_head_bwd": [C.POINTER(GradSrc), i32, vp, vp, vp, vp, vp, i64, i32, vp, vp],
    "rac_reparam_fwd": [vp, vp, vp, vp, i64, vp],
    "rac_reparam_bwd": [vp, vp, vp, vp, i64, vp],
    "rac_pack_input": [vp, vp, vp, i32, i32, vp, i32, i32, vp],
    "rac_first_layer_fwd": [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp],
    "rac_first_layer_fwd_split": [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp],
    "rac_head_fwd": [vp, vp, vp, vp, i32, i32, i32, vp],
    "rac_head_dgrad": [vp, vp, vp, i32, i32, i32, vp],
    "rac_head_fwd_split": [vp, vp, i32, vp, vp, vp, i32, i32, i32, vp],
    "rac_thin_wgrad": [vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp],
    "rac_unpack_grad": [vp, i32, vp, vp, i32, i32, vp],
    "rac_zero_region": [vp, vp, vp, i32, i32, vp],
    "rac_composite_fwd": [vp, vp, vp, i32, i32, vp],
    "rac_composite_bwd": [vp, vp, vp, vp, vp, i32, i32, vp],
    "rac_recon_loss_fwd": [i32, vp, vp, vp, f32, vp, vp, vp, i32, i32, vp],
    "rac_recon_loss_bwd": [i32, vp, vp, vp, f32, vp, vp, vp, vp, i32, i32, vp],
    "rac_kl_fwd": [vp, vp, vp, vp, i64, i32, vp, vp, vp],
    "rac_kl_bwd": [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp],
    "rac_psnr_ssim": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "rac_cem_step_tail": [vp, vp, vp, vp, vp, vp, i32, f32, i32, vp, vp, i32, i32, vp],
    "rac_cem_robot_inputs": [vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, f32, f32, f32, f32, vp, vp, i32, i32, i32, i32, i32, vp],
    "rac_adam_step": [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, vp],
    "rac_adam_frag_multi": [vp, i32, i64, f32, f32, f32, f32, i32, vp],
    "rac_adam_frag_multi_bounded": [vp, i32, i64, i32, f32, f32, f32, f32, i32, vp],
    "rac_amax_bound": [vp, vp, vp, i32, f32, vp],
    "rac_adam_ranges": [vp, vp, vp, vp, vp, i32, i64, f32, f32, f32, f32, i32, vp],
    "rac_version": [],
    "rac_device_arch": [],
    "rac_last_error": [],
}
_RET = {"rac_device_arch": C.c_char_p, "rac_last_error": C.c_char_p, "rac_absmax_blocks": i64, "rac_colsum_blocks": i64,
        "rac_weight_frag_blocks": i64}
EXPORTS = tuple(_SIGS)

_lib = None


class RacError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen librac_hip.so and type every export; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RacError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C robot_aware_control_amd/csrc` (no CPU fallback exists)")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RET.get(name, C.c_int)
    if lib.rac_version() != ABI_VERSION:
        raise RacError(f"{LIB_PATH} is ABI version {lib.rac_version()}, this binding needs {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """The current HIP stream's handle.  `torch.cuda.current_stream().cuda_stream` builds a Stream object through several
    Python layers (8.4 us per call, measured: 1 000+ launches per step of the small-batch / scheduled-sampling paths made the
    HOST the bound, tools/host_profile.py); torch's raw accessors return the same handle in ~0.3 us."""
    if _raw_stream is not None and _get_device is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int | None:
    """Device pointer of a tensor (None passes NULL)."""
    if t is None:
        return None
    return t.data_ptr()


_FN = {}


def call(name: str, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise RacError(f"{name} failed ({rc}): {load().rac_last_error().decode()}")
