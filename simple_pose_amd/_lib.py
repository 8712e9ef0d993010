"""ctypes binding of libsimple_pose_hip.so (C ABI: include/simple_pose_hip.h).

There is NO fallback: if the HIP library is missing or a call fails, this module raises.  The product path never
routes through torch ops or the oracle.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

from .build import LIB_PATH as _DEFAULT_LIB_PATH

# SIMPLE_POSE_HIP_LIB lets a diagnostic build of the same library (tools/) be loaded instead; never a fallback path.
LIB_PATH = os.environ.get("SIMPLE_POSE_HIP_LIB", _DEFAULT_LIB_PATH)

SP_CONV_RELU = 0x1
SP_CONV_OUT_NCHW = 0x2
SP_CONV_PIXEL_SHUFFLE = 0x4
SP_CONV_BF16 = 0x8
SP_CONV_OUT_F32 = 0x10
SP_CONV_BN_Y_MASK = 0x20
CONV_TILES = ((128, 128), (64, 128), (128, 64), (64, 64), (256, 64), (128, 32))
ABI_VERSION = 35
SP_CONV_KERNEL_IGEMM, SP_CONV_KERNEL_RING, SP_CONV_KERNEL_PW, SP_CONV_KERNEL_RING_LW, SP_CONV_KERNEL_RING_LW4 = 0, 1, 2, 3, 4
RING_LW4_TILES = ((192, 128), (128, 128), (96, 128), (256, 128), (128, 256), (96, 256), (64, 128))   # kernel = SP_CONV_KERNEL_RING_LW4 (four MFMA waves + four loader waves)
RING_LW_TILES = ((256, 128), (128, 256), (256, 64), (128, 128), (192, 128))   # kernel = SP_CONV_KERNEL_RING_LW (bf16; the ring with loader waves)
RING_TILES = ((256, 256), (256, 128), (128, 256), (256, 64), (128, 128), (192, 128), (192, 256))   # kernel = SP_CONV_KERNEL_RING (bf16)


class HipLibraryError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    """Mirror of `sp_conv_desc` (field order is ABI)."""
    _fields_ = [(n, c_int32) for n in (
        "batch", "in_h", "in_w", "c_in", "grid_h", "grid_w", "c_out", "n_pad", "taps_h", "taps_w", "k_pad", "stride",
        "dy0", "dy_step", "dx0", "dx_step", "out_h", "out_w", "out_c", "oy_mul", "oy_add", "ox_mul", "ox_add",
        "phases_y", "phases_x")] + [("flags", c_uint32), ("tile_m", c_int32), ("tile_n", c_int32), ("stride_x", c_int32), ("kernel", c_int32), ("c_in_group", c_int32)]


class WgradJob(ctypes.Structure):
    """Mirror of `sp_wgrad_job` (one layer of sp_conv2d_wgrad_batched)."""
    _fields_ = [("desc", ConvDesc), ("g", c_void_p), ("a", c_void_p), ("dw", c_void_p), ("dst_stride_n", c_int64), ("dst_stride_c", c_int64),
                ("g_channels", c_int32), ("n_valid", c_int32), ("c_valid", c_int32), ("kw_valid", c_int32)]


# every symbol include/simple_pose_hip.h declares: name -> (restype, argtypes)
_P = c_void_p
SYMBOLS = {
    "sp_abi_version": (c_int, []),
    "sp_last_error": (ctypes.c_char_p, []),
    "sp_nchw_to_nhwc4": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_conv2d_fwd": (c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "sp_conv2d_ring_ok": (c_int, [ctypes.POINTER(ConvDesc)]),
    "sp_conv2d_kernel_name": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_int, ctypes.c_char_p, c_int]),
    "sp_conv2d_default_tile": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "sp_maxpool3x3s2_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_pixel_shuffle2_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_global_avg_pool_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "sp_se_gate_add_relu_nhwc": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "sp_upsample_add_nhwc": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "sp_nchw_to_nhwc8_bf16": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_maxpool3x3s2_nhwc_bf16": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_pixel_shuffle2_nhwc_bf16": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_upsample_add_nhwc_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "sp_global_avg_pool_nhwc_bf16": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "sp_se_gate_add_relu_nhwc_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "sp_heat_map_to_axis": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "sp_decode_gauss_taylor": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "sp_decode_basic": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "sp_heat_map_acc": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P]),
    "sp_u8hwc_bgr_to_nchw_f32": (c_int, [_P, _P, c_int, c_int, c_int, ctypes.POINTER(c_float), _P]),
    "sp_encode_gauss_refine": (c_int, [_P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P]),
    "sp_encode_gauss_basic": (c_int, [_P, c_int, c_int, c_int, c_int, c_float, c_int, _P, _P, _P]),
    "sp_bn_train_stats_nhwc": (c_int, [_P, c_int, c_int64, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "sp_bn_apply_maxpool_nhwc": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_bn_maxpool_bwd_nhwc": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "sp_bn_apply_nhwc": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, _P, _P]),
    "sp_bn_train_partial_nhwc": (c_int, [_P, c_int, c_int64, c_int, _P, _P, _P]),
    "sp_bn_train_finalize": (c_int, [_P, c_int64, c_int, c_float, c_float, _P, _P, _P, _P, _P]),
    "sp_bn_apply_sums_nhwc": (c_int, [_P, c_int, _P, c_int64, c_float, c_float, _P, _P, _P, _P, c_int64, c_int, c_int, _P, _P, _P, _P, _P]),
    "sp_bn_bwd_sums_from_conv2": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "sp_bn_train_bwd_reduce_nhwc": (c_int, [_P, c_int, _P, _P, _P, _P, c_int64, c_int, _P, _P, _P, _P]),
    "sp_bn_train_bwd_apply_nhwc": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int, _P, _P, c_int, _P]),
    "sp_bn_train_bwd_nhwc": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int64, c_int, _P, _P, _P, _P, c_int, _P, _P]),
    "sp_channel_sum_nchw": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "sp_channel_sum_nhwc": (c_int, [_P, c_int64, c_int, _P, _P, _P]),
    "sp_maxpool3x3s2_bwd_nhwc": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_adam_step": (c_int, [_P, _P, _P, _P, c_int64, c_double, c_double, c_double, c_double, c_int, c_float, _P]),
    "sp_upsample_add_bwd_nhwc": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P]),
    "sp_se_gate_bwd_reduce": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, _P, _P]),
    "sp_se_sigmoid_bwd": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, _P]),
    "sp_relu_bwd_rows": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, _P]),
    "sp_se_gate_bwd_apply": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "sp_adam_set_scalars": (c_int, [c_double, c_double, c_double, c_double, c_int, c_float, _P, _P]),
    "sp_adam_step_dev": (c_int, [_P, _P, _P, _P, c_int64, _P, _P]),
    "sp_upsample_add_n_nhwc": (c_int, [_P, c_int, c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(c_int32), _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "sp_comm_available": (c_int, []),
    "sp_comm_unique_id": (c_int, [_P]),
    "sp_comm_create": (c_int, [_P, c_int, c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "sp_comm_allreduce_sum_f32": (c_int, [_P, _P, c_int64, _P]),
    "sp_comm_allreduce_sum_f64": (c_int, [_P, _P, c_int64, _P]),
    "sp_comm_info": (c_int, [_P, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "sp_comm_destroy": (c_int, [_P]),
    "sp_conv2d_wgrad": (c_int, [ctypes.POINTER(ConvDesc), _P, c_int, _P, c_int, c_int, c_int, c_int64, c_int64, _P, _P, c_int64, _P]),
    "sp_conv2d_wgrad_workspace": (c_int, [ctypes.POINTER(WgradJob), c_int, ctypes.POINTER(c_int64)]),
    "sp_conv2d_wgrad_batched": (c_int, [ctypes.POINTER(WgradJob), c_int, _P, c_int64, _P]),
    "sp_stream_delay_us": (c_int, [c_double, _P]),
    "sp_permute4_f32": (c_int, [_P, _P, c_int, ctypes.POINTER(c_int32), ctypes.POINTER(c_int64), ctypes.POINTER(c_int32), c_int64, c_int64, _P]),
    "sp_permute4_batched": (c_int, [_P, _P, c_int, c_int, _P]),
    "sp_permute4_batched_tiled": (c_int, [_P, _P, c_int, c_int, _P]),
    "sp_pose_score": (c_int, [_P, c_int, c_int, _P, _P]),
    "sp_pose_rescore": (c_int, [_P, _P, c_int, c_int, c_double, _P, _P, _P]),
    "sp_oks_nms": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_double, c_double, _P, _P, _P]),
    "sp_pixel_unshuffle2_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_pixel_unshuffle2_nhwc_bf16": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_maxpool3x3s2_idx_nhwc": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_maxpool3x3s2_bwd_idx_nhwc": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_nchw_to_nhwc_pad": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "sp_warp_affine_u8c3": (c_int, [_P, c_int, c_int, _P, c_int, _P, c_int, c_int, _P]),
    "sp_conv2d_bn_stats_rows": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_int)]),
    "sp_conv2d_fwd_bn_stats": (c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, c_int, _P]),
    "sp_conv2d_fwd_bn_stats_abn": (c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "sp_bn_train_stats_from_conv": (c_int, [_P, _P, c_int, c_int, c_int64, c_int, c_float, c_float, _P, _P, _P, _P, _P]),
    "sp_conv2d_dgrad_bn_bwd_stats": (c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "sp_conv2d_dgrad_bn_bwd_stats2": (c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "sp_conv2d_dgrad_bn_bwd_stats_macc": (c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "sp_conv2d_dgrad_phases": (c_int, [ctypes.POINTER(ConvDesc), c_int, _P, ctypes.POINTER(_P), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "sp_bn_sums_from_conv": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    "sp_bn_fold_apply_nhwc": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int64, c_float, c_float, _P, _P, _P, _P, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "sp_bn_fold_bwd_apply_nhwc": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, c_int64, c_int64, c_int, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "sp_bn_bwd_sums_from_conv": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "sp_bn_bwd_sums_from_conv_pair": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "sp_u8hwc_bgr_to_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "sp_nchw_to_nhwc4_bf16": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_conv3x3_direct_ok": (c_int, [ctypes.POINTER(ConvDesc)]),
    "sp_conv3x3_direct": (c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "sp_basic_block_c32_ok": (c_int, [ctypes.POINTER(ConvDesc)]),
    "sp_basic_block_c32": (c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sp_basic_block_c64_ok": (c_int, [ctypes.POINTER(ConvDesc)]),
    "sp_basic_block_c64": (c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sp_dual_pw_bf16_ok": (c_int, [c_int64, c_int, c_int, c_int]),
    "sp_dual_pw_f32_ok": (c_int, [c_int64, c_int, c_int, c_int]),
    "sp_dual_pw_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, _P]),
    "sp_dual_pw_bf16": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, _P]),
    "sp_bottleneck_c64_ok": (c_int, [ctypes.POINTER(ConvDesc)]),
    "sp_conv2d_pw_ok": (c_int, [ctypes.POINTER(ConvDesc)]),
    "sp_hrnet_stem_ok": (c_int, [c_int, c_int, c_int]),
    "sp_hrnet_stem": (c_int, [_P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "sp_hrnet_transition1_ok": (c_int, [c_int, c_int, c_int]),
    "sp_hrnet_transition1": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sp_stem7_pool_ok": (c_int, [c_int, c_int, c_int]),
    "sp_stem7_pool": (c_int, [_P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_stem7_pool_u8": (c_int, [_P, ctypes.POINTER(ctypes.c_float), _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "sp_bottleneck_c64": (c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sp_masked_mse": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "sp_conv_packed_dims": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "sp_pack_conv_weights": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "sp_pack_conv_weights_grouped": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "sp_pack_conv_weights_grouped_taps": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "sp_conv2d_wgrad_grouped_workspace": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int64)]),
    "sp_conv2d_wgrad_grouped": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int64, _P]),
    "sp_pack_deconv_k4s2p1": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P]),
    "sp_fold_bn": (c_int, [_P, _P, _P, _P, c_int, c_float, c_int, _P, _P, _P]),
}

_lib = None


def lib():
    """Load the library (once).  Raises HipLibraryError if it has not been built - no silent fallback."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} is missing: build it with `python -m simple_pose_amd.build` (hipcc, gfx950). "
                "simple_pose_amd has no CPU/torch fallback.")
        # PyTorch owns the device memory and the streams we launch on, so the process must run ONE HIP runtime: torch's
        # bundled libamdhip64 (same SONAME as /opt/rocm's).  Importing torch first makes the loader bind our library's
        # libamdhip64.so.7 dependency to that copy; loaded the other way round torch ends up on /opt/rocm's runtime next
        # to its own bundled HSA/comgr and reports "no ROCm-capable device".
        import torch  # noqa: F401

        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if handle.sp_abi_version() != ABI_VERSION:
            raise HipLibraryError(f"ABI mismatch: library {handle.sp_abi_version()} vs binding {ABI_VERSION}")
        _lib = handle
    return _lib


def conv_kernel_name(desc, has_residual: bool = False, variant: int = 0) -> str:
    """Kernel instantiation a launch of `desc` resolves to (sp_conv2d_kernel_name: the library's own dispatch names it)."""
    buf = ctypes.create_string_buffer(256)
    check(lib().sp_conv2d_kernel_name(desc, int(has_residual), variant, buf, 256), "sp_conv2d_kernel_name")
    return buf.value.decode()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().sp_last_error().decode("utf-8", "replace")
        raise HipLibraryError(f"{what or 'libsimple_pose_hip'} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None) - a plain int: every pointer parameter is declared c_void_p, ctypes converts."""
    return None if t is None else t.data_ptr()


# The train step issues ~350 launches per step from Python; `torch.cuda.current_stream()` costs ~5 us a call (device index lookups,
# `is_available`, an os.environ read) and was a sixth of the step's host time.  While a PoseTrainer tape runs it pins the handle of the
# stream it is issuing to here (`pin_stream`); everything else asks torch as before.
# The pin is per THREAD and per DEVICE: a data-loader thread that encodes targets, a decode on another GPU or a second trainer never see the
# trainer's handle (round-4 advisor finding: a process-global pin handed it to every caller).
import threading

_pin = threading.local()


def _device_index(device) -> int:
    """Index of a torch.device / int / None (None: the current device)."""
    if device is None:
        import torch
        return torch.cuda.current_device()
    if isinstance(device, int):
        return device
    idx = getattr(device, "index", None)
    if idx is None:
        import torch
        return torch.cuda.current_device()
    return idx


def pin_stream(pin):
    """Make `current_stream(device)` on THIS thread return a pinned handle for that device.  `pin`: (hipStream_t as c_void_p / int, device
    index) or None (ask torch again).  Returns the previous pin (hand it back to restore)."""
    prev = getattr(_pin, "value", None)
    _pin.value = pin
    return prev


def current_stream(device=None):
    """torch's current stream of `device` (a tensor's .device; default: the current device) as a hipStream_t."""
    pin = getattr(_pin, "value", None)
    if pin is not None and (device is None or _device_index(device) == pin[1]):
        return pin[0]
    import torch

    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def same_device(*tensors):
    """The one GPU every given tensor lives on (None entries skipped); raises when operands are spread over devices - the kernels
    take raw pointers, a foreign pointer would fault."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise HipLibraryError(f"operands on different devices: {dev} and {t.device}")
    return dev


def require_cuda_f32(t, name: str):
    import torch

    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise HipLibraryError(f"{name}: tensor is on {t.device}; simple_pose_amd runs on the MI355X only (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()
