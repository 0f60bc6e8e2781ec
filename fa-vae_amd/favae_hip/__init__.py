"""ctypes binding of libfavae_hip.so (C ABI declared in include/favae_hip.h).

Plumbing only: PyTorch owns device memory and streams, every call below forwards raw device pointers, sizes and
the current HIP stream to the hand-written gfx950 kernels.  There is NO fallback: if the shared library is missing
or a call fails, a RuntimeError is raised (the product path must never silently run anything else).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, byref, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; streams that share a queue serialise.  This
# package runs the weight gradients on a second stream, and with an RCCL process group in the process (more streams) that stream ended up
# on the compute stream's queue: 8.6 % of the step at world 1 (ops._side_stream).  Ask for 8 queues while the runtime has not read the
# variable yet; ops._side_stream() additionally probes for a stream that really overlaps.
if "GPU_MAX_HW_QUEUES" not in os.environ and not torch.cuda.is_initialized():
    os.environ["GPU_MAX_HW_QUEUES"] = "8"


def _parse_queues(v):
    try:
        return int(v)
    except (TypeError, ValueError):
        return None


# What HIP saw: the value of the variable at import time, valid only if HIP was not yet initialised then (a later export changes the
# environment, not the runtime).  None = unknown (HIP was up before this import, or the value does not parse): TrainStep warns.
HW_QUEUES_AT_INIT = None if torch.cuda.is_initialized() else _parse_queues(os.environ.get("GPU_MAX_HW_QUEUES"))

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FAVAE_HIP_LIB") or os.path.join(_HERE, "libfavae_hip.so")     # FAVAE_HIP_LIB: same-box A/B of two builds
ABI_VERSION = 22

GATHER_PLAIN, GATHER_UPSAMPLE2, GATHER_DILATE2 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_LEAKY02, ACT_RELU = 0, 1, 2, 3

_ERR = {1: "bad argument", 2: "kernel launch failed", 3: "unsupported shape", 4: "workspace too small"}


class ConvDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("N", "Hin", "Win", "Cin", "Hout", "Wout", "Cout", "KH", "KW", "stride", "pad",
                                       "gather", "act", "affine_per_image", "lat_step", "lat_side", "lat_oh", "lat_ow",
                                       "pad_dw", "w_rec_offset")]


class WinoJob(Structure):          # include/favae_hip.h favae_wino_job
    _fields_ = [("w", c_void_p), ("out", c_void_p), ("amax", c_void_p), ("Cout", c_int32), ("Cin", c_int32), ("flip", c_int32),
                ("block0", c_int32)]


class ReduceJob(Structure):
    _fields_ = [("part", c_void_p), ("out", c_void_p), ("n", c_int64), ("slabs", c_int32), ("accumulate", c_int32)]


# name -> (restype, argtypes); mirrors include/favae_hip.h one to one (tests/test_abi.py checks the symbol list)
_P, _S = c_void_p, c_void_p
SIGNATURES = {
    "favae_abi_version": (c_int, []),
    "favae_conv_fwd": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _S]),
    "favae_conv_wants_split_weights": (c_int, [POINTER(ConvDesc), c_int]),
    "favae_set_conv_mode": (c_int, [c_int]),
    "favae_get_conv_mode": (c_int, []),
    "favae_split_weights_bytes": (c_size_t, [c_int64, c_int]),
    "favae_split_weights": (c_int, [_P, _P, c_int64, c_int, _S]),
    "favae_conv_fwd_split": (c_int, [POINTER(ConvDesc), _P, _P, c_int, _P, _P, _P, _P, _P, _P, _S]),
    "favae_absmax": (c_int, [_P, c_int64, _P, _S]),
    "favae_conv_wgrad_workspace": (c_size_t, [POINTER(ConvDesc)]),
    "favae_conv_wgrad": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, c_int, _P, c_size_t, _S]),
    "favae_weight_flip": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _S]),
    "favae_weight_flip_split": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _S]),
    "favae_downsample_dgrad_weights": (c_int, [_P, _P, c_int, c_int, c_int, _P, _S]),
    "favae_upsample_weights": (c_int, [_P, _P, c_int, c_int, _S]),
    "favae_upsample_wgrad_fold": (c_int, [_P, _P, c_int, c_int, c_int, _S]),
    "favae_conv_subpixel_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "favae_colsum_workspace": (c_size_t, [c_int64, c_int]),
    "favae_colsum": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, c_size_t, _S]),
    "favae_upsample2x_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _S]),
    "favae_gn_workspace": (c_size_t, [c_int, c_int64, c_int]),
    "favae_gn_stats": (c_int, [_P, _P, _P, c_int, c_int64, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P, c_size_t, _S]),
    "favae_gn_stats_bf16": (c_int, [_P, _P, _P, c_int, c_int64, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P, c_size_t, _S]),
    "favae_conv_bf16io_ok": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "favae_cast_bf16": (c_int, [_P, _P, c_int64, _S]),
    "favae_cast_f32": (c_int, [_P, _P, c_int64, _S]),
    "favae_gn_act_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, c_size_t, _S]),
    "favae_bn_update_running": (c_int, [_P, _P, c_int, c_int64, c_float, c_float, _P, _P, _S]),
    "favae_bgemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, _P, c_int64, c_int64, _P, c_int64, c_int64, _P,
                            c_int64, c_int64, c_int, c_int, _S]),
    "favae_softmax_rows": (c_int, [_P, _P, c_int64, c_int, _S]),
    "favae_softmax_rows_bwd": (c_int, [_P, _P, _P, c_int64, c_int, c_float, _S]),
    "favae_bgemm_sp": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, _P, c_int64, c_int64, _P, _P, c_int64, c_int64, _P, _P,
                               c_int64, c_int64, c_int, c_int, _S]),
    "favae_softmax_rows_lse": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, c_int, _S]),
    "favae_attn_bwd_point": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_float, _P, _S]),
    "favae_rowdot": (c_int, [_P, _P, _P, c_int64, c_int, _S]),
    "favae_blur_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _S]),
    "favae_blur_bwd_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "favae_blur_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _S]),
    "favae_blur_bwd_add": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t, _S]),
    "favae_ffl_spec_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "favae_ffl_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "favae_ffl_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, c_size_t, _S]),
    "favae_ffl_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _S]),
    "favae_vq_workspace": (c_size_t, [c_int, c_int, c_int]),
    "favae_vq_lookup": (c_int, [_P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, c_size_t, _S]),
    "favae_vq_segment_workspace": (c_size_t, [c_int, c_int]),
    "favae_vq_segment_sum": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, c_size_t, _S]),
    "favae_vq_ema_update": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, ctypes.c_double, _S]),
    "favae_vq_ste": (c_int, [_P, _P, _P, c_int64, _S]),
    "favae_hinge_mean": (c_int, [_P, c_int64, c_int, _P, _P, c_size_t, _S]),
    "favae_hinge_mean_bwd": (c_int, [_P, _P, c_int64, c_int, _P, _S]),
    "favae_act_bwd": (c_int, [_P, _P, c_int, c_int64, _P, _S]),
    "favae_lpips_level_workspace": (c_size_t, [c_int]),
    "favae_lpips_level": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, c_size_t, _S]),
    "favae_lpips_level_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _S]),
    "favae_maxpool2": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _S]),
    "favae_maxpool2_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _S]),
    "favae_channel_affine": (c_int, [_P, _P, _P, c_int64, c_int, _P, _S]),
    "favae_reduce_workspace": (c_size_t, [c_int64]),
    "favae_absdiff_sum": (c_int, [_P, _P, c_int64, c_float, _P, _P, c_size_t, _S]),
    "favae_sqdiff_sum": (c_int, [_P, _P, c_int64, c_float, _P, _P, c_size_t, _S]),
    "favae_absdiff_bwd": (c_int, [_P, _P, _P, c_float, c_int64, _P, _P, _S]),
    "favae_sqdiff_bwd": (c_int, [_P, _P, _P, c_float, c_int64, _P, _P, _S]),
    "favae_axpby": (c_int, [_P, c_float, _P, c_float, c_int64, _S]),
    "favae_u8_to_float_nhwc": (c_int, [_P, _P, c_int64, c_int, POINTER(ctypes.c_float), POINTER(ctypes.c_float), _S]),
    "favae_affine_rows": (c_int, [_P, _P, _P, _P, c_int, c_int64, c_int, c_int, _S]),
    "favae_layernorm_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_float, _S]),
    "favae_layernorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int, _S]),
    "favae_dropout": (c_int, [_P, _P, _P, c_int64, c_float, ctypes.c_uint32, _S]),
    "favae_nchw_to_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _S]),
    "favae_nhwc_to_nchw": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _S]),
    "favae_adam_step": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_int, c_float, _S]),
    "favae_conv_stats_tiles": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "favae_conv_fwd_split_stats": (c_int, [POINTER(ConvDesc), _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P, _S]),
    "favae_gn_stats_tiles": (c_int, [_P, c_int, _P, _P, c_int, c_int64, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P, c_size_t, _S]),
    "favae_conv_gnbwd_tiles": (c_int, [POINTER(ConvDesc), c_int]),
    "favae_gn_bwd_tiles_workspace": (c_size_t, [c_int, c_int, c_int]),
    "favae_conv_dgrad_gnbwd": (c_int, [POINTER(ConvDesc), _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, _S]),
    "favae_gn_act_bwd_tiles": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P,
                                       c_size_t, _S]),
    "favae_gn_bwd_colsum_blocks": (c_int, [c_int, c_int64, c_int]),
    "favae_gn_act_bwd_colsum": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P,
                                        c_size_t, _P, _P, _S]),
    "favae_colsum_finish": (c_int, [_P, c_int, c_int, _P, c_int, _S]),
    "favae_conv_wgrad_slabs": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, c_size_t, POINTER(c_int), _S]),
    "favae_reduce_slabs_grouped": (c_int, [POINTER(ReduceJob), c_int, _S]),
    "favae_split_weights_amax": (c_int, [_P, _P, c_int64, c_int, _P, _S]),
    "favae_conv_wino_ok": (c_int, [_P, c_int]),
    "favae_set_wino": (c_int, [c_int]),
    "favae_get_wino": (c_int, []),
    "favae_conv_wino4_ok": (c_int, [_P, c_int]),
    "favae_set_wino4": (c_int, [c_int]),
    "favae_set_wino_wide": (c_int, [c_int]),
    "favae_wino4_weights_bytes": (c_size_t, [c_int, c_int]),
    "favae_set_zero_arena": (c_int, [c_void_p, c_size_t]),
    "favae_wino_weights_grouped": (c_int, [_P, _P, c_int, _S]),
    "favae_wino_weights_bytes": (c_size_t, [c_int, c_int]),
    "favae_wino_weights": (c_int, [_P, _P, c_int, c_int, c_int, _P, _S]),
    "favae_segment_absmax": (c_int, [_P, _P, c_int, _P, _P, c_int, _P, _S]),
    "favae_prof_enable": (c_int, [c_int]),
    "favae_prof_reset": (c_int, []),
    "favae_prof_report": (c_int64, [ctypes.c_char_p, c_int64]),
}

_lib = None


def load():
    """dlopen the library (once).  Raises if it has not been built: there is no other code path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                           "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == ABI mismatch, let it propagate
        fn.restype = res
        fn.argtypes = args
    if lib.favae_abi_version() != ABI_VERSION:
        raise RuntimeError("libfavae_hip ABI version mismatch; rebuild")
    _lib = lib
    return lib


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


_hook = None


def set_call_hook(fn):
    """fn(name, args, launch) or None.  Used by bench.py to bracket selected launches with HIP events on the launch
    stream; `launch()` performs the real call.  Never used to reroute compute."""
    global _hook
    _hook = fn


def call(name, *args):
    """Invoke an int-returning entry point on the current stream (stream appended automatically)."""
    lib = load()
    if _hook is None:
        _chk(getattr(lib, name)(*args, stream()), name)
    else:
        _hook(name, args, lambda: _chk(getattr(lib, name)(*args, stream()), name))


def query(name, *args):
    return getattr(load(), name)(*args)


_ws_cache = {}


def workspace(nbytes: int, device, tag: str = "") -> torch.Tensor:
    """Stream-ordered scratch from PyTorch's caching allocator (a fresh tensor per call keeps stream semantics simple)."""
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def make_conv_desc(N, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, gather=GATHER_PLAIN, act=ACT_NONE,
                   affine_per_image=1, lattice=(0, 0, 0, 0), pad_dw=0, w_rec_offset=0):
    return ConvDesc(N, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, gather, act, affine_per_image, lattice[0], lattice[1],
                    lattice[2], lattice[3], pad_dw, w_rec_offset)
