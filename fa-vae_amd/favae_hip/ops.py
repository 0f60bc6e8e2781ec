"""torch.autograd.Function wrappers around libfavae_hip (the only compute path; no fallbacks).

Tensors are ordinary torch CUDA tensors shaped (N, C, H, W) whose *memory* is channels-last (NHWC), so the module
code reads like the reference's NCHW code while every kernel sees coalesced channel-fastest data.
"""
from __future__ import annotations

import math
import ctypes
import os
from ctypes import byref

import torch

from . import (ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_SILU, GATHER_DILATE2, GATHER_PLAIN, GATHER_UPSAMPLE2, ReduceJob, call,
               make_conv_desc, ptr, query, workspace)

CL = torch.channels_last


# ---------------------------------------------------------------------------------------------------------------
# layout helpers
# ---------------------------------------------------------------------------------------------------------------
def _require_gpu(t, bf16_ok=False):
    if not t.is_cuda:
        raise RuntimeError("favae_hip ops run on MI355X only (tensor is on %s); there is no CPU fallback" % t.device)
    if t.dtype != torch.float32 and not (bf16_ok and t.dtype == torch.bfloat16):
        raise RuntimeError("favae_hip ops compute in fp32 (got %s)" % t.dtype)


def new_cl(N, C, H, W, device, dtype=torch.float32):
    return torch.empty((N, C, H, W), dtype=dtype, device=device, memory_format=CL)


# ---- bf16 activation STORAGE (round 6; BASELINE configs[4] "bf16", favae_scripts/train_favae.py:239-240) ---------------------------
# In the conv mode b1 (one bf16 plane per operand) the dense 3x3 convs of the ResnetBlock chain -- forward, data gradient with the
# GroupNorm-backward epilogue, weight gradient -- and the GroupNorm-backward apply pass between them read and write their activation
# tensors as torch.bfloat16 (include/favae_hip.h FAVAE_PLANES_BF16IO / FAVAE_ACT_BF16IO): the big layers of that mode sit at the HBM
# roofline in fp32 I/O.  Every other op keeps fp32 I/O: to_cl() widens a bf16 input with one conversion pass, and the autograd engine
# converts the fp32 gradient such an op returns back to the dtype of the tensor it belongs to.  Statistics, losses, the codebook lookup,
# FFL and Adam stay fp32 (reference: models/l2_quantize.py:391,395 keep the quantizer out of autocast).
# FAVAE_BF16_STORAGE=0 / set_bf16_storage(False): activations stay fp32 in b1 as in rounds 1-5 (A/B arm).
_BF16_STORAGE = os.environ.get("FAVAE_BF16_STORAGE", "1") != "0"
BF16IO_PLANES = 0x400                  # include/favae_hip.h FAVAE_PLANES_BF16IO
BF16IO_ACT = 0x200                     # include/favae_hip.h FAVAE_ACT_BF16IO


def set_bf16_storage(on):
    global _BF16_STORAGE
    prev, _BF16_STORAGE = _BF16_STORAGE, bool(on)
    return prev


def bf16_storage():
    return _BF16_STORAGE and query("favae_get_conv_mode") == 4


def cast_bf16(t):
    """fp32 -> bf16 copy of a dense tensor (same shape and strides), round to nearest even; by-products riding on t are carried over"""
    if t.dtype == torch.bfloat16:
        return t
    out = torch.empty_like(t, dtype=torch.bfloat16)
    call("favae_cast_bf16", ptr(t), ptr(out), t.numel())
    for attr in ("_favae_gnstats", "_favae_amax", "_favae_dycs"):
        st = getattr(t, attr, None)
        if st is not None:
            if attr == "_favae_gnstats":
                st = (st[0], st[1], out._version)
            elif attr == "_favae_amax":
                st = (st[0], out._version, st[2])
            else:
                st = (st[0], st[1], st[2], out._version, st[4])
            setattr(out, attr, st)
    return out


def cast_f32(t):
    if t.dtype == torch.float32:
        return t
    out = torch.empty_like(t, dtype=torch.float32)
    call("favae_cast_f32", ptr(t), ptr(out), t.numel())
    return out


def _is_cl(t):
    """True when the memory of (N,C,H,W) tensor `t` is NHWC-dense (size-1 dims have no say)."""
    N, C, H, W = t.shape
    want = (H * W * C, 1, W * C, C)
    return all(sz == 1 or st == w for sz, st, w in zip(t.shape, t.stride(), want))


def to_cl(t, keep_bf16=False):
    """Return `t` (N,C,H,W) with NHWC memory; NCHW-contiguous inputs go through the HIP transpose kernel.  A bf16 tensor (bf16
    activation storage) is widened to fp32 unless the caller has a bf16 kernel for it (keep_bf16)."""
    if t.dtype == torch.bfloat16 and t.is_cuda:
        if _is_cl(t):
            return t if keep_bf16 else cast_f32(t)
        # a bf16 tensor in another layout: only a gradient the autograd engine converted from a caller's NCHW fp32 tensor (tests, module
        # boundaries) -- widen it with torch, lay it out below, narrow it again if the caller keeps bf16
        t32 = to_cl(t.float())
        return cast_bf16(t32) if keep_bf16 else t32
    _require_gpu(t)
    if _is_cl(t):
        return t
    N, C, H, W = t.shape
    if not t.is_contiguous():
        t = t.contiguous()
    y = new_cl(N, C, H, W, t.device)
    call("favae_nchw_to_nhwc", ptr(t), ptr(y), N, C, H, W)
    return y


def to_nchw(t):
    """NCHW-contiguous copy of a channels-last tensor (module boundary only)."""
    _require_gpu(t)
    if t.is_contiguous():
        return t
    t = to_cl(t)
    N, C, H, W = t.shape
    y = torch.empty((N, C, H, W), dtype=torch.float32, device=t.device)
    call("favae_nhwc_to_nchw", ptr(t), ptr(y), N, C, H, W)
    return y


class _ToCLFn(torch.autograd.Function):
    """Autograd-transparent layout change (values unchanged, so the gradient passes straight through)."""

    @staticmethod
    def forward(ctx, t):
        return to_cl(t)

    @staticmethod
    def backward(ctx, g):
        return g


def u8_to_float(x_u8, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """(N,H,W,C) uint8 device tensor -> (N,C,H,W) fp32 channels_last tensor, (u/255 - mean)/std: the ToTensor + Normalize tail of
    datasets/general_dataloader.py:33-38 on the device (bit-identical to torchvision's fp32 arithmetic), written straight in the
    layout the convs read.  The batch crosses PCIe as bytes."""
    if not (x_u8.is_cuda and x_u8.dtype == torch.uint8 and x_u8.dim() == 4 and x_u8.is_contiguous()):
        raise RuntimeError("u8_to_float needs a contiguous (N,H,W,C) uint8 tensor on the GPU (no CPU path)")
    N, H, W, C = x_u8.shape
    if not (1 <= C <= 4 and len(mean) == C and len(std) == C):
        raise RuntimeError("u8_to_float: 1..4 channels with one mean/std per channel")
    out = torch.empty((N, H, W, C), dtype=torch.float32, device=x_u8.device)
    m = (ctypes.c_float * C)(*[float(v) for v in mean])
    sd = (ctypes.c_float * C)(*[float(v) for v in std])
    call("favae_u8_to_float_nhwc", ptr(x_u8), ptr(out), N * H * W, C, m, sd)
    return out.permute(0, 3, 1, 2)


def as_cl(t):
    """Module-boundary version of to_cl(): safe on tensors that require grad."""
    if _is_cl(t) and t.is_cuda:
        return t
    return _ToCLFn.apply(t) if t.requires_grad else to_cl(t)


class _AddFn(torch.autograd.Function):
    """c = a + b on the HIP axpby kernel (trunk + FCM adds of the convolutional FCM decoders, codec.py:533-548)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = to_cl(a), to_cl(b)
        out = torch.empty_like(a)
        call("favae_axpby", ptr(a), 1.0, ptr(out), 0.0, a.numel())
        call("favae_axpby", ptr(b), 1.0, ptr(out), 1.0, a.numel())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _AddFn.apply(a, b)


def weight_ohwi(w):
    """(Cout,Cin,KH,KW) parameter -> pointer-compatible OHWI tensor (zero copy when the parameter is channels-last)."""
    _require_gpu(w)
    if w.dim() == 2:
        return w if w.is_contiguous() else w.contiguous()
    if _is_cl(w):
        return w
    return to_cl(w)


# ---------------------------------------------------------------------------------------------------------------
# GroupNorm statistics (no autograd by itself; used inside FusedConv)
# ---------------------------------------------------------------------------------------------------------------
# per-tile (sum y, sum y^2) emitted by the conv that produced a tensor ride on the tensor object as `_favae_gnstats`
_GNSTATS_FUSE = os.environ.get("FAVAE_GNSTATS_FUSE", "1") != "0"


def gn_stats(x, gamma, beta, groups, eps=1e-5, with_bound=False):
    """GroupNorm statistics + the per-(image, channel) affine the conv kernels apply on load.  with_bound: also return the
    device scalar bounding |act(GN(x))| that the fp16 split-precision conv kernels scale their operand with.
    When the conv that produced x left its per-tile partial sums on the tensor (FusedConvFn.forward), the streaming pass over x is
    skipped and only the finalize kernel runs."""
    N, C, H, W = x.shape
    dev = x.device
    mean = torch.empty((N, groups), dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    scale = torch.empty((N, C), dtype=torch.float32, device=dev)
    shift = torch.empty_like(scale)
    bound = torch.empty((1,), dtype=torch.float32, device=dev) if with_bound else None
    nb = query("favae_gn_workspace", N, H * W, C)
    ws = workspace(nb, dev)
    pre = getattr(x, "_favae_gnstats", None) if _GNSTATS_FUSE else None
    if pre is not None and pre[0].numel() == N * pre[1] * C * 2 and pre[2] == x._version:
        call("favae_gn_stats_tiles", ptr(pre[0]), pre[1], ptr(gamma), ptr(beta), N, H * W, C, groups, eps, ptr(mean), ptr(rstd),
             ptr(scale), ptr(shift), ptr(bound), ptr(ws), ws.numel())
    else:
        call("favae_gn_stats_bf16" if x.dtype == torch.bfloat16 else "favae_gn_stats", ptr(x), ptr(gamma), ptr(beta), N, H * W, C, groups,
             eps, ptr(mean), ptr(rstd), ptr(scale), ptr(shift), ptr(bound), ptr(ws), ws.numel())
    if with_bound:
        return mean, rstd, scale, shift, bound
    return mean, rstd, scale, shift


_FP16_PLANES = None


class WeightMaxima:
    """max |w| of every parameter that is a view of one flat buffer, refreshed by ONE launch (favae_segment_absmax) -- the fp16
    split-precision convs need the maximum of their weight tensor on every call; without this each conv call runs its own reduction
    (98 launches + 98 memsets per training step).  refresh() is called by the owner of the flat buffer after every update of it; a
    parameter's entry is used only while the parameter's version counter, the flat buffer's version counter (a write THROUGH the flat
    buffer -- pflat.copy_(ckpt), dist.broadcast(pflat) -- moves that one only: `p.data = view` gives every parameter a counter of its
    own) and the storage's update count (raw-pointer writes: the optimizer kernel, invalidate_weight_caches) are the ones seen at refresh
    time."""

    def __init__(self, flat, params):
        self.flat, self.params = flat, [p for p in params if p.dim() >= 2]
        dev = flat.device
        base = flat.data_ptr()
        offs, cseg, cfirst = [], [], []
        self.params.sort(key=lambda p: p.data_ptr())
        for i, p in enumerate(self.params):
            o = (p.data_ptr() - base) // 4
            offs.append(o)
            for f in range(o, o + p.numel(), 4096):
                cseg.append(i)
                cfirst.append(f)
        ends = [o + p.numel() for o, p in zip(offs, self.params)]
        self.nseg = len(self.params)
        # the kernel reads seg_off[s + 1] as the END of segment s: store ends shifted by one (seg_off[0] unused)
        self.seg_off = torch.tensor([0] + ends, dtype=torch.int64, device=dev)
        self.chunk_seg = torch.tensor(cseg, dtype=torch.int32, device=dev)
        self.chunk_first = torch.tensor(cfirst, dtype=torch.int64, device=dev)
        self.out = torch.zeros((self.nseg,), dtype=torch.float32, device=dev)
        self.versions = [None] * self.nseg
        for i, p in enumerate(self.params):
            p._favae_wmax = (self, i)

    def refresh(self):
        call("favae_segment_absmax", ptr(self.flat), ptr(self.seg_off), self.nseg, ptr(self.chunk_seg), ptr(self.chunk_first),
             self.chunk_seg.numel(), ptr(self.out))
        self.versions = [_weights_key(p) for p in self.params]


def _weights_key(p):
    """what a cache entry derived from parameter p (its max|w|, its Winograd records) is valid for: p's version counter, the version
    counter of the flat buffer p is a view of (if any) and the update count of the storage"""
    ent = getattr(p, "_favae_wmax", None)
    return (p._version, ent[0].flat._version if ent is not None else -1, _weights_epoch(p))


def _weight_amax(w):
    ent = getattr(w, "_favae_wmax", None)
    if ent is None:
        return None
    wm, i = ent
    if wm.versions[i] != _weights_key(w):
        return None
    return wm.out[i:i + 1]


# The optimizer kernel writes parameters through a raw pointer: no tensor version counter moves.  Caches derived from the weights (their
# maxima, their Winograd records) therefore also carry the update count of the storage the weights live in: adam_step() bumps it, the
# cache's refresh() records it, and an entry is used only while both the parameter's version and that count are the ones seen at refresh.
_WEIGHT_EPOCH = {}


def _storage_key(t):
    return t.untyped_storage().data_ptr()


def _touch_weights(t):
    k = _storage_key(t)
    _WEIGHT_EPOCH[k] = _WEIGHT_EPOCH.get(k, 0) + 1


def _weights_epoch(t):
    return _WEIGHT_EPOCH.get(_storage_key(t), 0)


def invalidate_weight_caches(t):
    """Call after writing parameters that live in the storage of tensor `t` in a way no version counter sees (a raw-pointer kernel, a
    foreign library): every cached max|w| and Winograd record of parameters in that storage is dropped until the owner's next refresh()."""
    _touch_weights(t)


class WinoRecords:
    """Winograd weight records (csrc/conv_wino.h) of every dense 3x3 conv weight in `params`, both directions, made by ONE launch
    (favae_wino_weights_grouped) -- 136 small launches per training step otherwise.  Needs the maxima of a WeightMaxima over the same
    parameters (the records' scale); refresh() is called by the owner of the parameters after every update, behind the maxima's refresh.
    A parameter's records are used only while its version counter is the one seen at refresh time."""

    def __init__(self, params):
        self.params, self._jobs, self.bufs, self.versions = [], [], {}, {}
        self.dev, self.store, self._dirty, self._have = None, None, False, set()
        for p in params:
            ent = getattr(p, "_favae_wmax", None)
            if ent is None or p.dim() != 4 or tuple(p.shape[2:]) != (3, 3) or not _is_cl(p):
                continue
            co, ci = int(p.shape[0]), int(p.shape[1])
            for flip in (0, 1):
                o, i = (ci, co) if flip else (co, ci)
                if o % 64 or i % 16:
                    continue
                self._add(p, flip, co, ci)
            self.dev = p.device
            self.params.append(p)
        for p in self.params:
            p._favae_wino = self

    def _add(self, p, key, co, ci):
        self._jobs.append((p, key, co, ci))
        self._have.add((id(p), key))
        self._dirty = True

    @property
    def n(self):
        return len(self._jobs)

    def want(self, p, key):
        """F(4x4, 3x3) records (key = 2 forward, 3 data gradient; csrc/conv_wino4.h) are made for the (weight, direction) pairs a conv launch
        actually asked for -- which layers tile into that kernel depends on their spatial size, which this object does not know.  The first
        request is served by a one-off favae_wino_weights launch of the caller; from the next refresh() on the grouped launch makes them."""
        if (id(p), key) in self._have or not any(q is p for q in self.params):
            return
        self._add(p, key, int(p.shape[0]), int(p.shape[1]))

    def _materialize(self):
        """record store + device job table, on the first refresh that needs them (64 bytes per (Cout, Cin) pair and direction, 144 for the
        F(4x4) records: ~1 GB for the f=16 model -- not spent when the conv mode is not h3 or the Winograd path is off)"""
        from . import WinoJob
        # the records a mode reads: h3 / h1 the fp16 ones (h1: their head plane), b1 those with a bf16 head plane (key | 4, asked for by the
        # conv launches through want())
        bf = 4 if get_conv_mode() == "b1" else 0
        self._mode_bf = bf
        jobs = [j for j in self._jobs if (j[1] & 4) == bf]
        self._njobs = len(jobs)
        if not jobs:
            self.store, self.bufs, self._dirty = None, {}, False
            return
        sizes = [int(query("favae_wino4_weights_bytes" if key & 2 else "favae_wino_weights_bytes", co, ci)) for (p, key, co, ci) in jobs]
        offs, nbytes = [], 0
        for sz in sizes:
            offs.append(nbytes)
            nbytes += (sz + 255) // 256 * 256
        self.nbytes = nbytes
        self.store = torch.empty((nbytes,), dtype=torch.uint8, device=self.dev)
        self.bufs = {}
        arr = (WinoJob * len(jobs))()
        block_job, b0 = [], 0
        for k, (p, key, co, ci) in enumerate(jobs):
            wm, i = p._favae_wmax
            nb = (co * ci // 8 + 255) // 256
            arr[k].w, arr[k].out, arr[k].amax = p.data_ptr(), self.store.data_ptr() + offs[k], wm.out[i:i + 1].data_ptr()
            arr[k].Cout, arr[k].Cin, arr[k].flip, arr[k].block0 = co, ci, key, b0
            block_job += [k] * nb
            b0 += nb
            self.bufs.setdefault(id(p), {})[key] = self.store[offs[k]:offs[k] + sizes[k]]
        self.jobs = torch.frombuffer(bytearray(bytes(memoryview(arr))), dtype=torch.uint8).to(self.dev)
        self.block_job = torch.tensor(block_job, dtype=torch.int32, device=self.dev)
        self.nblocks = b0
        self._dirty = False

    def refresh(self):
        on = bool(query("favae_get_wino"))
        mode = get_conv_mode()
        # the one-plane modes take the Winograd kernel only with FAVAE_WINO1 on (b1: and only in the wide tiling): without them the
        # library never reads a record and the store (~1 GB) and its grouped launch would be spent for nothing
        if mode in ("h1", "b1") and (os.environ.get("FAVAE_WINO1", "1") == "0" or
                                     (mode == "b1" and os.environ.get("FAVAE_WINO_WIDE", "1") == "0")):
            on = False
        if not self.n or not on or mode not in ("h3", "h1", "b1"):
            self.versions = {}
            self.store, self.bufs = None, {}
            return
        if self.store is None or self._dirty or getattr(self, "_mode_bf", None) != (4 if mode == "b1" else 0):
            self._materialize()
        if not getattr(self, "_njobs", 0):
            self.versions = {}
            return
        call("favae_wino_weights_grouped", ptr(self.jobs), ptr(self.block_job), self.nblocks)
        self.versions = {id(p): _weights_key(p) for p in self.params}

    def get(self, p, flip):
        if self.versions.get(id(p)) != _weights_key(p):
            return None
        return self.bufs.get(id(p), {}).get(flip)


def _wino_cached(w, flip):
    wr = getattr(w, "_favae_wino", None)
    if wr is None:
        return None
    if flip & 6:
        wr.want(w, flip)
    return wr.get(w, flip)


# Winograd F(4x4, 3x3) (csrc/conv_wino4.h): 0.56 x the matrix work of the F(2x2) kernel at ~6 x its rounding error (2.3e-6 rms of the output
# range per conv).  FAVAE_WINO4: "0" (default) never; "1" data gradients only -- no codebook index depends on them and their parity bar is
# 5e-3; "2" also the forward convs of modules that opted in (`wino4_forward(True)` around the call: the decoder, behind the quantizer).
# "1" was the default until the F(2x2) kernel got its 16 x 8 x 128 tiling (conv_wino.h WIDE): that one is as fast at 256^2 and faster
# below with F(2x2)'s rounding, and the step is 0.4 ms shorter without F(4x4) (profiles/r05_wide_step_ab.txt, r05_wide_bench.txt).
_WINO4 = os.environ.get("FAVAE_WINO4", "0")
_WINO4_FWD = [False]


class wino4_forward:
    """context: forward convs launched inside may take the F(4x4, 3x3) kernel when FAVAE_WINO4=2 (never the encoder: indices)"""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = _WINO4_FWD[0]
        _WINO4_FWD[0] = self.on
        return self

    def __exit__(self, *exc):
        _WINO4_FWD[0] = self.prev
        return False


def set_wino4(mode):
    """"0" / "1" / "2" (see FAVAE_WINO4); returns the previous mode"""
    global _WINO4
    prev, _WINO4 = _WINO4, str(mode)
    return prev


def set_wino_wide(on):
    """16 x 8-pixel x 128-channel tiling of the F(2x2) Winograd kernel on / off (FAVAE_WINO_WIDE; bit-identical results); returns the
    previous setting"""
    return query("favae_set_wino_wide", 1 if on else 0)


_WINO4_FWD_MINPIX = int(os.environ.get("FAVAE_WINO4_FWD_MINPIX", "65536"))   # forward: only where it measured faster (128 -> 128 @256^2: 1.12 x)


# One-plane Winograd (csrc/conv_wino.h PLN = 1 / 4) in the 16-bit mixed-precision modes: h1 (one fp16 plane) everywhere the kernel applies;
# b1 (one bf16 plane) on the DATA GRADIENTS only -- B^T d B in bf16 costs 1.65 x the direct bf16 conv's error, and with the forward convs on
# it the step leaves what the REFERENCE does under bf16 autocast (cfg5_256: 14 index flips / loss_l1 7.0e-3 against 12 / 2.4e-3; the direct
# kernels: 6 / 7.8e-4).  The forward convs of the LPIPS feature extractor are the exception (see _wino1_wanted).  FAVAE_WINO1_FWD=1 forces every
# forward conv too (experiments).
_WINO1_FWD = os.environ.get("FAVAE_WINO1_FWD", "0") == "1"


def _wino1_wanted(planes, dgrad, act=ACT_NONE):
    # ReLU on load = the frozen VGG16 of LPIPS (losses/lpips.py: the only ReLU convs): behind the reconstruction, no index, x_recon or
    # reconstruction loss depends on it -- its forward convs take the Winograd kernel in b1 as well
    return planes != 4 or dgrad or _WINO1_FWD or act == ACT_RELU


def _wino4_wanted(d, has_affine, dgrad):
    if _WINO4 == "0" or not (dgrad or (_WINO4 == "2" and _WINO4_FWD[0] and d.Hin * d.Win >= _WINO4_FWD_MINPIX)):
        return False
    return bool(query("favae_conv_wino4_ok", byref(d), 1 if has_affine else 0))


def _fp16_planes():
    """True when the library runs its split-precision convs with scaled fp16 planes, which need operand ranges
    (FAVAE_CONV_MODE=h3, the default: two planes; h1: one plane, the 16-bit mixed-precision mode)."""
    global _FP16_PLANES
    if _FP16_PLANES is None:
        d = make_conv_desc(1, 16, 16, 128, 16, 16, 128, 3, 3, 1, 1, GATHER_PLAIN, ACT_NONE, 1)
        _FP16_PLANES = query("favae_conv_wants_split_weights", byref(d), 0) in (1, 2)
    return _FP16_PLANES


_WREC = {1: 8, 2: 16, 3: 24, 4: 8}    # bytes of one pre-split record (4 weights) per scheme id
_WINO_FLIP_FWD = os.environ.get("FAVAE_WINO_FLIP_FWD", "1") != "0"
PLANES_WINO = 0x100                   # include/favae_hip.h FAVAE_PLANES_WINO: the records are Winograd records (favae_wino_weights)
PLANES_WINO4 = 0x200                  # ... F(4x4, 3x3) records (flip | 2): conv3x3_wino4_sp_kernel
CONV_MODES = {"fp32": 0, "h1": 1, "h3": 2, "b6": 3, "b1": 4}


def set_conv_mode(mode):
    """Select the conv arithmetic at run time (same meaning as FAVAE_CONV_MODE): "h3" (default, fp32-grade, two scaled fp16
    planes), "b6" (fp32-grade, three bf16 planes), "fp32" (fp32-MFMA kernels), "h1" -- ONE scaled fp16 plane with fp32
    accumulation -- or "b1" -- ONE bf16 plane (round to nearest even) with fp32 accumulation: the 16-bit mixed-precision modes
    (b1 is what `accelerate --mixed_precision bf16` gives the reference's convs, BASELINE configs[4]; h1 keeps three more
    significand bits), not fp32-grade.  Returns the previous mode name."""
    global _FP16_PLANES
    prev = query("favae_get_conv_mode")
    _chk_mode = CONV_MODES[mode]
    if query("favae_set_conv_mode", _chk_mode) != 0:
        raise RuntimeError("favae_set_conv_mode failed")
    _FP16_PLANES = None
    return {v: k for k, v in CONV_MODES.items()}[prev]


def get_conv_mode():
    return {v: k for k, v in CONV_MODES.items()}[query("favae_get_conv_mode")]


# ---- zero arena: pre-zeroed atomicMax targets ------------------------------------------------------------------------------
# The max|x| scalars of the fp16 split scheme are atomicMax targets that must start at zero; each library call zeroes its own with a
# 4-byte memset launch (117-156 per training step, 5 us + a queue gap each) unless the pointer lies in a range the caller keeps zero
# (include/favae_hip.h, favae_set_zero_arena).  TrainStep.step() zeroes ONE buffer per step (zero_arena_reset) and the allocation sites
# below hand out its floats one after the other; outside a TrainStep (inference, tests) or when the buffer is used up they fall back to
# torch.empty + the library's own memset.  A by-product scalar that rides on a tensor (`_favae_amax`, `_favae_dycs`) carries the arena
# epoch it was taken in: after the next reset the slot is zero again and must not be trusted.
_ARENA = {"buf": None, "pos": 0, "epoch": 0, "n": 4096, "stream": None}


_ZERO_ARENA = os.environ.get("FAVAE_ZERO_ARENA", "1") != "0"      # A/B switch (0: every library call zeroes its own target)


def zero_arena_reset(dev):
    """start of a training step: every slot handed out before is invalid from here on; one memset for all of this step's targets"""
    a = _ARENA
    if not _ZERO_ARENA:
        return
    if a["buf"] is None or a["buf"].device != torch.device(dev):
        a["buf"] = torch.zeros((a["n"],), dtype=torch.float32, device=dev)
        query("favae_set_zero_arena", a["buf"].data_ptr(), a["n"] * 4)
    else:
        a["buf"].zero_()
    a["pos"] = 0
    a["epoch"] += 1
    a["stream"] = torch.cuda.current_stream()       # slots are zero in THIS stream's order only


def zero_arena_off():
    """drop the arena (the library zeroes every target itself again)"""
    _ARENA["buf"], _ARENA["pos"] = None, 0
    _ARENA["epoch"] += 1
    query("favae_set_zero_arena", None, 0)


def _in_arena(t):
    """does tensor t (a float32[1] operand range) live in the zero arena?"""
    b = _ARENA["buf"]
    return t is not None and b is not None and b.data_ptr() <= t.data_ptr() < b.data_ptr() + b.numel() * 4


def _check_arena_epoch(ctx, *ranges):
    """an operand range saved for backward may be an arena slot, which TrainStep.step() / zero_arena_reset() zeroes and hands
    out again.  A graph built before such a reset and back-propagated after it (retained graph, a forward outside step(), two TrainSteps
    interleaved) would scale its fp16 planes with max|x| = 0 or a stranger's value, silently.  Fail loudly instead.  (Slots that went
    through ctx.save_for_backward are also caught by autograd itself: they are views of the arena buffer, whose in-place zero_() bumps
    the version counter -- "modified by an inplace operation"; tests/test_gpu_model.py::test_backward_across_an_arena_reset_fails_loudly.)"""
    if getattr(ctx, "arena_epoch", None) != _ARENA["epoch"] and any(_in_arena(r) for r in ranges):
        raise RuntimeError("favae_hip: this graph was built before the last zero_arena_reset() (TrainStep.step() starts with one); "
                           "operand ranges it saved in the zero arena are gone -- run forward and backward inside the same step")


def _max_target(dev):
    """a float32[1] that is zero in stream order when the kernel that atomicMax'es into it runs: the next arena slot, else a fresh tensor
    (which the library memsets)"""
    a = _ARENA
    if a["buf"] is not None and a["pos"] < a["n"] and a["buf"].device == torch.device(dev) and torch.cuda.current_stream() == a["stream"]:
        a["pos"] += 1
        return a["buf"][a["pos"] - 1:a["pos"]]
    return torch.empty((1,), dtype=torch.float32, device=dev)


def absmax(t):
    """device scalar max|t| (operand range of the fp16 split-precision conv kernels); taken from the producing conv's epilogue when
    it left one on the tensor (`_favae_amax`, valid while the version counter is unchanged and the arena slot has not been recycled)"""
    pre = getattr(t, "_favae_amax", None)
    if pre is not None and pre[1] == t._version and pre[2] == _ARENA["epoch"]:
        return pre[0]
    out = _max_target(t.device)
    call("favae_absmax", ptr(t), t.numel(), ptr(out))
    return out


_DIRECT_GRAD = True
_SUBPIXEL_DGRAD = os.environ.get("FAVAE_SUBPIXEL_DGRAD", "1") != "0"     # Downsample data gradient by output parity (A/B switch)

# ---- second HIP stream for the weight gradients ----------------------------------------------------------------------------
# In a conv's backward the weight gradient (matrix-pipe bound, accumulated straight into the flat gradient buffer) is
# independent of the chain  dgrad -> GroupNorm backward -> next layer: it is launched on a side stream, right after the data
# gradient, and overlaps with the HBM-bound kernels of that chain (GroupNorm passes, bias gradients, blur / FFT backward).
# FAVAE_WGRAD_STREAM=0 disables it (A/B switch).
# Operand lifetime: the autograd engine accumulates IN PLACE into a gradient tensor it holds the last reference to (e.g. the
# second consumer of the `g` that _AddFn.backward hands to both inputs, or a `dres = dy` alias), and the caching allocator hands a
# freed block out again in main-stream order -- neither is ordered against the side stream.  So every tensor a side-stream launch
# reads stays referenced in _SIDE["pending"] until an event recorded behind that launch has completed (or the main stream has
# waited for the side stream): a live reference forces the engine's out-of-place add and keeps the block out of the allocator.
# _SIDE["delay"] > 0 (tests only) puts a busy-wait of that many cycles in front of every side-stream launch.
_SIDE = {"stream": None, "used": False, "on": os.environ.get("FAVAE_WGRAD_STREAM", "1") != "0", "pending": [], "delay": 0,
         "jobs": [], "targets": set()}
# Split-K slab reductions of the side-stream weight gradients are collected and run as ONE grouped launch per flush (end of backward,
# or right before a gradient segment is handed to the all-reduce) instead of one latency-bound launch per layer.  FAVAE_DEFER_REDUCE=0:
# every weight gradient reduces its own slabs (A/B switch; the sums are taken in the same order: bit-identical).
_DEFER_REDUCE = os.environ.get("FAVAE_DEFER_REDUCE", "1") != "0"
_FLUSH_EVERY = int(os.environ.get("FAVAE_FLUSH_EVERY", "12"))
_SERIALIZE_MFMA = os.environ.get("FAVAE_SERIALIZE_MFMA", "0") == "1"


def streams_overlap(a, b, cycles=4_000_000):
    """diagnostic: do two HIP streams run a spin kernel each concurrently (torch.cuda._sleep, timed against one alone)?  NOT a test for
    the queue oversubscription described at _side_stream(): there every pair of streams still passes this probe."""
    ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        ea0.record(a)
        torch.cuda._sleep(cycles)
        ea1.record(a)
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        eb0.record(a)
        torch.cuda._sleep(cycles)
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles)
    a.wait_stream(b)
    with torch.cuda.stream(a):
        eb1.record(a)
    torch.cuda.synchronize()
    return eb0.elapsed_time(eb1) < 1.5 * ea0.elapsed_time(ea1)


def _side_stream():
    if _SIDE["stream"] is None:
        # FAVAE_SIDE_PRIORITY (A/B switch): HIP priority of the weight-gradient stream (torch: lower number = higher priority, clamped
        # to the device's range); default = the normal priority the main stream has.
        # HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default).  With an RCCL process group in the process
        # (its internal streams + the communication stream) the step's two compute streams lost ALL of their overlap -- the in-step
        # kernel times equal the single-stream ones, 134.2 -> 145.8 ms per step at world 1, i.e. at every N >= 2 -- although any two
        # streams still run a pair of spin kernels concurrently (streams_overlap above: the cross-stream event waits of every
        # weight-gradient launch are what serialises on an oversubscribed queue).  8 queues remove it (134.7 ms with the full
        # distributed path): favae_hip/__init__.py and bench.py ask for them before HIP is initialised; a process that initialised HIP
        # earlier with the default gets a warning from TrainStep (profiles/r05_dist_overhead.txt).
        _SIDE["stream"] = torch.cuda.Stream(priority=int(os.environ.get("FAVAE_SIDE_PRIORITY", "0")))
    return _SIDE["stream"]


def _defer_reduction(part, out, n, slabs, accumulate):
    """queue out[:n] (+)= sum of the `slabs` slabs in `part` for the next flush_reductions() (side stream)"""
    key = out.data_ptr()
    if key in _SIDE["targets"]:            # a second gradient for the same parameter in this pass (a module applied twice): the
        flush_reductions()                 # grouped kernel must not accumulate twice into one range concurrently
    _SIDE["targets"].add(key)
    _SIDE["jobs"].append((part, out, int(n), int(slabs), int(accumulate)))
    if len(_SIDE["jobs"]) >= _FLUSH_EVERY:      # keep the flushes inside backward: only the last, short one is exposed at its end
        flush_reductions()


def flush_reductions():
    """one grouped launch (side stream, behind the weight-gradient kernels that wrote the slabs) for every queued slab reduction"""
    jobs = _SIDE["jobs"]
    if not jobs:
        return
    arr = (ReduceJob * len(jobs))()
    for a, (part, out, n, slabs, acc) in zip(arr, jobs):
        a.part, a.out, a.n, a.slabs, a.accumulate = part.data_ptr(), out.data_ptr(), n, slabs, acc
    side = _side_stream()
    with torch.cuda.stream(side):
        call("favae_reduce_slabs_grouped", arr, len(jobs))
        ev = torch.cuda.Event()
        ev.record(side)
    _SIDE["pending"].append((ev, [t for j in jobs for t in j[:2]]))
    _SIDE["jobs"] = []
    _SIDE["targets"] = set()


def reset_side_state():
    """Start of a training step: drop slab reductions a previous, ABORTED backward pass left queued (an exception in a later node, an
    OOM retry) -- flushed into the freshly zeroed gradient buffer they would corrupt this step's gradients without any error.  The main
    stream is first ordered behind whatever the side stream still runs, so the dropped workspaces can be reused safely."""
    if _SIDE["jobs"] or _SIDE["targets"] or _SIDE["used"] or _SIDE["pending"]:
        if _SIDE["stream"] is not None:
            torch.cuda.current_stream().wait_stream(_SIDE["stream"])
        _SIDE["jobs"], _SIDE["targets"], _SIDE["used"] = [], set(), False
        _SIDE["pending"].clear()


def side_stream_flushed():
    """the weight-gradient stream with every queued slab reduction launched on it (what a gradient exchange has to wait for)"""
    flush_reductions()
    return _SIDE["stream"]


def sync_side_stream():
    """make the current stream wait for everything queued on the side stream.  Queued automatically as an end-of-backward
    callback of the autograd engine by the first conv that uses the side stream, so gradients are complete (in stream order)
    whenever .backward() / autograd.grad() returns."""
    flush_reductions()
    if _SIDE["used"]:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
        _SIDE["used"] = False
    _SIDE["pending"].clear()               # later main-stream work is ordered behind the side stream now


def _side_launch(fn, operands):
    """Run fn() (kernel launches) on the side stream after everything queued on the current stream, keep `operands` alive until the
    side stream is past these launches, and make the end of the running backward pass wait for them."""
    side = _side_stream()
    pend = _SIDE["pending"]
    while pend and pend[0][0].query():                     # retire launches the side stream has finished
        pend.pop(0)
    side.wait_stream(torch.cuda.current_stream())          # operands (dy, its range from the bias-gradient pass) are ready
    with torch.cuda.stream(side):
        if _SIDE["delay"]:
            torch.cuda._sleep(int(_SIDE["delay"]))
        fn()
        ev = torch.cuda.Event()
        ev.record(side)
    pend.append((ev, [t for t in operands if t is not None]))
    _mark_side_used()


def _mark_side_used():
    """the running backward pass ends with the main stream waiting for the side stream (and the queued reductions flushed)"""
    if not _SIDE["used"]:
        _SIDE["used"] = True
        torch.autograd.Variable._execution_engine.queue_callback(sync_side_stream)


_GRAD_ONLY = None          # None | "data" | "param"  (see grad_only)


class grad_only:
    """Context manager for partial backward passes (the adaptive weight of train_favae.py:32-39 needs d loss / d x_recon and
    d x_recon / d last-layer-weight, nothing else): "data" -> conv nodes skip their weight / bias gradients, "param" -> conv
    nodes skip the data gradient (and the GroupNorm backward behind it).  Skipped gradients are returned as None, so the mode
    must only be used around autograd.grad() calls that do not ask for them."""

    def __init__(self, what):
        assert what in ("data", "param")
        self.what = what

    def __enter__(self):
        global _GRAD_ONLY
        self.prev, _GRAD_ONLY = _GRAD_ONLY, self.what

    def __exit__(self, *a):
        global _GRAD_ONLY
        _GRAD_ONLY = self.prev


class no_direct_grad:
    """Context manager: gradients are returned to autograd instead of being accumulated straight into the flat gradient
    buffer -- needed by torch.autograd.grad() calls (the adaptive weight of train_favae.py:32-39), which must neither touch
    .grad nor receive None."""

    def __enter__(self):
        global _DIRECT_GRAD
        self.prev, _DIRECT_GRAD = _DIRECT_GRAD, False

    def __exit__(self, *a):
        global _DIRECT_GRAD
        _DIRECT_GRAD = self.prev


# favae_step.FlatAdam(direct_grads=True) turns this on: a training loop that is not TrainStep may call torch.autograd.grad() on its own
# (the adaptive weight of train_favae.py:32-39) without knowing about no_direct_grad -- every direct accumulation then first asks the
# engine what the running graph task does with that parameter's gradient.  (TrainStep wraps its own autograd.grad() calls and leaves
# the check off: one engine query per parameter and backward pass is host time its step does not need to spend.)
_ENGINE_CHECK = False


def _engine_accumulates(p):
    """does the running graph task ACCUMULATE the gradient of leaf `p` into p.grad (.backward(), backward(inputs=[..p..]))?  False inside
    torch.autograd.grad(): w.r.t. p the engine captures the returned tensor (the query raises for a leaf there), w.r.t. anything else
    p's gradient is not wanted at all -- in both cases .grad must not be touched and `None` must not stand in for a gradient."""
    try:
        return bool(torch._C._will_engine_execute_node(torch.autograd.graph.get_gradient_edge(p).node))
    except RuntimeError:
        return False


def _engine_runs(node):
    """will the running graph task execute `node` (a non-leaf's grad_fn)?  False inside torch.autograd.grad() / backward(inputs=...) for
    a node that leads to none of the requested inputs: a gradient flowing there is dropped by the engine, so it need not be formed."""
    try:
        return node is None or bool(torch._C._will_engine_execute_node(node))
    except RuntimeError:
        return True


def _direct_grad(p):
    """Gradient target for parameter `p` when its .grad is a pre-zeroed view of a flat gradient buffer owned by
    favae_step.TrainStep / FlatAdam (marked `_favae_flat`): the reduction kernels then ACCUMULATE into it and autograd is handed
    None, which removes one AccumulateGrad add kernel per parameter.  None -> ordinary autograd path."""
    if not _DIRECT_GRAD or p is None or not getattr(p, "_favae_flat", False) or p.grad is None:
        return None
    if _ENGINE_CHECK and not _engine_accumulates(p):
        return None
    g = p.grad
    if g.dim() == 4:
        return g if _is_cl(g) else None
    return g if g.is_contiguous() else None


# ---- weight gradients of a training loop that is NOT TrainStep / FlatAdam(direct): ordinary tensors, still on the second stream --------
# A gradient RETURNED to autograd has to be finished in main-stream order: AccumulateGrad (and, under DDP, the reducer's bucket copy)
# consumes it on the main stream right behind the node that returned it -- the engine runs AccumulateGrad nodes first.  Weight gradients
# handed back that way therefore ran on the MAIN stream in rounds 1-5 and such a loop paid the single-stream step (profiles/
# r06_ref_loop.txt: 147.7 ms against 129.7).  Round 6: at the START of a forward pass (VQGANFCM.forward) every dense conv weight is passed
# through an identity node (_LateGradFn); the convs take that alias.  The engine orders ready nodes by creation sequence, latest first,
# so these nodes -- created before anything else of the pass -- run LAST: a conv's backward launches its weight gradient on the side
# stream into a fresh tensor and hands it to its identity node, where it waits untouched until the rest of the backward pass has been
# queued; the first identity node to run makes the main stream wait for the side stream, and only then do AccumulateGrad / DDP's hooks see
# the tensors.  (DDP's bucket all-reduces then all start at the end of backward: 331 MB over xGMI, a few ms, against 15 ms regained.)
# A weight used twice in one pass takes the alias only the first time (two gradients meeting at one node would be added by the engine
# on the main stream while the side stream may still be writing them).  FAVAE_LATE_GRADS=0: off (A/B switch).
_LATE = {"map": {}, "on": os.environ.get("FAVAE_LATE_GRADS", "1") != "0"}


class _LateGradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p):
        return p.view_as(p)

    @staticmethod
    def backward(ctx, g):
        sync_side_stream()                 # main stream behind every weight gradient of the side stream (later calls: nothing left)
        return g


def late_weights(params):
    """Start of a forward pass of a module that owns dense conv weights: aliases (see above) for those of `params` that are ordinary
    trainable 4-D parameters (not views of a flat gradient buffer with direct accumulation: TrainStep / FlatAdam have their own path)."""
    m = {}
    if _LATE["on"] and _SIDE["on"] and torch.is_grad_enabled():
        for p in params:
            if p.requires_grad and p.dim() == 4 and p.is_cuda and not getattr(p, "_favae_flat", False):
                w = _LateGradFn.apply(p)
                w._favae_late = True
                m[id(p)] = [w, False]
    _LATE["map"] = m


def _late_alias(w):
    ent = _LATE["map"].get(id(w)) if _LATE["map"] else None
    if ent is None or ent[1] or not torch.is_grad_enabled():
        return w
    ent[1] = True
    return ent[0]


class ConvCfg:
    """Static description of one fused conv site."""
    __slots__ = ("kh", "kw", "stride", "pad", "pad_br", "upsample", "act", "groups", "eps", "norm")

    def __init__(self, kh, kw, stride=1, pad=0, pad_br=None, upsample=False, act=ACT_SILU, groups=32, eps=1e-5, norm="group"):
        """norm: how the fused input transform act(affine(x)) gets its affine --
        "group": GroupNorm(groups) of x, statistics per image (gn_w/gn_b = its weight/bias);
        "batch": BatchNorm2d batch statistics over (N,H,W) (gn_w/gn_b = its weight/bias; bn_batch_stats() supplies `stats`);
        "act":   no normalisation, the activation alone (gn_w/gn_b = None)."""
        self.kh, self.kw, self.stride, self.pad = kh, kw, stride, pad
        self.pad_br = pad if pad_br is None else pad_br
        self.upsample, self.act, self.groups, self.eps, self.norm = upsample, act, groups, eps, norm

    def out_hw(self, H, W):
        if self.upsample:
            H, W = 2 * H, 2 * W
        Ho = (H + self.pad + self.pad_br - self.kh) // self.stride + 1
        Wo = (W + self.pad + self.pad_br - self.kw) // self.stride + 1
        return Ho, Wo


# GroupNorm-backward pass 1 (two tensor reads per GroupNorm) inside the epilogue of the data-gradient conv (A/B switch)
_GNBWD_FUSE = os.environ.get("FAVAE_GNBWD_FUSE", "1") != "0"
# Bias gradient + operand range of dy as by-products of the GroupNorm-backward apply pass that WRITES dy (A/B switch).  They ride on
# the gradient tensor as `_favae_dycs` = (per-block column sums, blocks, max|dy| device scalar, tensor version); the conv whose
# output gradient the tensor is picks them up when the version counter is unchanged (nothing accumulated into it in place).
_DYCS_FUSE = os.environ.get("FAVAE_DYCS_FUSE", "1") != "0"


def _dy_byproducts(dy, C):
    pre = getattr(dy, "_favae_dycs", None) if _DYCS_FUSE else None
    if (pre is not None and pre[3] == dy._version and pre[0].numel() == pre[1] * C and dy.shape[1] == C
            and pre[4] == _ARENA["epoch"]):
        return pre
    return None


def _bias_grad_and_range(dy, p_b, need_b, want_range, M, Cout, dev):
    """db (or None when accumulated into the flat buffer) and the device scalar max|dy| (None unless asked for): from the
    by-products of the pass that wrote dy when it left them, else one streaming pass over dy (favae_colsum / favae_absmax)."""
    db = None
    dyb = None
    pre = _dy_byproducts(dy, Cout)
    if pre is not None:
        if need_b:
            tgt = _direct_grad(p_b)
            if tgt is not None and _DEFER_REDUCE and _SIDE["on"]:
                # second stage of the column sum (2048 partial rows of C floats: one latency-bound 80 us launch per layer when run
                # alone) joins the grouped slab reductions of the side stream; the partials were written on THIS stream
                side = _side_stream()
                side.wait_stream(torch.cuda.current_stream())
                _defer_reduction(pre[0], tgt, Cout, pre[1], 1)
                _mark_side_used()
            else:
                if tgt is None:
                    db = torch.empty((Cout,), dtype=torch.float32, device=dev)
                else:
                    _order_behind_deferred(tgt)
                call("favae_colsum_finish", ptr(pre[0]), pre[1], Cout, ptr(db if tgt is None else tgt), 0 if tgt is None else 1)
        return db, (pre[2] if want_range else None)
    if want_range:
        dyb = _max_target(dev)
    if (need_b or want_range) and dy.dtype == torch.bfloat16:       # no by-product: one conversion pass in front of the column sums
        dy = cast_f32(dy)
    if need_b:
        ws = workspace(query("favae_colsum_workspace", M, Cout), dev)
        tgt = _direct_grad(p_b)
        if tgt is None:
            db = torch.empty((Cout,), dtype=torch.float32, device=dev)
        else:
            _order_behind_deferred(tgt)
        call("favae_colsum", ptr(dy), ptr(db if tgt is None else tgt), M, Cout, 0 if tgt is None else 1, ptr(dyb), ptr(ws), ws.numel())
    elif want_range:
        call("favae_absmax", ptr(dy), dy.numel(), ptr(dyb))
    return db, dyb


def _order_behind_deferred(tgt):
    """a main-stream accumulation into `tgt` while a deferred (side-stream) reduction into the same range is still queued (a module
    applied twice in one pass): run the queued reductions and wait for them first"""
    if tgt.data_ptr() in _SIDE["targets"]:
        flush_reductions()
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])


def _wino_records(w, co, ci, flip, w_amax):
    """Winograd weight records of the OHWI tensor w (csrc/conv_wino.h): flip = 0 forward conv, 1 its data gradient; + 2: F(4x4, 3x3)
    records (csrc/conv_wino4.h)."""
    wsp = torch.empty(query("favae_wino4_weights_bytes" if flip & 2 else "favae_wino_weights_bytes", co, ci), dtype=torch.uint8, device=w.device)
    call("favae_wino_weights", ptr(w), ptr(wsp), co, ci, flip, ptr(w_amax))
    return wsp


def _conv_launch(d, x, w_ohwi, b, resid, scale, shift, y, x_bound=None, flip_of=None, gnbwd=None, stats_out=None,
                 y_amax=None, wino_rec=None, bf16io=False):
    """conv forward / data gradient.  When the library runs this shape on the split-precision matrix path the weights are
    pre-split once per call (instead of once per tile in the K loop); the fp16 scheme (2 planes) also needs the operand
    range: `x_bound` = device scalar >= max|T(x)| (computed here for an un-transformed operand when not supplied).
    flip_of = (w, Cout, KH, KW, Cin, w_absmax): the weights of this (data-gradient) conv are the flip of the OHWI tensor w;
    w_ohwi may then be None -- records are written by one flip+split kernel that reuses the forward's max|w|.
    Returns the device float holding max|w| when pre-split records were made (fp16 scheme), else None."""
    planes = query("favae_conv_wants_split_weights", byref(d), 0 if scale is None else 1)
    w_amax = None
    if (planes in (1, 2, 4) and _wino1_wanted(planes, flip_of is not None or gnbwd is not None, d.act)
            and query("favae_conv_wino_ok", byref(d), 0 if scale is None else 1)):
        # dense 3x3 conv of the h3 scheme (and, where Cout % 128 == 0, of the one-plane 16-bit modes h1 / b1): Winograd F(2x2, 3x3) kernel,
        # records = G g G^T in fragment order (csrc/conv_wino.h); h1 reads the head plane of the h3 records, b1 has bf16 ones (flip | 4)
        if wino_rec is not None:                 # made by the forward pass (FusedConvFn.forward)
            wsp, f43 = wino_rec
        else:
            if flip_of is not None:
                w, co, kh, kw, ci, w_amax = flip_of
                flip = 1
            else:
                w, co, ci, flip = w_ohwi, d.Cout, d.Cin, 0        # OHWI memory, whatever the logical shape
            f43 = planes == 2 and _wino4_wanted(d, scale is not None, flip == 1)
            if f43:
                flip |= 2
            if planes == 4:
                flip |= 4
            wsp = _wino_cached(w, flip)              # made for the whole model after the optimizer step (WinoRecords)
            if wsp is None:
                if w_amax is None:
                    w_amax = _weight_amax(w)
                wsp = _wino_records(w, co, ci, flip, w_amax)
        w_amax = wsp[:4].view(torch.float32)
        planes |= PLANES_WINO | (PLANES_WINO4 if f43 else 0)
    if planes & PLANES_WINO:
        pass
    elif planes:
        if flip_of is not None and (planes in (3, 4) or flip_of[5] is not None) and os.environ.get("FAVAE_FLIP_SPLIT", "1") != "0":
            w, co, kh, kw, ci, w_amax = flip_of
            n = w.numel()
            wsp = torch.empty(query("favae_split_weights_bytes", n, planes), dtype=torch.uint8, device=w.device)
            call("favae_weight_flip_split", ptr(w), ptr(wsp), co, kh, kw, ci, planes, ptr(w_amax))
        else:
            if w_ohwi is None:
                w_ohwi = _flipped(flip_of)
            n = w_ohwi.numel()
            wsp = torch.empty(query("favae_split_weights_bytes", n, planes), dtype=torch.uint8, device=w_ohwi.device)
            cached = _weight_amax(w_ohwi) if planes in (1, 2) else None
            if cached is not None:               # max|w| refreshed once per optimizer step for all weights (WeightMaxima)
                call("favae_split_weights_amax", ptr(w_ohwi), ptr(wsp), n, planes, ptr(cached))
            else:
                call("favae_split_weights", ptr(w_ohwi), ptr(wsp), n, planes)
            if planes in (1, 2):
                w_amax = wsp[:4].view(torch.float32)          # header of the record buffer (keeps the buffer alive while saved)
    if bf16io:                               # bf16 activation storage: x, resid, y (and the GroupNorm input of gnbwd) are bf16 tensors
        if (planes & 0xFF) != 4:
            raise RuntimeError("bf16 activation storage needs the one-bf16-plane conv mode (b1)")
        planes |= BF16IO_PLANES
    if planes:
        if (planes & 0xFF) in (1, 2) and x_bound is None:
            if scale is not None:
                raise RuntimeError("a transformed conv operand needs its range bound (gn_stats(with_bound=True))")
            x_bound = absmax(x)
        if gnbwd is not None:                # data gradient + GroupNorm-backward partial sums (favae_conv_dgrad_gnbwd)
            gx, gmean, grstd, ggw, ggb, ggroups, gact, gws = gnbwd
            call("favae_conv_dgrad_gnbwd", byref(d), ptr(x), ptr(wsp), planes, ptr(x_bound), ptr(y), ptr(gx), ptr(gmean), ptr(grstd),
                 ptr(ggw), ptr(ggb), ggroups, gact, ptr(gws), gws.numel())
        elif stats_out is not None:          # forward + per-tile statistics of the output (pass 1 of the next GroupNorm) + max|y|
            call("favae_conv_fwd_split_stats", byref(d), ptr(x), ptr(wsp), planes, ptr(x_bound), ptr(b), ptr(resid), ptr(scale),
                 ptr(shift), ptr(y), ptr(stats_out), stats_out.numel() * 8, ptr(y_amax))
        else:
            call("favae_conv_fwd_split", byref(d), ptr(x), ptr(wsp), planes, ptr(x_bound), ptr(b), ptr(resid), ptr(scale),
                 ptr(shift), ptr(y))
    else:
        if gnbwd is not None or stats_out is not None:
            raise RuntimeError("fused GroupNorm sums need the split matrix path")
        if w_ohwi is None:
            w_ohwi = _flipped(flip_of)
        call("favae_conv_fwd", byref(d), ptr(x), ptr(w_ohwi), ptr(b), ptr(resid), ptr(scale), ptr(shift), ptr(y))
    return w_amax


def _flipped(flip_of):
    """wt[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci] as an fp32 tensor (convs that do not take pre-split records)"""
    w, co, kh, kw, ci, _ = flip_of
    wt = torch.empty((ci, kh, kw, co), dtype=torch.float32, device=w.device)
    call("favae_weight_flip", ptr(w), ptr(wt), co, kh, kw, ci)
    return wt


def _bf16_conv_ok(N, Hin, Win, Cin, Ho, Wo, Cout, cfg, has_aff, has_gn, need_grad):
    """bf16 activation storage for this conv: the mode is on and every kernel the conv will run -- forward, data gradient (with the
    GroupNorm-backward epilogue when a GroupNorm sits in front), weight gradient, apply pass -- has the bf16 instantiation"""
    if not bf16_storage() or cfg.norm in ("batch", "act") or cfg.upsample or cfg.stride != 1 or cfg.kh != 3 or cfg.kw != 3:
        return False
    act = cfg.act if has_aff else ACT_NONE
    d = make_conv_desc(N, Hin, Win, Cin, Ho, Wo, Cout, cfg.kh, cfg.kw, cfg.stride, cfg.pad, GATHER_PLAIN, act, 1)
    if not query("favae_conv_bf16io_ok", byref(d), 1 if has_aff else 0, 0):
        return False
    if need_grad:
        d2 = make_conv_desc(N, Ho, Wo, Cout, Hin, Win, Cin, cfg.kh, cfg.kw, 1, cfg.kh - 1 - cfg.pad, GATHER_PLAIN, ACT_NONE, 1)
        if not query("favae_conv_bf16io_ok", byref(d2), 0, 1 if has_gn else 0) or not query("favae_conv_bf16io_ok", byref(d), 1 if has_aff else 0, 2):
            return False
        if has_gn and not query("favae_gn_bwd_colsum_blocks", N, Hin * Win, Cin):
            return False
    return True


class FusedConvFn(torch.autograd.Function):
    """y = conv(act(GN(x)), w) + b + resid   -- GN/act optional (gn_w is None -> plain conv).

    Reference sites: ResnetBlock/NonResnetBlock halves (models/codec.py:38-46), Downsample (:26-29), Upsample (:17-18),
    conv_in/final (:140,170-175), attention in/out projections (:92).

    pass_input=True additionally returns x itself as a second output.  A caller that also feeds x to a later op (the skip
    connection of a ResnetBlock / AttnBlock) uses that alias instead of x: the skip gradient then arrives HERE, next to dy,
    and is added inside the GroupNorm-backward kernel (dx_add) instead of by a separate autograd accumulation kernel."""

    @staticmethod
    def forward(ctx, x, w, b, gn_w, gn_b, resid, cfg, pass_input=False, stats=None):
        N, Cin, Hin, Win = x.shape
        w4 = w if w.dim() == 4 else w.view(w.shape[0], w.shape[1], 1, 1)
        wk = weight_ohwi(w4)
        Cout = w4.shape[0]
        Ho, Wo = cfg.out_hw(Hin, Win)
        # bf16 activation storage (b1 mode): x, resid and y of this conv are bf16 tensors when all of its kernels can take them
        st = _bf16_conv_ok(N, Hin, Win, Cin, Ho, Wo, Cout, cfg, cfg.norm in ("batch", "act") or gn_w is not None, gn_w is not None,
                           any(ctx.needs_input_grad))
        x = to_cl(x, keep_bf16=st)
        if st:
            x = cast_bf16(x)
        dev = x.device
        mean = rstd = scale = shift = xb = None
        per_image = 1
        if cfg.norm == "batch":                   # BatchNorm batch statistics, computed by bn_batch_stats() (running stats there)
            mean, rstd, scale, shift, xb = stats
            per_image = 0
        elif cfg.norm == "act":                   # activation only: identity affine shared by all images
            scale = torch.ones((1, Cin), dtype=torch.float32, device=dev)
            shift = torch.zeros((1, Cin), dtype=torch.float32, device=dev)
            xb = absmax(x) if _fp16_planes() else None           # |act(x)| <= |x| for SiLU / LeakyReLU
            per_image = 0
        elif gn_w is not None:
            mean, rstd, scale, shift, xb = gn_stats(x, gn_w, gn_b, cfg.groups, cfg.eps, with_bound=True)
        if resid is not None:
            resid = to_cl(resid, keep_bf16=st)
            if st:
                resid = cast_bf16(resid)
        y = new_cl(N, Cout, Ho, Wo, dev, torch.bfloat16 if st else torch.float32)
        d = make_conv_desc(N, Hin, Win, Cin, Ho, Wo, Cout, cfg.kh, cfg.kw, cfg.stride, cfg.pad,
                           GATHER_UPSAMPLE2 if cfg.upsample else GATHER_PLAIN, cfg.act if scale is not None else ACT_NONE,
                           per_image)
        if xb is None and query("favae_conv_wants_split_weights", byref(d), 0) in (1, 2):
            xb = absmax(x)
        # per-tile (sum y, sum y^2) of the output for the GroupNorm that will consume it (almost every 3x3 conv feeds one)
        st_tiles = 0
        st_part = None
        if _GNSTATS_FUSE:
            # the tile grid is the kernel's: F(4x4) records (decoder forward under FAVAE_WINO4=2) have their own
            st_tiles = query("favae_conv_stats_tiles", byref(d), 0 if scale is None else 1,
                             PLANES_WINO4 if _wino4_wanted(d, scale is not None, False) else 0)
            if st_tiles:
                st_part = torch.empty((N * st_tiles * Cout * 2,), dtype=torch.float64, device=dev)
        y_amax = _max_target(dev) if st_part is not None else None
        w_amax = _conv_launch(d, x, wk, b, resid, scale, shift, y, xb, stats_out=st_part, y_amax=y_amax, bf16io=st)
        if st_part is not None:
            y._favae_gnstats = (st_part, st_tiles, y._version)
            y._favae_amax = (y_amax, y._version, _ARENA["epoch"])
        # Winograd records of the data gradient, made HERE: in the backward pass this small kernel would sit on the critical chain
        # next to the weight-gradient stream (measured 72 us per layer there against 9 us alone)
        ctx.wflip = None
        if (_WINO_FLIP_FWD and w_amax is not None and cfg.stride == 1 and not cfg.upsample and cfg.kh == 3
                and any(ctx.needs_input_grad) and (ctx.needs_input_grad[0] or gn_w is not None)):   # no backward (no_grad, eval): no records
            d2 = make_conv_desc(N, Ho, Wo, Cout, Hin, Win, Cin, cfg.kh, cfg.kw, 1, cfg.kh - 1 - cfg.pad, GATHER_PLAIN, ACT_NONE, 1)
            pl2 = query("favae_conv_wants_split_weights", byref(d2), 0)
            if pl2 in (1, 2, 4) and _wino1_wanted(pl2, True) and query("favae_conv_wino_ok", byref(d2), 0):
                f43 = pl2 == 2 and _wino4_wanted(d2, False, True)
                key = (3 if f43 else 1) | (4 if pl2 == 4 else 0)
                rec = _wino_cached(wk, key)
                if rec is None:
                    rec = _wino_records(wk, Cout, Cin, key, w_amax)
                ctx.wflip = (rec, f43)
        ctx.cfg = cfg
        ctx.st = st
        ctx.arena_epoch = _ARENA["epoch"]
        ctx.has_b = b is not None
        ctx.has_gn = gn_w is not None
        ctx.has_xform = scale is not None
        ctx.per_image = per_image
        ctx.has_res = resid is not None
        ctx.w_dim = w.dim()
        ctx.params = (w, b, gn_w, gn_b)           # to reach pre-assigned flat-buffer gradients (see _direct_grad)
        ctx.save_for_backward(x, wk, gn_w, gn_b, mean, rstd, scale, shift, xb, w_amax)
        if pass_input:
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, wk, gn_w, gn_b, mean, rstd, scale, shift, xb, w_amax = ctx.saved_tensors
        _check_arena_epoch(ctx, xb)
        st = ctx.st                               # bf16 activation storage: dy, da, dskip, dx are bf16 tensors like x
        adt = torch.bfloat16 if st else torch.float32
        if dskip is not None:
            dskip = to_cl(dskip, keep_bf16=st)
            if st:
                dskip = cast_bf16(dskip)
        cfg = ctx.cfg
        dy = to_cl(dy, keep_bf16=st)
        if st:
            dy = cast_bf16(dy)                    # (the engine hands dy over in y's dtype: a no-op)
        N, Cin, Hin, Win = x.shape
        _, Cout, Ho, Wo = dy.shape
        dev = x.device
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_b and ctx.needs_input_grad[2]
        has_gn = ctx.has_gn
        if _GRAD_ONLY == "data":
            need_w = need_b = False
        elif _GRAD_ONLY == "param":
            need_x = has_gn = False
        dx = dw = db = dgw = dgb = None
        gather = GATHER_UPSAMPLE2 if cfg.upsample else GATHER_PLAIN
        act = cfg.act if ctx.has_xform else ACT_NONE
        p_w, p_b, p_gw, p_gb = ctx.params
        # bias gradient + range of dy for the fp16 split scheme: by-products of the pass that wrote dy, else one pass over dy
        want_range = xb is not None and _fp16_planes()
        db, dyb = _bias_grad_and_range(dy, p_b, need_b, want_range, N * Ho * Wo, Cout, dev)
        # The weight gradient is launched AFTER this conv's data gradient (see run_wgrad below)
        run_wgrad = None
        if need_w and getattr(p_w, "_favae_late", False) and not _engine_runs(p_w.grad_fn):
            # torch.autograd.grad() for something else (the adaptive weight, train_favae.py:32-39): needs_input_grad says "requires
            # grad", the engine knows that nobody waits for this one
            need_w = False
        if need_w:
            d = make_conv_desc(N, Hin, Win, Cin, Ho, Wo, Cout, cfg.kh, cfg.kw, cfg.stride, cfg.pad, gather, act, ctx.per_image)
            ws = workspace(query("favae_conv_wgrad_workspace", byref(d)), dev)
            tgt = _direct_grad(p_w)
            dwk = None
            if tgt is None:
                dwk = torch.empty((Cout, cfg.kh, cfg.kw, Cin), dtype=torch.float32, device=dev)
                dw = dwk.permute(0, 3, 1, 2)                  # (Cout,Cin,KH,KW) view with channels-last strides
                if ctx.w_dim == 2:
                    dw = dwk.view(Cout, Cin)
            wd = d if not st else make_conv_desc(N, Hin, Win, Cin, Ho, Wo, Cout, cfg.kh, cfg.kw, cfg.stride, cfg.pad, gather,
                                                 act | BF16IO_ACT, ctx.per_image)          # x and dy are bf16 (include/favae_hip.h)
            wws, wtgt = ws, (tgt if tgt is not None else dwk)
            acc = 1 if tgt is not None else 0

            defer = _DEFER_REDUCE and tgt is not None and _SIDE["on"]

            def launch_wgrad():
                if defer:                                     # slabs stay in the workspace until flush_reductions()
                    slabs = ctypes.c_int(0)
                    call("favae_conv_wgrad_slabs", byref(wd), ptr(x), ptr(dy), ptr(scale), ptr(shift), ptr(xb), ptr(dyb), ptr(wws),
                         wws.numel(), byref(slabs))
                    _defer_reduction(wws, wtgt, Cout * cfg.kh * cfg.kw * Cin, slabs.value, acc)
                else:
                    call("favae_conv_wgrad", byref(wd), ptr(x), ptr(dy), ptr(scale), ptr(shift), ptr(xb), ptr(dyb), ptr(wtgt), acc,
                         ptr(wws), wws.numel())

            late = tgt is None and getattr(p_w, "_favae_late", False) and _SIDE["on"]
            if late:                                          # ordinary gradient tensor, delivered at the end of backward (_LateGradFn)
                def run_wgrad():
                    _side_launch(launch_wgrad, (x, dy, scale, shift, xb, dyb, wws, dwk))
            elif tgt is not None and _SIDE["on"]:             # flat gradient buffer, side stream (see _SIDE)
                # launched AFTER this conv's data gradient (below): the side stream then starts it next to the HBM-bound
                # GroupNorm-backward / bias-gradient kernels that follow instead of next to the other matrix-bound kernel
                # (measured: 204 -> 196 ms/step; launching it before the data gradient only gave 208 -> 204)
                def run_wgrad():
                    _side_launch(launch_wgrad, (x, dy, scale, shift, xb, dyb, wws))
            else:
                run_wgrad = launch_wgrad
        if need_x or has_gn:
            if has_gn and mean is None:
                raise RuntimeError("gradient through a normalisation with frozen (running) statistics is not implemented")
            phased = None
            if (cfg.stride == 2 and not cfg.upsample and cfg.kh == 3 and cfg.kw == 3 and cfg.pad == 0 and cfg.pad_br == 1
                    and Hin % 2 == 0 and Win % 2 == 0 and Cout % 16 == 0 and _SUBPIXEL_DGRAD):
                # Downsample: data gradient by output parity (favae_downsample_dgrad_weights) instead of a zero-dilated input
                d00 = make_conv_desc(N, Ho, Wo, Cout, Ho, Wo, Cin, 2, 2, 1, 1, GATHER_PLAIN, ACT_NONE, 1, lattice=(2, 1, 0, 0))
                planes = query("favae_conv_wants_split_weights", byref(d00), 0)
                if planes in (3, 4) or (planes in (1, 2) and w_amax is not None):
                    phased = planes
            if phased:
                planes = phased
                da = new_cl(N, Cin, Hin, Win, dev)
                wph = torch.empty(query("favae_split_weights_bytes", 9 * Cin * Cout, planes), dtype=torch.uint8, device=dev)
                call("favae_downsample_dgrad_weights", ptr(wk), ptr(wph), Cout, Cin, planes, ptr(w_amax))
                rec = (Cin * Cout // 4) * _WREC[planes]
                if planes in (1, 2) and dyb is None:
                    dyb = absmax(dy)
                for ph, (khn, kwn, off) in enumerate(((2, 2, 0), (2, 1, 4), (1, 2, 6), (1, 1, 8))):
                    ph_pad, pw_pad = khn - 1, kwn - 1
                    dph = make_conv_desc(N, Ho, Wo, Cout, Ho, Wo, Cin, khn, kwn, 1, ph_pad, GATHER_PLAIN, ACT_NONE, 1,
                                         lattice=(2, 1, ph >> 1, ph & 1), pad_dw=pw_pad - ph_pad, w_rec_offset=off * rec)
                    call("favae_conv_fwd_split", byref(dph), ptr(dy), ptr(wph), planes, ptr(dyb), None, None, None, None, ptr(da))
            elif cfg.stride == 1:
                Hv, Wv = (2 * Hin, 2 * Win) if cfg.upsample else (Hin, Win)
                g2, pad2 = GATHER_PLAIN, cfg.kh - 1 - cfg.pad
            elif cfg.stride == 2 and not cfg.upsample:
                Hv, Wv = Hin, Win
                g2, pad2 = GATHER_DILATE2, cfg.kh - 1 - cfg.pad
            else:
                raise RuntimeError("unsupported conv geometry for the data gradient")
            gn_tiles, gn_ws, act_gn = 0, None, act
            if not phased:
                da = new_cl(N, Cin, Hv, Wv, dev, adt)
                d2 = make_conv_desc(N, Ho, Wo, Cout, Hv, Wv, Cin, cfg.kh, cfg.kw, 1, pad2, g2, ACT_NONE, 1)
                gnb = None
                if (_GNBWD_FUSE and has_gn and cfg.norm == "group" and not cfg.upsample and mean is not None
                        and (dyb is not None or not _fp16_planes())):
                    # the tile grid is the kernel's, and the kernel is the one the records were made for AT FORWARD TIME (ctx.wflip):
                    # a set_wino4() between forward and backward must not change the grid under the records
                    f43_now = ctx.wflip[1] if ctx.wflip is not None else _wino4_wanted(d2, False, True)
                    gn_tiles = query("favae_conv_gnbwd_tiles", byref(d2), PLANES_WINO4 if f43_now else 0)
                    if gn_tiles:
                        gn_ws = workspace(query("favae_gn_bwd_tiles_workspace", N, gn_tiles, Cin), dev)
                        gnb = (x, mean, rstd, gn_w, gn_b, cfg.groups, act_gn, gn_ws)
                if _SERIALIZE_MFMA and gn_tiles and _SIDE["used"]:
                    # experiment: the dense data gradient starts only after the weight gradient of the layer behind it has retired
                    torch.cuda.current_stream().wait_stream(_SIDE["stream"])
                _conv_launch(d2, dy, None, None, None, None, None, da, dyb, flip_of=(wk, Cout, cfg.kh, cfg.kw, Cin, w_amax),
                             gnbwd=gnb, wino_rec=ctx.wflip, bf16io=st)
            if run_wgrad is not None:
                run_wgrad()
                run_wgrad = None
            if cfg.upsample:
                dlow = new_cl(N, Cin, Hin, Win, dev)
                call("favae_upsample2x_bwd", ptr(da), ptr(dlow), N, Hin, Win, Cin)
                da = dlow
            if has_gn:
                dx = new_cl(N, Cin, Hin, Win, dev, adt)
                tg, tb = _direct_grad(p_gw), _direct_grad(p_gb)
                direct = tg is not None and tb is not None
                if not direct:
                    dgw = torch.empty((Cin,), dtype=torch.float32, device=dev)
                    dgb = torch.empty_like(dgw)
                # BatchNorm = GroupNorm with one channel per group over the batch folded into the pixel dimension
                gN, gHW, gG = (1, N * Hin * Win, Cin) if cfg.norm == "batch" else (N, Hin * Win, cfg.groups)
                # (FAVAE_DYCS_FUSE=0 runs the apply pass without its column-sum epilogue.  In round 4 that variant returned wrong first
                # components next to the weight-gradient stream: a store-data hazard of its 128-bit buffer stores, fixed in common.h bstore)
                cs_blocks = query("favae_gn_bwd_colsum_blocks", gN, gHW, Cin) if _DYCS_FUSE else 0
                if not gn_tiles:
                    gn_ws = workspace(query("favae_gn_workspace", gN, gHW, Cin), dev)
                act_ap = act
                if st:
                    act_ap |= BF16IO_ACT                      # da, x, dskip, dx are bf16 tensors (include/favae_hip.h FAVAE_ACT_BF16IO)
                if cs_blocks:                                 # the apply pass also leaves colsum / max|dx| for the conv in front
                    cs_part = torch.empty((cs_blocks * Cin,), dtype=torch.float32, device=dev)
                    cs_amax = torch.empty((1,), dtype=torch.float32, device=dev)
                    call("favae_gn_act_bwd_colsum", ptr(da), ptr(x), ptr(gn_w), ptr(gn_b), ptr(mean), ptr(rstd), gN, gHW, Cin, gG, act_ap,
                         ptr(dskip), ptr(dx), ptr(tg if direct else dgw), ptr(tb if direct else dgb), 1 if direct else 0, gn_tiles,
                         ptr(gn_ws), gn_ws.numel(), ptr(cs_part), ptr(cs_amax))
                    dx._favae_dycs = (cs_part, cs_blocks, cs_amax, dx._version, _ARENA["epoch"])
                elif gn_tiles:                                # pass 1 came out of the data-gradient conv's epilogue
                    call("favae_gn_act_bwd_tiles", ptr(da), ptr(x), ptr(gn_w), ptr(gn_b), ptr(mean), ptr(rstd), gN, gHW, Cin,
                         gG, act_ap, ptr(dskip), ptr(dx), ptr(tg if direct else dgw), ptr(tb if direct else dgb), 1 if direct else 0,
                         gn_tiles, ptr(gn_ws), gn_ws.numel())
                else:
                    call("favae_gn_act_bwd", ptr(da), ptr(x), ptr(gn_w), ptr(gn_b), ptr(mean), ptr(rstd), gN, gHW, Cin,
                         gG, act | (BF16IO_ACT if st else 0), ptr(dskip), ptr(dx), ptr(tg if direct else dgw), ptr(tb if direct else dgb),
                         1 if direct else 0, ptr(gn_ws), gn_ws.numel())
                dskip = None                                  # consumed by the kernel (dx = GN-backward + dskip)
            elif ctx.has_xform:                               # activation without normalisation
                dx = new_cl(N, Cin, Hin, Win, dev)
                call("favae_act_bwd", ptr(da), ptr(x), act, da.numel(), ptr(dx))
            else:
                dx = da
        if run_wgrad is not None:                             # no data gradient asked for: nothing to wait for
            run_wgrad()
        if dskip is not None:
            dx = dskip if dx is None else dx + dskip
        dres = dy if ctx.has_res else None
        return dx, dw, db, dgw, dgb, dres, None, None, None


def _phase_desc(N, H, W, Cin, Cout, ph, rec_off=0, dgrad=False):
    """One of the four phase convs of an Upsample (include/favae_hip.h, favae_conv_desc.lat_*): output parity (py, px)."""
    py, px = ph >> 1, ph & 1
    if not dgrad:      # forward / weight gradient: dense low-resolution x -> every second pixel of y (offset py, px)
        return make_conv_desc(N, H, W, Cin, H, W, Cout, 2, 2, 1, 1 - py, GATHER_PLAIN, ACT_NONE, 1, lattice=(2, 1, py, px),
                              pad_dw=py - px, w_rec_offset=rec_off)
    # data gradient: every second pixel of dy (Cout channels) -> dense dx (Cin channels), flipped 2x2 weights
    return make_conv_desc(N, H, W, Cout, H, W, Cin, 2, 2, 1, py, GATHER_PLAIN, ACT_NONE, 1, lattice=(2, 2, py, px), pad_dw=px - py)


class UpsampleConvFn(torch.autograd.Function):
    """y = conv3x3(nearest_x2(x), w) + b  (Upsample, models/codec.py:11-18) as four phase-wise 2x2 convolutions of the
    low-resolution input with summed weights: 16 instead of 36 multiply-adds per low-resolution pixel in the forward pass, the
    data gradient (which lands directly on the low-resolution grid: no 2x2 gather of a full-resolution gradient) and the
    weight gradient.  The weights are summed in fp32 before the products instead of after: rounding-level differences only."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = to_cl(x)
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        dev = x.device
        wk = weight_ohwi(w)
        weff = torch.empty((16 * Cout * Cin,), dtype=torch.float32, device=dev)
        call("favae_upsample_weights", ptr(wk), ptr(weff), Cout, Cin)
        planes = query("favae_conv_wants_split_weights", byref(_phase_desc(N, H, W, Cin, Cout, 0)), 0)
        wsp = torch.empty(query("favae_split_weights_bytes", weff.numel(), planes), dtype=torch.uint8, device=dev)
        call("favae_split_weights", ptr(weff), ptr(wsp), weff.numel(), planes)
        rec = (Cout * 4 * Cin // 4) * _WREC[planes]
        xb = absmax(x) if planes in (1, 2) else None
        y = new_cl(N, Cout, 2 * H, 2 * W, dev)
        for ph in range(4):
            call("favae_conv_fwd_split", byref(_phase_desc(N, H, W, Cin, Cout, ph, ph * rec)), ptr(x), ptr(wsp), planes, ptr(xb),
                 ptr(b), None, None, None, ptr(y))
        ctx.planes, ctx.has_b, ctx.params = planes, b is not None, (w, b)
        ctx.save_for_backward(x, weff, xb, wsp[:4].view(torch.float32) if planes in (1, 2) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weff, xb, w_amax = ctx.saved_tensors
        dy = to_cl(dy)
        N, Cin, H, W = x.shape
        Cout = dy.shape[1]
        dev = x.device
        planes = ctx.planes
        p_w, p_b = ctx.params
        dx = dw = db = None
        db, dyb = _bias_grad_and_range(dy, p_b, ctx.has_b and ctx.needs_input_grad[2], planes in (1, 2), N * 4 * H * W, Cout, dev)
        late = None
        late_w = getattr(p_w, "_favae_late", False) and _SIDE["on"]       # ordinary gradient, delivered at the end of backward (_LateGradFn)
        if ctx.needs_input_grad[1] and not (late_w and not _engine_runs(p_w.grad_fn)):
            tgt = _direct_grad(p_w)
            dweff = torch.empty((16 * Cout * Cin,), dtype=torch.float32, device=dev)
            d0 = _phase_desc(N, H, W, Cin, Cout, 0)
            wss = workspace(query("favae_conv_wgrad_workspace", byref(d0)), dev)
            dwk = None if tgt is not None else torch.empty((Cout, 3, 3, Cin), dtype=torch.float32, device=dev)

            def wgrad():
                for ph in range(4):
                    call("favae_conv_wgrad", byref(_phase_desc(N, H, W, Cin, Cout, ph)), ptr(x), ptr(dy), None, None, ptr(xb),
                         ptr(dyb), dweff.data_ptr() + ph * 4 * Cout * 4 * Cin, 0, ptr(wss), wss.numel())
                call("favae_upsample_wgrad_fold", ptr(dweff), ptr(tgt if tgt is not None else dwk), Cout, Cin,
                     1 if tgt is not None else 0)
            if (tgt is not None or late_w) and _SIDE["on"]:
                def late():
                    _side_launch(wgrad, (x, dy, xb, dyb, wss, dweff, dwk))
                if dwk is not None:
                    dw = dwk.permute(0, 3, 1, 2)
            else:
                wgrad()
                if dwk is not None:
                    dw = dwk.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[0]:
            dx = new_cl(N, Cin, H, W, dev)
            nph = Cout * 4 * Cin
            for ph in range(4):
                wfl = torch.empty(query("favae_split_weights_bytes", nph, planes), dtype=torch.uint8, device=dev)
                call("favae_weight_flip_split", weff.data_ptr() + ph * nph * 4, ptr(wfl), Cout, 2, 2, Cin, planes, ptr(w_amax))
                call("favae_conv_fwd_split", byref(_phase_desc(N, H, W, Cin, Cout, ph, dgrad=True)), ptr(dy), ptr(wfl), planes,
                     ptr(dyb), None, ptr(dx) if ph else None, None, None, ptr(dx))
        if late is not None:
            late()
        return dx, dw, db


def fused_conv(x, w, b=None, gn_w=None, gn_b=None, resid=None, cfg=None, pass_input=False, stats=None):
    if (cfg.upsample and gn_w is None and resid is None and not pass_input and cfg.kh == 3 and cfg.kw == 3 and cfg.stride == 1
            and cfg.pad == 1 and w.dim() == 4
            and query("favae_conv_subpixel_ok", x.shape[0], x.shape[2], x.shape[3], x.shape[1], w.shape[0])):
        return UpsampleConvFn.apply(x, _late_alias(w), b)
    return FusedConvFn.apply(x, _late_alias(w), b, gn_w, gn_b, resid, cfg, pass_input, stats)


@torch.no_grad()
def bn_batch_stats(x, bn, training):
    """Statistics of nn.BatchNorm2d `bn` for the fused conv that consumes act(bn(x)) (cfg.norm == "batch"): batch statistics
    (+ running-stat update exactly like nn.BatchNorm2d: momentum, unbiased variance) in training mode, running statistics
    otherwise.  Returns (mean, rstd, scale, shift, bound); mean/rstd are None for frozen statistics."""
    x = to_cl(x)
    N, C, H, W = x.shape
    dev = x.device
    if training or not bn.track_running_stats:
        xv = x.permute(0, 2, 3, 1).reshape(1, N * H, W, C).permute(0, 3, 1, 2)       # (1,C,N*H,W) view of the NHWC memory
        mean, rstd, scale, shift, bound = gn_stats(xv, bn.weight, bn.bias, C, bn.eps, with_bound=True)
        if training and bn.track_running_stats:
            call("favae_bn_update_running", ptr(mean), ptr(rstd), C, N * H * W, bn.eps, bn.momentum, ptr(bn.running_mean),
                 ptr(bn.running_var))
            bn.num_batches_tracked += 1
        return mean, rstd, scale, shift, bound
    inv = torch.rsqrt(bn.running_var + bn.eps)
    scale = (bn.weight * inv).reshape(1, C).contiguous()
    shift = (bn.bias - bn.running_mean * bn.weight * inv).reshape(1, C).contiguous()
    bound = absmax(x) * scale.abs().max() + shift.abs().max() if _fp16_planes() else None
    return None, None, scale, shift, bound


class HingeMeanFn(torch.autograd.Function):
    """mean_i h(x_i): mode 0: -x (hinge_g_loss), 1: relu(1-x), 2: relu(1+x)   (losses/hinge.py:5-14)."""

    @staticmethod
    def forward(ctx, x, mode):
        x = x.contiguous()
        out = torch.empty((), dtype=torch.float32, device=x.device)
        ws = workspace(query("favae_reduce_workspace", x.numel()), x.device)
        call("favae_hinge_mean", ptr(x), x.numel(), mode, ptr(out), ptr(ws), ws.numel())
        ctx.mode = mode
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        call("favae_hinge_mean_bwd", ptr(x), ptr(g.contiguous()), x.numel(), ctx.mode, ptr(dx))
        return dx, None


# ---------------------------------------------------------------------------------------------------------------
# LPIPS pieces (losses/lpips.py:17-110): ScalingLayer, 2x2 max pooling, one level of the distance
# ---------------------------------------------------------------------------------------------------------------
class ChannelAffineFn(torch.autograd.Function):
    """(x - shift_c) / scale_c   (ScalingLayer.forward, losses/lpips.py:61-62); shift/scale are buffers (no gradient)."""

    @staticmethod
    def forward(ctx, x, shift, scale):
        x = to_cl(x)
        y = torch.empty_like(x)
        call("favae_channel_affine", ptr(x), ptr(shift), ptr(scale), x.numel(), x.shape[1], ptr(y))
        ctx.save_for_backward(scale)
        return y

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        g = to_cl(g)
        dx = torch.empty_like(g)
        call("favae_channel_affine", ptr(g), None, ptr(scale), g.numel(), g.shape[1], ptr(dx))
        return dx, None, None


class MaxPool2Fn(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=2, stride=2) (torchvision vgg16 features[4,9,16,23] as sliced by losses/lpips.py:88-96)."""

    @staticmethod
    def forward(ctx, x):
        x = to_cl(x)
        N, C, H, W = x.shape
        y = new_cl(N, C, H // 2, W // 2, x.device)
        call("favae_maxpool2", ptr(x), N, H, W, C, ptr(y))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        N, C, H, W = x.shape
        g = to_cl(g)
        dx = new_cl(N, C, H, W, x.device)
        call("favae_maxpool2_bwd", ptr(x), ptr(g), N, H, W, C, ptr(dx))
        return dx


class LpipsLevelFn(torch.autograd.Function):
    """val[n] (+)= mean_hw sum_c w_c (a^ - b^)^2 of one VGG level (losses/lpips.py:44-48).  a, b: PRE-activation features of
    lpips()'s first / second argument (the ReLU is applied here); `val` is the running (N,) sum over levels (None for the
    first level) -- its gradient passes through unchanged.  Gradient flows to b only (the step's lpips(x, x_recon))."""

    @staticmethod
    def forward(ctx, a, b, w, val):
        a, b = to_cl(a), to_cl(b)
        N, C, H, W = b.shape
        out = torch.empty((N,), dtype=torch.float32, device=b.device) if val is None else val.clone()
        wv = w.reshape(-1).contiguous()
        ws = workspace(query("favae_lpips_level_workspace", N), b.device)
        call("favae_lpips_level", ptr(a), ptr(b), ptr(wv), N, H * W, C, ptr(out), 0 if val is None else 1, ptr(ws), ws.numel())
        ctx.save_for_backward(a, b, wv)
        ctx.has_val = val is not None
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, wv = ctx.saved_tensors
        N, C, H, W = b.shape
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            raise RuntimeError("LPIPS: only the second image argument carries a gradient (train_favae.py:77 lpips(x, x_recon))")
        db = None
        if ctx.needs_input_grad[1]:
            db = new_cl(N, C, H, W, b.device)
            call("favae_lpips_level_bwd", ptr(a), ptr(b), ptr(wv), ptr(g.contiguous()), N, H * W, C, ptr(db))
        return None, db, None, (g if ctx.has_val else None)


# ---------------------------------------------------------------------------------------------------------------
# single-head attention core (models/codec.py:92,99): qkv (N,3C,H,W) -> o (N,C,H,W)
# ---------------------------------------------------------------------------------------------------------------
_ATTN_TILED = os.environ.get("FAVAE_ATTN_TILED", "1") != "0"      # 0: the materialised fp32-MFMA attention of round 1 (A/B switch)
_ATTN_CHUNK_ELEMS = 48 << 20        # scores of one query chunk, all images: 192 MB fp32 -- stays in the 256 MB Infinity Cache


def _attn_chunk_rows(N, L):
    rq = _ATTN_CHUNK_ELEMS // (N * L)
    rq = max(128, (rq // 128) * 128)
    return min(L, rq)


class AttnCoreFn(torch.autograd.Function):
    """Single-head attention core softmax(q k^T / sqrt(C)) v on the packed in-projection output (N, 3C, H, W) channels-last
    (F.scaled_dot_product_attention inside nn.MultiheadAttention, models/codec.py:92,99) WITHOUT the (N, L, L) matrices: queries are
    processed in chunks whose score tile (all images) fits the Infinity Cache; only the row log-sum-exp (N, L) is saved and the
    backward pass recomputes the probabilities of a chunk as exp(s - lse) (flash-attention recomputation).  Every
    product runs on the split-precision matrix path (favae_bgemm_sp: fp32-grade products from two scaled fp16 planes): forward
    2 GEMMs (4 L^2 C FLOP per image), backward 5 (10 L^2 C).  Operand ranges: max|qkv| for q, k, v; 1 for the probabilities;
    max|dO|; max|dS| from the point-wise backward kernel.
    f=4 at 256x256 (L = 4096, C = 512, batch 16): 2 x 1.07 GB of saved activations per block less than the materialised version."""

    @staticmethod
    def forward(ctx, qkv):
        qkv = to_cl(qkv)
        N, C3, H, W = qkv.shape
        C, L = C3 // 3, H * W
        dev = qkv.device
        alpha = 1.0 / math.sqrt(C)
        q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * C, qkv.data_ptr() + 8 * C
        o = new_cl(N, C, H, W, dev)
        if not (_ATTN_TILED and _fp16_planes() and C % 4 == 0 and L % 4 == 0):
            P = torch.empty((N, L, L), dtype=torch.float32, device=dev)
            call("favae_bgemm", 0, 0, L, L, C, alpha, q, C3, L * C3, k, C3, L * C3, ptr(P), L, L * L, N, 0)
            call("favae_softmax_rows", ptr(P), ptr(P), N * L, L)
            call("favae_bgemm", 0, 1, L, C, L, 1.0, ptr(P), L, L * L, v, C3, L * C3, ptr(o), C, L * C, N, 0)
            ctx.tiled = False
            ctx.save_for_backward(qkv, P)
            return o
        amax = absmax(qkv)
        one = torch.ones((1,), dtype=torch.float32, device=dev)
        lse = torch.empty((N, L), dtype=torch.float32, device=dev)
        RQ = _attn_chunk_rows(N, L)
        S = torch.empty((N * RQ * L,), dtype=torch.float32, device=dev)
        for r0 in range(0, L, RQ):
            rq = min(RQ, L - r0)
            call("favae_bgemm_sp", 0, 0, rq, L, C, alpha, q + 4 * r0 * C3, C3, L * C3, ptr(amax), k, C3, L * C3, ptr(amax), ptr(S), L,
                 rq * L, N, 0)
            call("favae_softmax_rows_lse", ptr(S), ptr(lse), N * rq, L, rq, r0, L)
            call("favae_bgemm_sp", 0, 1, rq, C, L, 1.0, ptr(S), L, rq * L, ptr(one), v, C3, L * C3, ptr(amax), o.data_ptr() + 4 * r0 * C,
                 C, L * C, N, 0)
        ctx.tiled = True
        ctx.save_for_backward(qkv, o, lse, amax, one)
        return o

    @staticmethod
    def backward(ctx, do):
        do = to_cl(do)
        if not ctx.tiled:
            qkv, P = ctx.saved_tensors
            N, C3, H, W = qkv.shape
            C, L = C3 // 3, H * W
            dev = qkv.device
            alpha = 1.0 / math.sqrt(C)
            dqkv = new_cl(N, C3, H, W, dev)
            q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * C, qkv.data_ptr() + 8 * C
            dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + 4 * C, dqkv.data_ptr() + 8 * C
            # dV[j][c] = sum_i P[i][j] dO[i][c]
            call("favae_bgemm", 1, 1, L, C, L, 1.0, ptr(P), L, L * L, ptr(do), C, L * C, dv, C3, L * C3, N, 0)
            # dP[i][j] = sum_c dO[i][c] V[j][c]
            dP = torch.empty((N, L, L), dtype=torch.float32, device=dev)
            call("favae_bgemm", 0, 0, L, L, C, 1.0, ptr(do), C, L * C, v, C3, L * C3, ptr(dP), L, L * L, N, 0)
            call("favae_softmax_rows_bwd", ptr(P), ptr(dP), ptr(dP), N * L, L, alpha)      # dS (alpha folded in)
            # dQ[i][c] = sum_j dS[i][j] K[j][c] ; dK[j][c] = sum_i dS[i][j] Q[i][c]
            call("favae_bgemm", 0, 1, L, C, L, 1.0, ptr(dP), L, L * L, k, C3, L * C3, dq, C3, L * C3, N, 0)
            call("favae_bgemm", 1, 1, L, C, L, 1.0, ptr(dP), L, L * L, q, C3, L * C3, dk, C3, L * C3, N, 0)
            return dqkv
        qkv, o, lse, amax, one = ctx.saved_tensors
        N, C3, H, W = qkv.shape
        C, L = C3 // 3, H * W
        dev = qkv.device
        alpha = 1.0 / math.sqrt(C)
        dqkv = new_cl(N, C3, H, W, dev)
        q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * C, qkv.data_ptr() + 8 * C
        dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + 4 * C, dqkv.data_ptr() + 8 * C
        dmax = absmax(do)
        delta = torch.empty((N, L), dtype=torch.float32, device=dev)
        call("favae_rowdot", ptr(do), ptr(o), ptr(delta), N * L, C)
        RQ = _attn_chunk_rows(N, L)
        S = torch.empty((N * RQ * L,), dtype=torch.float32, device=dev)
        dP = torch.empty((N * RQ * L,), dtype=torch.float32, device=dev)
        for ci, r0 in enumerate(range(0, L, RQ)):
            rq = min(RQ, L - r0)
            qc, doc = q + 4 * r0 * C3, do.data_ptr() + 4 * r0 * C
            dsmax = _max_target(dev)                 # max|dS| of THIS chunk (an arena slot is zero once: one per chunk)
            # scores of the chunk again, dP = dO V^T, then in place: S <- P = exp(S - lse), dP <- dS = alpha P (dP - delta)
            call("favae_bgemm_sp", 0, 0, rq, L, C, alpha, qc, C3, L * C3, ptr(amax), k, C3, L * C3, ptr(amax), ptr(S), L, rq * L, N, 0)
            call("favae_bgemm_sp", 0, 0, rq, L, C, 1.0, doc, C, L * C, ptr(dmax), v, C3, L * C3, ptr(amax), ptr(dP), L, rq * L, N, 0)
            call("favae_attn_bwd_point", ptr(S), ptr(dP), ptr(lse), ptr(delta), N * rq, L, rq, r0, L, alpha, ptr(dsmax))
            acc = 1 if ci else 0
            # dV[j][c] (+)= sum_i P[i][j] dO[i][c]        (contraction over the chunk's queries: both operands [k][rows])
            call("favae_bgemm_sp", 1, 1, L, C, rq, 1.0, ptr(S), L, rq * L, ptr(one), doc, C, L * C, ptr(dmax), dv, C3, L * C3, N, acc)
            # dQ[i][c] = sum_j dS[i][j] K[j][c]
            call("favae_bgemm_sp", 0, 1, rq, C, L, 1.0, ptr(dP), L, rq * L, ptr(dsmax), k, C3, L * C3, ptr(amax), dq + 4 * r0 * C3, C3,
                 L * C3, N, 0)
            # dK[j][c] (+)= sum_i dS[i][j] Q[i][c]
            call("favae_bgemm_sp", 1, 1, L, C, rq, 1.0, ptr(dP), L, rq * L, ptr(dsmax), qc, C3, L * C3, ptr(amax), dk, C3, L * C3, N, acc)
        return dqkv


# ---------------------------------------------------------------------------------------------------------------
# attention FCM (TransEncoderBlock, models/codec.py:108-122): materialised GroupNorm, LayerNorm, dropout, 8-head attention
# ---------------------------------------------------------------------------------------------------------------
_DROP = {"seed": 0, "calls": 0}


def set_dropout_seed(seed):
    """Base seed of the counter-based dropout masks (favae_dropout): call once per training step with a fresh value (TrainStep
    uses its step count).  Every dropout site of the following forward takes the next derived seed; the backward regenerates the
    mask from the seed its forward saved.  The oracle derives the same sequence (favae_oracle.DropoutState)."""
    _DROP["seed"] = int(seed) & 0xFFFFFFFF
    _DROP["calls"] = 0


def _next_dropout_seed():
    _DROP["calls"] += 1
    return (_DROP["seed"] * 0x9E3779B1 + 0x85EBCA6B * _DROP["calls"]) & 0xFFFFFFFF


def _dense(t):
    return to_cl(t) if t.dim() == 4 else t.contiguous()


class DropoutFn(torch.autograd.Function):
    """y = keep(seed, i) [and x > 0] ? x / (1 - p) : 0 over the tensor's memory order (nn.Dropout in train mode; relu=True fuses
    the ReLU in front of it: linear2(dropout(relu(linear1(x)))) of the encoder layer's feed-forward block)."""

    @staticmethod
    def forward(ctx, x, p, seed, relu):
        x = _dense(x)
        _require_gpu(x)
        y = torch.empty_like(x)
        call("favae_dropout", ptr(x), ptr(x) if relu else None, ptr(y), x.numel(), float(p), seed)
        ctx.p, ctx.seed, ctx.relu = float(p), seed, relu
        ctx.save_for_backward(x if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _dense(dy)
        dx = torch.empty_like(dy)
        call("favae_dropout", ptr(dy), ptr(x) if ctx.relu else None, ptr(dx), dy.numel(), ctx.p, ctx.seed)
        return dx, None, None, None


def dropout(x, p, training, relu=False):
    """nn.Dropout(p) (optionally behind a ReLU).  Eval mode / p == 0: identity (or the plain ReLU)."""
    p = float(p) if training else 0.0
    if p == 0.0 and not relu:
        return x
    return DropoutFn.apply(x, p, _next_dropout_seed() if p > 0.0 else 0, relu)


class GNApplyFn(torch.autograd.Function):
    """y = act(GroupNorm(x)) materialised (TransEncoderBlock.norm, whose output is also the block's residual branch; the
    GN-SiLU in front of a training-mode Dropout).  Statistics and backward are the kernels of the fused path."""

    @staticmethod
    def forward(ctx, x, gn_w, gn_b, groups, eps, act):
        x = to_cl(x)
        _require_gpu(x)
        N, C, H, W = x.shape
        mean, rstd, scale, shift = gn_stats(x, gn_w, gn_b, groups, eps)
        y = new_cl(N, C, H, W, x.device)
        call("favae_affine_rows", ptr(x), ptr(scale), ptr(shift), ptr(y), N, H * W, C, act)
        ctx.groups, ctx.act, ctx.params = groups, act, (gn_w, gn_b)
        ctx.save_for_backward(x, gn_w, gn_b, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gn_w, gn_b, mean, rstd = ctx.saved_tensors
        dy = to_cl(dy)
        N, C, H, W = x.shape
        dev = x.device
        dx = new_cl(N, C, H, W, dev)
        tg, tb = _direct_grad(ctx.params[0]), _direct_grad(ctx.params[1])
        direct = tg is not None and tb is not None
        dgw = dgb = None
        if not direct:
            dgw = torch.empty((C,), dtype=torch.float32, device=dev)
            dgb = torch.empty_like(dgw)
        ws = workspace(query("favae_gn_workspace", N, H * W, C), dev)
        call("favae_gn_act_bwd", ptr(dy), ptr(x), ptr(gn_w), ptr(gn_b), ptr(mean), ptr(rstd), N, H * W, C, ctx.groups, ctx.act,
             None, ptr(dx), ptr(tg if direct else dgw), ptr(tb if direct else dgb), 1 if direct else 0, ptr(ws), ws.numel())
        return dx, dgw, dgb, None, None, None


def gn_apply(x, gn_w, gn_b, groups=32, eps=1e-5, act=ACT_NONE):
    return GNApplyFn.apply(x, gn_w, gn_b, groups, eps, act)


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm(C) over the channel dimension of an (N, C, H, W) channels-last tensor = over each token's C features
    (norm1 / norm2 of nn.TransformerEncoderLayer with batch_first tokens (B, HW, C): the same memory)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = to_cl(x)
        _require_gpu(x)
        N, C, H, W = x.shape
        rows = N * H * W
        dev = x.device
        y = new_cl(N, C, H, W, dev)
        mean = torch.empty((rows,), dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        call("favae_layernorm_fwd", ptr(x), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, C, float(eps))
        ctx.params = (w, b)
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dy = to_cl(dy)
        N, C, H, W = x.shape
        rows = N * H * W
        dev = x.device
        dx = new_cl(N, C, H, W, dev)
        t = new_cl(N, C, H, W, dev)
        call("favae_layernorm_bwd", ptr(dy), ptr(x), ptr(w), ptr(mean), ptr(rstd), ptr(dx), ptr(t), rows, C)
        out = []
        for src, prm in ((t, ctx.params[0]), (dy, ctx.params[1])):           # dgamma = colsum(dy * xhat), dbeta = colsum(dy)
            ws = workspace(query("favae_colsum_workspace", rows, C), dev)
            tgt = _direct_grad(prm)
            g = None
            if tgt is None:
                g = torch.empty((C,), dtype=torch.float32, device=dev)
            call("favae_colsum", ptr(src), ptr(g if tgt is None else tgt), rows, C, 0 if tgt is None else 1, None, ptr(ws), ws.numel())
            out.append(g)
        return dx, out[0], out[1], None


def layer_norm(x, w, b, eps=1e-5):
    return LayerNormFn.apply(x, w, b, eps)


class MHACoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh)) [dropout] v per head, on the packed in-projection output (N, 3C, H, W) channels-last =
    tokens (N, L, 3C); head h owns features [h dh, (h+1) dh) of each of q, k, v (nn.MultiheadAttention, batch_first).  One
    batched GEMM per head and product (head = pointer offset + row stride 3C); the probabilities of all heads share one
    (N, heads, L, L) buffer so softmax / dropout / softmax-backward are single launches."""

    @staticmethod
    def forward(ctx, qkv, heads, p, seed):
        qkv = to_cl(qkv)
        _require_gpu(qkv)
        N, C3, H, W = qkv.shape
        C, L = C3 // 3, H * W
        dh = C // heads
        dev = qkv.device
        alpha = 1.0 / math.sqrt(dh)
        P = torch.empty((N, heads, L, L), dtype=torch.float32, device=dev)
        base = qkv.data_ptr()
        for h in range(heads):
            q, k = base + 4 * h * dh, base + 4 * (C + h * dh)
            call("favae_bgemm", 0, 0, L, L, dh, alpha, q, C3, L * C3, k, C3, L * C3, P.data_ptr() + 4 * h * L * L, L, heads * L * L, N, 0)
        call("favae_softmax_rows", ptr(P), ptr(P), N * heads * L, L)
        Pd = P
        if p > 0.0:
            Pd = torch.empty_like(P)
            call("favae_dropout", ptr(P), None, ptr(Pd), P.numel(), float(p), seed)
        o = new_cl(N, C, H, W, dev)
        for h in range(heads):
            v = base + 4 * (2 * C + h * dh)
            call("favae_bgemm", 0, 1, L, dh, L, 1.0, Pd.data_ptr() + 4 * h * L * L, L, heads * L * L, v, C3, L * C3,
                 o.data_ptr() + 4 * h * dh, C, L * C, N, 0)
        ctx.heads, ctx.p, ctx.seed = heads, float(p), seed
        ctx.save_for_backward(qkv, P, Pd if p > 0.0 else None)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, P, Pd = ctx.saved_tensors
        if Pd is None:
            Pd = P
        do = to_cl(do)
        heads = ctx.heads
        N, C3, H, W = qkv.shape
        C, L = C3 // 3, H * W
        dh = C // heads
        dev = qkv.device
        alpha = 1.0 / math.sqrt(dh)
        dqkv = new_cl(N, C3, H, W, dev)
        base, dbase, dob = qkv.data_ptr(), dqkv.data_ptr(), do.data_ptr()
        dP = torch.empty_like(P)
        sP = heads * L * L
        for h in range(heads):
            v, dv = base + 4 * (2 * C + h * dh), dbase + 4 * (2 * C + h * dh)
            ph, doh = 4 * h * L * L, dob + 4 * h * dh
            call("favae_bgemm", 1, 1, L, dh, L, 1.0, Pd.data_ptr() + ph, L, sP, doh, C, L * C, dv, C3, L * C3, N, 0)
            call("favae_bgemm", 0, 0, L, L, dh, 1.0, doh, C, L * C, v, C3, L * C3, dP.data_ptr() + ph, L, sP, N, 0)
        if ctx.p > 0.0:
            call("favae_dropout", ptr(dP), None, ptr(dP), dP.numel(), ctx.p, ctx.seed)
        call("favae_softmax_rows_bwd", ptr(P), ptr(dP), ptr(dP), N * heads * L, L, alpha)
        for h in range(heads):
            q, k = base + 4 * h * dh, base + 4 * (C + h * dh)
            dq, dk = dbase + 4 * h * dh, dbase + 4 * (C + h * dh)
            ph = 4 * h * L * L
            call("favae_bgemm", 0, 1, L, dh, L, 1.0, dP.data_ptr() + ph, L, sP, k, C3, L * C3, dq, C3, L * C3, N, 0)
            call("favae_bgemm", 1, 1, L, dh, L, 1.0, dP.data_ptr() + ph, L, sP, q, C3, L * C3, dk, C3, L * C3, N, 0)
        return dqkv, None, None, None


def mha_core(qkv, heads, p, training):
    p = float(p) if training else 0.0
    return MHACoreFn.apply(qkv, heads, p, _next_dropout_seed() if p > 0.0 else 0)


# ---------------------------------------------------------------------------------------------------------------
# learnable-sigma Gaussian blur (models/codec.py:255-277)
# ---------------------------------------------------------------------------------------------------------------
class BlurFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sigmas, index, ksize):
        x = to_cl(x)
        _require_gpu(sigmas)
        N, C, H, W = x.shape
        y = new_cl(N, C, H, W, x.device)
        sp = sigmas.data_ptr() + 4 * index
        call("favae_blur_fwd", ptr(x), sp, ksize, N, H, W, C, ptr(y))
        ctx.save_for_backward(x, sigmas)
        ctx.index, ctx.ksize = index, ksize
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sigmas = ctx.saved_tensors
        dy = to_cl(dy)
        N, C, H, W = x.shape
        dev = x.device
        need_x, need_s = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = new_cl(N, C, H, W, dev) if need_x else None
        ds = torch.zeros_like(sigmas) if need_s else None
        ws = workspace(query("favae_blur_bwd_workspace", ctx.ksize, N, H, W, C), dev)
        sp = sigmas.data_ptr() + 4 * ctx.index
        dsp = (ds.data_ptr() + 4 * ctx.index) if need_s else None
        call("favae_blur_bwd", ptr(x), ptr(dy), sp, ctx.ksize, N, H, W, C, ptr(dx), dsp, ptr(ws), ws.numel())
        return dx, ds, None, None


def gaussian_blur(x, sigmas, index, ksize):
    return BlurFn.apply(x, sigmas, index, ksize)


_BLUR_TAP = os.environ.get("FAVAE_BLUR_TAP", "1") != "0"       # A/B switch of BlurTapFn (0: plain blur node + autograd's accumulation)


class BlurTapFn(torch.autograd.Function):
    """(x, blur(x)) for a tap of the codec trunk (models/codec.py:209-215: the tensor is blurred for the DSL and ALSO flows on through
    the trunk).  The caller continues the trunk on the returned alias of x, so both gradients of x arrive in THIS node's backward and the
    trunk gradient is added inside the blur-backward kernel's store (favae_blur_bwd_add) -- autograd otherwise sums the two
    full-size gradient tensors with an ATen add kernel (1.3 ms per step on the 256x256 taps)."""

    @staticmethod
    def forward(ctx, x, sigmas, index, ksize):
        x = to_cl(x)
        _require_gpu(sigmas)
        N, C, H, W = x.shape
        y = new_cl(N, C, H, W, x.device)
        call("favae_blur_fwd", ptr(x), sigmas.data_ptr() + 4 * index, ksize, N, H, W, C, ptr(y))
        ctx.save_for_backward(x, sigmas)
        ctx.index, ctx.ksize = index, ksize
        return x, y

    @staticmethod
    def backward(ctx, gx, gy):
        x, sigmas = ctx.saved_tensors
        N, C, H, W = x.shape
        dev = x.device
        need_x, need_s = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if gy is None:                                   # the tap was not used by any loss
            return gx, None, None, None
        gy = to_cl(gy)
        gx = to_cl(gx) if gx is not None else None
        dx = new_cl(N, C, H, W, dev) if need_x else None
        ds = torch.zeros_like(sigmas) if need_s else None
        ws = workspace(query("favae_blur_bwd_workspace", ctx.ksize, N, H, W, C), dev)
        sp = sigmas.data_ptr() + 4 * ctx.index
        dsp = (ds.data_ptr() + 4 * ctx.index) if need_s else None
        if need_x and gx is not None:
            call("favae_blur_bwd_add", ptr(x), ptr(gy), sp, ctx.ksize, N, H, W, C, ptr(gx), ptr(dx), dsp, ptr(ws), ws.numel())
        else:
            call("favae_blur_bwd", ptr(x), ptr(gy), sp, ctx.ksize, N, H, W, C, ptr(dx), dsp, ptr(ws), ws.numel())
        return dx, ds, None, None


def blur_tap(x, sigmas, index, ksize):
    """-> (alias of x to continue the trunk on, blurred x)"""
    return BlurTapFn.apply(x, sigmas, index, ksize)


# ---------------------------------------------------------------------------------------------------------------
# focal frequency loss (pip focal-frequency-loss 0.3.0 semantics, alpha = 1)
# ---------------------------------------------------------------------------------------------------------------
class FFLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, loss_weight):
        pred, target = to_cl(pred), to_cl(target)
        if pred.shape != target.shape:
            raise RuntimeError("FFL: shape mismatch")
        N, C, H, W = pred.shape
        dev = pred.device
        spec = torch.empty((query("favae_ffl_spec_floats", N, H, W, C),), dtype=torch.float32, device=dev)   # half (Hermitian) spectrum
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        ws = workspace(query("favae_ffl_workspace", N, H, W, C), dev)
        call("favae_ffl_fwd", ptr(pred), ptr(target), N, H, W, C, float(loss_weight), ptr(loss), ptr(spec), ptr(ws), ws.numel())
        ctx.save_for_backward(spec)
        ctx.dims = (N, C, H, W)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (spec,) = ctx.saved_tensors
        N, C, H, W = ctx.dims
        dev = spec.device
        g = g.contiguous().float()
        need_p, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gp = new_cl(N, C, H, W, dev)
        gt = new_cl(N, C, H, W, dev) if need_t else None
        ws = workspace(query("favae_ffl_workspace", N, H, W, C), dev)
        call("favae_ffl_bwd", ptr(spec), ptr(g), N, H, W, C, ptr(gp), ptr(gt), ptr(ws), ws.numel())
        return (gp if need_p else None), gt, None


def focal_frequency_loss(pred, target, loss_weight=1.0):
    return FFLFn.apply(pred, target, loss_weight)


# ---------------------------------------------------------------------------------------------------------------
# vector quantiser pieces (models/l2_quantize.py)
# ---------------------------------------------------------------------------------------------------------------
def vq_lookup(tokens, embed, tie_eps=4e-6):
    """tokens (T,d) contiguous, embed (C,d) contiguous -> idx int64 (T,), zq (T,d), zn (T,d), en (C,d)."""
    _require_gpu(tokens)
    T, d = tokens.shape
    C = embed.shape[0]
    dev = tokens.device
    idx = torch.empty((T,), dtype=torch.int64, device=dev)
    zq = torch.empty((T, d), dtype=torch.float32, device=dev)
    zn = torch.empty_like(zq)
    en = torch.empty((C, d), dtype=torch.float32, device=dev)
    ws = workspace(query("favae_vq_workspace", T, d, C), dev)
    call("favae_vq_lookup", ptr(tokens), ptr(embed), T, d, C, float(tie_eps), ptr(idx), ptr(zq), ptr(zn), ptr(en), ptr(ws),
         ws.numel())
    return idx, zq, zn, en


def vq_segment_sum(zn, idx, C):
    T, d = zn.shape
    bins = torch.empty((C,), dtype=torch.float32, device=zn.device)
    esum = torch.empty((C, d), dtype=torch.float32, device=zn.device)
    ws = workspace(query("favae_vq_segment_workspace", T, C), zn.device)
    call("favae_vq_segment_sum", ptr(zn), ptr(idx), T, d, C, ptr(bins), ptr(esum), ptr(ws), ws.numel())
    return bins, esum


def vq_ema_update(embed, cluster_size, en, bins, esum, decay):
    C, d = en.shape
    call("favae_vq_ema_update", ptr(embed), ptr(cluster_size), ptr(en), ptr(bins), ptr(esum), C, d, float(decay))


class VQStraightThroughFn(torch.autograd.Function):
    """(x, zq) -> quantize = x + (zq - x).detach(), loss = w * mse(zq, x)   (l2_quantize.py:553-561)."""

    @staticmethod
    def forward(ctx, x, zq, weight):
        n = x.numel()
        out = torch.empty_like(x)
        call("favae_vq_ste", ptr(x), ptr(zq), ptr(out), n)
        loss = torch.zeros((1,), dtype=torch.float32, device=x.device)
        if weight > 0:
            ws = workspace(query("favae_reduce_workspace", n), x.device)
            call("favae_sqdiff_sum", ptr(zq), ptr(x), n, float(weight) / n, ptr(loss), ptr(ws), ws.numel())
        ctx.save_for_backward(x, zq)
        ctx.weight = weight
        return out, loss

    @staticmethod
    def backward(ctx, gq, gloss):
        x, zq = ctx.saved_tensors
        n = x.numel()
        gq = gq.contiguous()
        gx = torch.empty_like(x)
        if gloss is None:
            gloss = torch.zeros((1,), dtype=torch.float32, device=x.device)
        call("favae_sqdiff_bwd", ptr(x), ptr(zq), ptr(gloss.contiguous()), 2.0 * ctx.weight / n, n, ptr(gq), ptr(gx))
        return gx, None, None


# ---------------------------------------------------------------------------------------------------------------
# L1 reconstruction loss (favae_scripts/train_favae.py:76)
# ---------------------------------------------------------------------------------------------------------------
class L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = to_cl(a), to_cl(b)
        n = a.numel()
        loss = torch.empty((1,), dtype=torch.float32, device=a.device)
        ws = workspace(query("favae_reduce_workspace", n), a.device)
        call("favae_absdiff_sum", ptr(a), ptr(b), n, 1.0 / n, ptr(loss), ptr(ws), ws.numel())
        ctx.save_for_backward(a, b)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        n = a.numel()
        g = g.contiguous().float()
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = torch.empty_like(a)
            call("favae_absdiff_bwd", ptr(a), ptr(b), ptr(g), 1.0 / n, n, None, ptr(ga))
        if ctx.needs_input_grad[1]:
            gb = torch.empty_like(b)
            call("favae_absdiff_bwd", ptr(a), ptr(b), ptr(g), -1.0 / n, n, None, ptr(gb))
        return ga, gb


def l1_loss(a, b):
    return L1Fn.apply(a, b)


def adam_step(p, g, m, v, step, lr, betas=(0.5, 0.9), eps=1e-8, grad_scale=1.0):
    """In-place Adam over flat fp32 buffers (favae_scripts/train_favae.py:297-305)."""
    _touch_weights(p)                                # per-weight caches (maxima, Winograd records) of this buffer are stale from here on
    call("favae_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
         int(step), float(grad_scale))
