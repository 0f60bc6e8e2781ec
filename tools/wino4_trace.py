#!/usr/bin/env python3
"""Where does a workgroup of the F(4x4, 3x3) Winograd kernel (csrc/conv_wino4.h) spend its cycles?  Trace build (tools/wino_trace.sh ->
tools/experiments/libfavae_trace.so): lane 0 of every wave stamps s_memtime at the phase boundaries; mean over 64 sampled workgroups.
usage: python tools/wino4_trace.py [cin cout hw batch]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "experiments", "libfavae_trace.so")
os.environ["FAVAE_HIP_LIB"] = LIB
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H
from favae_hip import ops as K
from ctypes import byref

cin, cout, hw, B = (int(v) for v in (sys.argv[1:5] + ["128", "128", "256", "32"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
raw = ctypes.CDLL(LIB)
raw.favae_debug_wino_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
N = 64 * 8 * 48 * 8


def read():
    buf = np.zeros(N, dtype=np.uint64)
    assert raw.favae_debug_wino_trace(buf.ctypes.data, buf.nbytes) == 0
    return buf.reshape(64, 8, 48, 8).astype(np.int64)


def report(name, t, KC):
    okg = (t[:, :, 0, 0] > 0).all(axis=1)
    print("== %s: %d sampled workgroups" % (name, int(okg.sum())))
    if not okg.any():
        return
    tt = t[okg]                                  # [wg][wave][48][8]
    total = tt[:, :, 47, 7] - tt[:, :, 0, 0]
    print("  wave lifetime %.0f cycles: prologue %.0f, K loop %.0f, epilogue %.0f" % (
        total.mean(), (tt[:, :, 0, 1] - tt[:, :, 0, 0]).mean(), (tt[:, :, 47, 0] - tt[:, :, 0, 1]).mean(), (tt[:, :, 47, 7] - tt[:, :, 47, 0]).mean()))
    e = tt[:, :, 47, :]
    print("  epilogue: pass-0 stores %.0f | barrier %.0f | pass-0 finish + pass-1 loads/stores %.0f | barrier %.0f | pass-1 finish %.0f | stats %.0f" % (
        (e[..., 1] - e[..., 0]).mean(), (e[..., 2] - e[..., 1]).mean(), (e[..., 4] - e[..., 2]).mean(), (e[..., 5] - e[..., 4]).mean(),
        (e[..., 6] - e[..., 5]).mean(), (e[..., 7] - e[..., 6]).mean()))
    rows = list(range(2, KC))                    # chunks 1 .. KC - 2 (steady state: a chunk before and behind)
    c = tt[:, :, rows, :]
    names = ["matrix phase (27 MFMAs, frag reads, ring loads) + halo store", "barrier 1 wait", "transform + split + V stores (+ halo load issue)", "barrier 2 wait"]
    tot = (c[..., 4] - c[..., 0]).mean()
    print("  -- K chunk period %.0f cycles" % tot)
    for i, nm in enumerate(names):
        d = c[..., i + 1] - c[..., i]
        print("     %-66s %7.0f cycles  %5.1f %%   (p10 %6.0f  p90 %6.0f)" % (nm, d.mean(), 100 * d.mean() / tot, np.percentile(d, 10), np.percentile(d, 90)))
    print("  27 MFMAs = 864 pipe cycles per wave, 1728 per SIMD and chunk; %d chunks" % KC)


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


K.set_wino4("2")
x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
b = torch.zeros(cout, device=dev)
gw, gb = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
mean, rstd, scale, shift, xb = K.gn_stats(x, gw, gb, 32, with_bound=True)
y = K.new_cl(B, cout, hw, hw, dev)
d = H.make_conv_desc(B, hw, hw, cin, hw, hw, cout, 3, 3, 1, 1, 0, H.ACT_SILU, 1)
with K.wino4_forward(True):
    fwd = lambda: K._conv_launch(d, x, w, b, None, scale, shift, y, xb)
    ms = timeit(fwd)
    read()
    fwd(); t = read()
print("forward %d->%d @%d batch %d: %.3f ms (with the stamps compiled in)" % (cin, cout, hw, B, ms))
report("forward <2,false,false>", t, cin // 16)
dy = torch.randn(B, cout, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
dyb = K.absmax(dy)
d2 = H.make_conv_desc(B, hw, hw, cout, hw, hw, cin, 3, 3, 1, 1, 0, 0, 1)
dx = K.new_cl(B, cin, hw, hw, dev)
gt = H.query("favae_conv_gnbwd_tiles", byref(d2), K.PLANES_WINO4)
gws = H.workspace(H.query("favae_gn_bwd_tiles_workspace", B, gt, cin), dev)
gnb = (x, mean, rstd, gw, gb, 32, H.ACT_SILU, gws)
wmax = K.absmax(w)
dg = lambda: K._conv_launch(d2, dy, None, None, None, None, None, dx, dyb, flip_of=(w, cout, 3, 3, cin, wmax), gnbwd=gnb)
ms = timeit(dg)
read()
dg(); t = read()
print("data gradient: %.3f ms" % ms)
report("data gradient <0,true,false>", t, cout // 16)
