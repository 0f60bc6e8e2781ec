#!/usr/bin/env python3
"""F(2x2, 3x3) Winograd kernel, 16 x 16-pixel x 64-channel workgroups against 16 x 8 x 128 (csrc/conv_wino.h, WIDE): forward with the
fused GroupNorm+SiLU (+ output statistics) and the data gradient with the GroupNorm-backward epilogue -- time per launch, whether the
results are bit-identical (they must be) and the difference of the per-tile partial sums after their reduction.
usage: python tools/wino_wide_bench.py [batch]"""
import os, sys
from ctypes import byref
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H
from favae_hip import ops as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
SHAPES = [(128, 128, 256), (128, 128, 128), (256, 128, 128), (128, 256, 128), (256, 256, 64), (256, 256, 32), (512, 512, 16), (256, 512, 32)]


def timeit(fn, n=10):
    for _ in range(4):               # warm-up: the first loop over a fresh shape measured 10-20 % slow (round 5: clocks / caches), which
        fn()                         # favoured whatever was timed second
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def rel(a, b):
    d = (a.double() - b.double())
    return "%.1e" % float(d.abs().max() / b.abs().max())


K.set_wino4("0")
torch.manual_seed(0)
for cin, cout, hw in SHAPES:
    x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device=dev) * 0.1
    gw, gb = 1 + 0.2 * torch.randn(cin, device=dev), 0.2 * torch.randn(cin, device=dev)
    mean, rstd, scale, shift, xb = K.gn_stats(x, gw, gb, 32, with_bound=True)
    d = H.make_conv_desc(B, hw, hw, cin, hw, hw, cout, 3, 3, 1, 1, 0, H.ACT_SILU, 1)
    d2 = H.make_conv_desc(B, hw, hw, cout, hw, hw, cin, 3, 3, 1, 1, 0, 0, 1)
    flops = 2.0 * B * hw * hw * cout * 9 * cin
    wmax = K.absmax(w)
    dy = torch.randn(B, cout, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    dyb = K.absmax(dy)
    res = {}
    for wide in (0, 1):
        K.set_wino_wide(wide)
        y = K.new_cl(B, cout, hw, hw, dev)
        tiles = H.query("favae_conv_stats_tiles", byref(d), 1, 0)
        st = torch.empty((B * tiles * cout * 2,), dtype=torch.float64, device=dev)
        ya = torch.zeros(1, device=dev)
        t_f = timeit(lambda: K._conv_launch(d, x, w, b, None, scale, shift, y, xb, stats_out=st, y_amax=ya))
        y2 = K.new_cl(B, cout, hw, hw, dev)
        t_p = timeit(lambda: K._conv_launch(d, x, w, b, x if cin == cout else None, scale, shift, y2, xb))
        dx = K.new_cl(B, cin, hw, hw, dev)
        gt = H.query("favae_conv_gnbwd_tiles", byref(d2), 0)
        gws = H.workspace(H.query("favae_gn_bwd_tiles_workspace", B, gt, cin), dev)
        gnb = (x, mean, rstd, gw, gb, 32, H.ACT_SILU, gws)
        t_d = timeit(lambda: K._conv_launch(d2, dy, None, None, None, None, None, dx, dyb, flip_of=(w, cout, 3, 3, cin, wmax), gnbwd=gnb))
        torch.cuda.synchronize()
        sts = st.view(B, tiles, cout, 2).sum(1)
        gs = gws[:B * gt * cin * 16].clone().view(torch.float64).view(B, gt, cin, 2).sum(1)
        res[wide] = (t_f, t_p, t_d, y.clone(), y2.clone(), dx.clone(), sts, gs, float(ya))
    t_44 = float("nan")
    K.set_wino4("1")
    if H.query("favae_conv_wino4_ok", byref(d2), 0):
        dx = K.new_cl(B, cin, hw, hw, dev)
        gt = H.query("favae_conv_gnbwd_tiles", byref(d2), K.PLANES_WINO4)
        gws = H.workspace(H.query("favae_gn_bwd_tiles_workspace", B, gt, cin), dev)
        gnb = (x, mean, rstd, gw, gb, 32, H.ACT_SILU, gws)
        t_44 = timeit(lambda: K._conv_launch(d2, dy, None, None, None, None, None, dx, dyb, flip_of=(w, cout, 3, 3, cin, wmax), gnbwd=gnb))
    K.set_wino4("0")
    f0, p0, d0, y0, z0, dx0, st0, g0, a0 = res[0]
    f1, p1, d1, y1, z1, dx1, st1, g1, a1 = res[1]
    same = "bits %s/%s/%s amax %s" % (torch.equal(y0, y1), torch.equal(z0, z1), torch.equal(dx0, dx1), a0 == a1)
    print(f"{cin:4d}->{cout:4d} @{hw:3d}: fwd+stats 64: {f0:7.1f} us ({flops/f0*1e-6:4.0f} TF) 128: {f1:7.1f} us x{f0/f1:.2f} | fwd+resid "
          f"{p0:7.1f} {p1:7.1f} x{p0/p1:.2f} | dgrad+gnb {d0:7.1f} {d1:7.1f} x{d0/d1:.2f} F44 {t_44:7.1f} | {same} stats {rel(st1, st0)} gnb {rel(g1, g0)}",
          flush=True)
K.set_wino_wide(1)
K.set_wino4("1")
