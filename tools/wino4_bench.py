#!/usr/bin/env python3
"""F(4x4, 3x3) against F(2x2, 3x3) Winograd kernels (csrc/conv_wino4.h / conv_wino.h) on the layer shapes that tile into both:
forward with the fused GroupNorm+SiLU (+ output statistics) and the data gradient with the GroupNorm-backward epilogue, time per
launch and the difference of the results (max / rms relative to the F(2x2) result's max / rms).
usage: python tools/wino4_bench.py [batch]"""
import os, sys
from ctypes import byref
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H
from favae_hip import ops as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
SHAPES = [(128, 128, 256), (128, 128, 128), (256, 128, 128), (256, 256, 64), (256, 256, 32)]


def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def rel(a, b):
    d = (a.double() - b.double())
    return "%.1e/%.1e" % (float(d.abs().max() / b.abs().max()), float(d.pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt()))


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
    res = {}
    for mode in ("0", "2"):
        K.set_wino4(mode)
        with K.wino4_forward(True):
            y = K.new_cl(B, cout, hw, hw, dev)
            tiles = H.query("favae_conv_stats_tiles", byref(d), 1, K.PLANES_WINO4 if mode == "2" else 0)
            st = torch.empty((B * tiles * cout * 2,), dtype=torch.float64, device=dev)
            ya = torch.zeros(1, device=dev)
            t_f = timeit(lambda: K._conv_launch(d, x, w, b, None, scale, shift, y, xb, stats_out=st, y_amax=ya))
            dy = torch.randn(B, cout, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
            dyb = K.absmax(dy)
            dx = K.new_cl(B, cin, hw, hw, dev)
            gt = H.query("favae_conv_gnbwd_tiles", byref(d2), K.PLANES_WINO4 if mode == "2" else 0)
            gws = H.workspace(H.query("favae_gn_bwd_tiles_workspace", B, gt, cin), dev)
            torch.manual_seed(1)
            dy.copy_(torch.randn(B, cout, hw, hw, device=dev))
            dyb = K.absmax(dy)
            gnb = (x, mean, rstd, gw, gb, 32, H.ACT_SILU, gws)
            t_d = timeit(lambda: K._conv_launch(d2, dy, None, None, None, None, None, dx, dyb, flip_of=(w, cout, 3, 3, cin, wmax), gnbwd=gnb))
            torch.cuda.synchronize()
            res[mode] = (t_f, t_d, y.clone(), dx.clone(), st.clone(), gws[:B * gt * cin * 16].clone().view(torch.float64))
    f0, d0, y0, dx0, st0, g0 = res["0"]
    f4, d4, y4, dx4, st4, g4 = res["2"]
    print(f"{cin:4d}->{cout:4d} @{hw:3d}: fwd F22 {f0:7.1f} us ({flops/f0*1e-6:5.0f} TF)  F44 {f4:7.1f} us ({flops/f4*1e-6:5.0f} TF) x{f0/f4:.2f} | "
          f"dgrad F22 {d0:7.1f} us  F44 {d4:7.1f} us x{d0/d4:.2f} | diff max/rms: y {rel(y4, y0)} dx {rel(dx4, dx0)} stats {rel(st4, st0)} gnb {rel(g4, g0)}",
          flush=True)
K.set_wino4("1")
