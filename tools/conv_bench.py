#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the layer shapes of the f=16 model (batch 32): TFLOP/s per shape for
forward (GN+SiLU fused), data gradient and weight gradient.  usage: python tools/conv_bench.py [batch]"""
import os, sys, math
from ctypes import byref
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H
from favae_hip import ops as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
SHAPES = [(128, 128, 256, 3), (128, 128, 128, 3), (256, 256, 64, 3), (256, 256, 32, 3), (512, 512, 16, 3), (256, 128, 128, 3),
          (512, 1536, 16, 1), (512, 512, 16, 1)]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


for cin, cout, hw, k in SHAPES:
    x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    if os.environ.get("CONV_BENCH_ZEROS") == "1":          # power experiment: all-zero operands (no data toggling in the matrix pipe)
        x.zero_(); w.zero_()
    b = torch.zeros(cout, device=dev)
    gw, gb = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    mean, rstd, scale, shift, xb = K.gn_stats(x, gw, gb, 32, with_bound=True)
    y = K.new_cl(B, cout, hw, hw, dev)
    d = H.make_conv_desc(B, hw, hw, cin, hw, hw, cout, k, k, 1, k // 2, 0, H.ACT_SILU, 1)
    flops = 2.0 * B * hw * hw * cout * k * k * cin
    t_f = timeit(lambda: K._conv_launch(d, x, w, b, None, scale, shift, y, xb))
    yb = K.absmax(y)
    wt = torch.empty(cin, k, k, cout, device=dev)
    H.call("favae_weight_flip", H.ptr(w), H.ptr(wt), cout, k, k, cin)
    d2 = H.make_conv_desc(B, hw, hw, cout, hw, hw, cin, k, k, 1, k // 2, 0, 0, 1)
    dx = K.new_cl(B, cin, hw, hw, dev)
    t_d = timeit(lambda: K._conv_launch(d2, y, wt, None, None, None, None, dx, yb))
    dw = torch.empty(cout, k, k, cin, device=dev)
    ws = H.workspace(H.query("favae_conv_wgrad_workspace", byref(d)), dev)
    t_w = timeit(lambda: H.call("favae_conv_wgrad", byref(d), H.ptr(x), H.ptr(y), H.ptr(scale), H.ptr(shift), H.ptr(xb), H.ptr(yb), H.ptr(dw), 0, H.ptr(ws), ws.numel()))
    # the same weight gradient without the fused GroupNorm+SiLU on the x operand (plain conv): isolates the cost of the transform
    d0 = H.make_conv_desc(B, hw, hw, cin, hw, hw, cout, k, k, 1, k // 2, 0, H.ACT_NONE, 1)
    xb0 = K.absmax(x)
    t_w0 = timeit(lambda: H.call("favae_conv_wgrad", byref(d0), H.ptr(x), H.ptr(y), None, None, H.ptr(xb0), H.ptr(yb), H.ptr(dw), 0, H.ptr(ws), ws.numel()))
    print(f"{cin:4d}->{cout:4d} @{hw:3d} k{k}: fwd {flops/t_f*1e-12:6.1f}  dgrad {flops/t_d*1e-12:6.1f}  wgrad {flops/t_w*1e-12:6.1f} (plain x: {flops/t_w0*1e-12:6.1f}) TFLOP/s   ({t_f*1e3:.2f} / {t_d*1e3:.2f} / {t_w*1e3:.2f} ms)", flush=True)
