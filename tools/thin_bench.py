#!/usr/bin/env python3
"""Micro-benchmark of the thin-channel 3x3 convs (RGB ends of the codec): conv_in 3->128 and conv_out 128->3 (GroupNorm+SiLU fused)
at 256x256, forward / data gradient / weight gradient.  usage: python tools/thin_bench.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
from favae_hip import ops as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
K._SIDE["on"] = False


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


cfg = K.ConvCfg(3, 3, 1, 1)
x3 = torch.randn(B, 3, 256, 256, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
w_in = (torch.randn(128, 3, 3, 3, device=dev) * 0.1).contiguous(memory_format=torch.channels_last).requires_grad_(True)
b_in = torch.zeros(128, device=dev, requires_grad=True)
y = K.fused_conv(x3, w_in, b_in, cfg=cfg)
gy = torch.randn_like(y)
print("conv_in  3->128 fwd        %.3f ms" % timeit(lambda: K.fused_conv(x3, w_in, b_in, cfg=cfg)))
print("conv_in  3->128 fwd+bwd(w) %.3f ms" % timeit(lambda: torch.autograd.grad(K.fused_conv(x3.detach(), w_in, b_in, cfg=cfg), (w_in, b_in), gy)))
x128 = torch.randn(B, 128, 256, 256, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
w_out = (torch.randn(3, 128, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
b_out = torch.zeros(3, device=dev, requires_grad=True)
gw, gb = torch.ones(128, device=dev, requires_grad=True), torch.zeros(128, device=dev, requires_grad=True)
cg = K.ConvCfg(3, 3, 1, 1, groups=32)
yo = K.fused_conv(x128, w_out, b_out, gw, gb, None, cg)
go = torch.randn_like(yo)
print("conv_out 128->3 (GN+SiLU) fwd      %.3f ms" % timeit(lambda: K.fused_conv(x128, w_out, b_out, gw, gb, None, cg)))
print("conv_out 128->3 (GN+SiLU) fwd+bwd  %.3f ms" % timeit(lambda: torch.autograd.grad(K.fused_conv(x128, w_out, b_out, gw, gb, None, cg), (x128, w_out, b_out, gw, gb), go)))
