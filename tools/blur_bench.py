#!/usr/bin/env python3
"""Micro-benchmark of the learnable-sigma 9-tap blur (forward, backward) on the FCM tap shapes.  usage: python tools/blur_bench.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
from favae_hip import ops as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
for C, hw in ((128, 256), (512, 16), (256, 16)):
    x = torch.randn(B, C, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sig = torch.full((4,), 3.0, device=dev, requires_grad=True)
    y = K.gaussian_blur(x, sig, 0, 9)
    gy = torch.randn_like(y)

    def t(fn, n=5):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    tf = t(lambda: K.gaussian_blur(x.detach(), sig.detach(), 0, 9))
    tb = t(lambda: torch.autograd.grad(K.gaussian_blur(x, sig, 0, 9), (x, sig), gy)) - tf
    by = x.numel() * 4 / 1e9
    print(f"blur ({B},{C},{hw},{hw}): fwd {tf:.3f} ms ({2 * by / tf:.2f} TB/s)  bwd {tb:.3f} ms ({3 * by / tb:.2f} TB/s)", flush=True)
