#!/usr/bin/env python3
"""Micro-benchmark of the focal-frequency loss (forward + backward) on the feature-pair shapes of the f=16 model.
usage: python tools/ffl_bench.py [batch]   -> ms and effective GB/s (algorithmic bytes of SURVEY 8(d): 2 reads fwd + 2 writes bwd
of the real tensors; the spectrum passes are internal traffic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
from favae_hip import ops as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
for C, hw in ((128, 256), (512, 16), (256, 16), (3, 256)):
    a = torch.randn(B, C, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(B, C, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)

    def run():
        l = K.focal_frequency_loss(a, b, 1.0)
        ga, gb = torch.autograd.grad(l, (a, b))
        return l

    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        l = run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    byt = 4.0 * a.numel() * 4
    print(f"FFL fwd+bwd ({B},{C},{hw},{hw}): {ms:.3f} ms  {byt / ms * 1e-6:.0f} GB/s algorithmic  loss {float(l):.6f}", flush=True)
