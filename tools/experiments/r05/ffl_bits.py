#!/usr/bin/env python3
"""sha256 of the focal-frequency loss value and gradients on fixed seeds -- run once per library (FAVAE_HIP_LIB=...) and diff"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip; favae_hip.load()
from favae_hip import ops as K
d = torch.device("cuda:0")
h = lambda t: hashlib.sha256(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:16]
for (N, C, H, W) in [(2, 128, 256, 256), (3, 512, 16, 16), (2, 3, 256, 256), (1, 64, 32, 32), (2, 32, 64, 128), (1, 16, 128, 8), (1, 8, 512, 4), (1, 4, 12, 24)]:
    torch.manual_seed(C + H)
    a = torch.randn(N, C, H, W, device=d).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(N, C, H, W, device=d).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    l = K.focal_frequency_loss(a, b, 1.0)
    ga, gb = torch.autograd.grad(l, (a, b))
    torch.cuda.synchronize()
    print(N, C, H, W, "%.9g" % float(l), h(l), h(ga), h(gb))
