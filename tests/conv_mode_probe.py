"""Child process of tests/test_gpu_ops.py::test_conv_other_modes: forward / data-gradient / weight-gradient parity of one
128-channel 3x3 conv (GroupNorm+SiLU fused, residual) and one Downsample under the FAVAE_CONV_MODE of the environment."""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
from favae_hip import ops as K   # noqa: E402

torch.manual_seed(3)
dev = torch.device("cuda:0")


def relerr(a, b):
    return float((a.detach().cpu().double() - b.detach().double()).abs().max() / b.detach().abs().max())


for (stride, pad, pbr, gn) in [(1, 1, 1, True), (2, 0, 1, False)]:
    N, C, H = 2, 128, 16
    x = torch.randn(N, C, H, H, requires_grad=True)
    w = (torch.randn(C, C, 3, 3) * math.sqrt(1.0 / (C * 9))).requires_grad_(True)
    b = (torch.randn(C) * 0.1).requires_grad_(True)
    gw = (1 + 0.2 * torch.randn(C)).requires_grad_(True)
    gb = (0.2 * torch.randn(C)).requires_grad_(True)
    h = F.silu(F.group_norm(x, 32, gw, gb)) if gn else x
    if pbr != pad:
        y = F.conv2d(F.pad(h, (pad, pbr, pad, pbr)), w, b, stride=stride)
    else:
        y = F.conv2d(h, w, b, stride=stride, padding=pad)
    gy = torch.randn_like(y)
    grads = torch.autograd.grad(y, (x, w, b), gy)
    xd, wd, bd = (t.detach().to(dev).requires_grad_(True) for t in (x, w, b))
    yd = K.fused_conv(xd, wd, bd, gw.detach().to(dev) if gn else None, gb.detach().to(dev) if gn else None, None,
                      K.ConvCfg(3, 3, stride, pad, pad_br=pbr))
    gd = torch.autograd.grad(yd, (xd, wd, bd), gy.to(dev))
    errs = [relerr(yd, y)] + [relerr(a, r) for a, r in zip(gd, grads)]
    print(os.environ.get("FAVAE_CONV_MODE"), "stride", stride, ["%.2e" % e for e in errs])
    assert max(errs) < 5e-5, errs
print("PROBE OK")
