#!/usr/bin/env python3
"""Error of the HIP convs (forward with fused GN+SiLU, plain forward, data gradient, weight gradient) against an fp64 CPU
reference: max-rel and rms-rel, next to the same errors of torch's fp32 CPU convolution.  Run it per scheme:
FAVAE_CONV_MODE=h3|b6|fp32 python tools/conv_accuracy.py.  `spread` scales a random per-element power-of-two factor into
the inputs (dynamic range of the operands, the case that stresses the fp16 planes)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch, torch.nn.functional as F
from favae_hip import ops as K
torch.manual_seed(0)
dev = torch.device("cuda:0")
print("mode", os.environ.get("FAVAE_CONV_MODE", "h3 (default)"))


def err(a, r):
    d = (a.double() - r)
    return "%.2e/%.2e" % (float(d.abs().max() / r.abs().max()), float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()))


for (N, C, Co, H, spread) in [(2, 128, 128, 64, 0), (2, 512, 512, 16, 0), (1, 256, 256, 32, 0), (1, 128, 128, 256, 0),
                              (2, 128, 128, 64, 12), (2, 128, 128, 64, 24)]:
    x = torch.randn(N, C, H, H)
    gy = torch.randn(N, Co, H, H) * 1e-4
    if spread:
        x = x * torch.exp2(-torch.randint(0, spread + 1, x.shape).float())
        gy = gy * torch.exp2(-torch.randint(0, spread + 1, gy.shape).float())
    w = torch.randn(Co, C, 3, 3) * math.sqrt(1.0 / (C * 9))
    b = torch.randn(Co) * 0.1
    gw, gb = 1 + 0.2 * torch.randn(C), 0.2 * torch.randn(C)
    cfg = K.ConvCfg(3, 3, 1, 1)
    # fp64 references
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    ref_gn = F.conv2d(F.silu(F.group_norm(xd, 32, gw.double(), gb.double())), wd, b.double(), padding=1)
    ref_pl = F.conv2d(xd, wd, b.double(), padding=1)
    ref_dx, ref_dw = torch.autograd.grad(ref_pl, (xd, wd), gy.double())
    # torch fp32 on the CPU
    xf = x.clone().requires_grad_(True)
    wf = w.clone().requires_grad_(True)
    cpu_gn = F.conv2d(F.silu(F.group_norm(xf, 32, gw, gb)), wf, b, padding=1)
    cpu_pl = F.conv2d(xf, wf, b, padding=1)
    cpu_dx, cpu_dw = torch.autograd.grad(cpu_pl, (xf, wf), gy)
    # HIP
    xg = x.to(dev).requires_grad_(True)
    wg = w.to(dev).requires_grad_(True)
    y_gn = K.fused_conv(xg, wg, b.to(dev), gw.to(dev), gb.to(dev), None, cfg)
    y_pl = K.fused_conv(xg, wg, b.to(dev), None, None, None, cfg)
    dx, dw = torch.autograd.grad(y_pl, (xg, wg), gy.to(dev))
    print(f"C={C:4d} H={H:3d} spread 2^-{spread:<2d} max/rms  gn+silu fwd: HIP {err(y_gn.detach().cpu(), ref_gn.detach())} cpu {err(cpu_gn.detach(), ref_gn.detach())}"
          f" | plain fwd: HIP {err(y_pl.detach().cpu(), ref_pl.detach())} cpu {err(cpu_pl.detach(), ref_pl.detach())}"
          f" | dgrad: HIP {err(dx.cpu(), ref_dx)} cpu {err(cpu_dx, ref_dx)} | wgrad: HIP {err(dw.cpu(), ref_dw)} cpu {err(cpu_dw, ref_dw)}", flush=True)
