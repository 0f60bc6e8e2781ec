#!/usr/bin/env python3
"""Error of the HIP conv (forward with fused GN+SiLU, and plain) against an fp64 CPU reference: max-rel and rms-rel."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch, torch.nn.functional as F
from favae_hip import ops as K
torch.manual_seed(0)
dev = torch.device("cuda:0")
for (N, C, Co, H) in [(2, 128, 128, 64), (2, 512, 512, 16), (1, 256, 256, 32)]:
    x = torch.randn(N, C, H, H)
    w = torch.randn(Co, C, 3, 3) * math.sqrt(1.0 / (C * 9))
    b = torch.randn(Co) * 0.1
    gw, gb = 1 + 0.2 * torch.randn(C), 0.2 * torch.randn(C)
    xd, wd = x.double(), w.double()
    ref_gn = F.conv2d(F.silu(F.group_norm(xd, 32, gw.double(), gb.double())), wd, b.double(), padding=1)
    ref_pl = F.conv2d(xd, wd, b.double(), padding=1)
    cpu_gn = F.conv2d(F.silu(F.group_norm(x, 32, gw, gb)), w, b, padding=1)
    y_gn = K.fused_conv(x.to(dev), w.to(dev), b.to(dev), gw.to(dev), gb.to(dev), None, K.ConvCfg(3, 3, 1, 1)).cpu().double()
    y_pl = K.fused_conv(x.to(dev), w.to(dev), b.to(dev), None, None, None, K.ConvCfg(3, 3, 1, 1)).cpu().double()
    def err(a, r):
        d = (a - r)
        return float(d.abs().max() / r.abs().max()), float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt())
    print(f"C={C:4d} H={H:3d}: HIP gn+silu conv max/rms {err(y_gn, ref_gn)[0]:.2e}/{err(y_gn, ref_gn)[1]:.2e} | HIP plain conv {err(y_pl, ref_pl)[0]:.2e}/{err(y_pl, ref_pl)[1]:.2e}"
          f" | torch-CPU fp32 gn conv {err(cpu_gn.double(), ref_gn)[0]:.2e}/{err(cpu_gn.double(), ref_gn)[1]:.2e}")
