#!/usr/bin/env python3
"""op-level: GN+SiLU+conv (+residual) backward at (4,128,64,64) with the by-products of the apply pass on / off"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H_
from favae_hip import ops as K
dev = torch.device("cuda", 0)
torch.manual_seed(1)
N, C, Hh, W = 4, 128, 64, 64
x = torch.randn(N, C, Hh, W)
w = torch.randn(C, C, 3, 3) * 0.03
b = torch.randn(C) * 0.1
gw, gb = 1 + 0.2 * torch.randn(C), 0.2 * torch.randn(C)
gy = torch.randn(N, C, Hh, W) * 1e-3
cfg = K.ConvCfg(3, 3, 1, 1, act=1, groups=32)
for resid in (False, True):
    outs = []
    for fuse in (True, False):
        K._DYCS_FUSE = fuse
        calls = []
        H_.set_call_hook(lambda name, args, launch: (calls.append(name), launch())[1])
        xs = x.to(dev).requires_grad_(True)
        ps = [t.to(dev).requires_grad_(True) for t in (w, b, gw, gb)]
        r = xs if resid else None
        y = K.fused_conv(xs, ps[0], ps[1], ps[2], ps[3], r, cfg)
        gr = torch.autograd.grad(y, [xs] + ps, gy.to(dev))
        K.sync_side_stream(); torch.cuda.synchronize()
        H_.set_call_hook(None)
        outs.append([g.cpu() for g in gr])
        print("resid", resid, "fuse", fuse, [c for c in calls if "gn_act" in c or "colsum" in c or "absmax" in c or "conv_fwd" in c or "gnbwd" in c])
    K._DYCS_FUSE = True
    for nm, a, bb in zip(["dx", "dw", "db", "dgamma", "dbeta"], *outs):
        print("   %-7s max diff %.3e of %.3e" % (nm, float((a - bb).abs().max()), float(a.abs().max())))
