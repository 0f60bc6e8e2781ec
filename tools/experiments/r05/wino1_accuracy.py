#!/usr/bin/env python3
"""one-plane (h1 / b1) Winograd kernel against the direct one-plane kernel: error of forward, data gradient and weight gradient relative to
the fp32-grade h3 result on one GroupNorm+SiLU conv"""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "fa-vae_amd"))
import torch
import favae_hip; favae_hip.load()
from favae_hip import ops as K
d = torch.device("cuda:0")
torch.manual_seed(0)
N, cin, cout, H, W = 2, 128, 128, 64, 64
x = torch.randn(N, cin, H, W, device=d)
w = torch.randn(cout, cin, 3, 3, device=d) * math.sqrt(1.0 / (9 * cin))
b = torch.randn(cout, device=d) * 0.1
gw, gb = 1 + 0.2 * torch.randn(cin, device=d), 0.2 * torch.randn(cin, device=d)
cfg = K.ConvCfg(3, 3, 1, 1, groups=16)
def run():
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = K.fused_conv(xg, wg, b, gw, gb, None, cfg)
    gx, gwt = torch.autograd.grad(y, [xg, wg], torch.ones_like(y) * 0.01 + 0.001 * y.detach())
    K.sync_side_stream(); torch.cuda.synchronize()
    return y.detach().double().cpu(), gx.double().cpu(), gwt.double().cpu()
ref = run()                                   # h3
for mode in ("h1", "b1"):
    K.set_conv_mode(mode)
    outs = {}
    for w1 in ("1", "0"):
        prev = favae_hip.query("favae_set_wino", 1 if w1 == "1" else 0)
        outs[w1] = run()
        favae_hip.query("favae_set_wino", prev)
    for name, i in (("y", 0), ("dx", 1), ("dw", 2)):
        s = float(ref[i].abs().max())
        print(mode, name, "wino %.2e  direct %.2e (max err / max of the h3 result)" % (float((outs["1"][i] - ref[i]).abs().max()) / s, float((outs["0"][i] - ref[i]).abs().max()) / s))
K.set_conv_mode("h3")
