#!/usr/bin/env python3
"""sha256 of conv results (forward with fused GroupNorm+SiLU, data gradient with the GroupNorm-backward epilogue, weight gradient) on
fixed seeds -- run once per library (FAVAE_HIP_LIB=...) and diff the output: do two builds produce the same bits?"""
import hashlib, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip; favae_hip.load()
from favae_hip import ops as K

d = torch.device("cuda:0")
h = lambda t: hashlib.sha256(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:16]
for (N, cin, cout, H, W) in [(2, 128, 128, 64, 64), (1, 64, 256, 32, 48), (2, 256, 64, 32, 32), (1, 512, 512, 16, 16)]:
    torch.manual_seed(cin + cout)
    x = torch.randn(N, cin, H, W, device=d).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, device=d) * math.sqrt(1.0 / (9 * cin))).requires_grad_(True)
    w2 = (torch.randn(cin, cout, 3, 3, device=d) * math.sqrt(1.0 / (9 * cout))).requires_grad_(True)
    b = torch.randn(cout, device=d) * 0.1
    gw, gb = (1 + 0.2 * torch.randn(cin, device=d)).requires_grad_(True), (0.2 * torch.randn(cin, device=d)).requires_grad_(True)
    gw2, gb2 = (1 + 0.2 * torch.randn(cout, device=d)).requires_grad_(True), (0.2 * torch.randn(cout, device=d)).requires_grad_(True)
    cfg = K.ConvCfg(3, 3, 1, 1, groups=16)
    y = K.fused_conv(x, w, b, gw, gb, None, cfg)
    z = K.fused_conv(y, w2, None, gw2, gb2, x, cfg)
    gs = torch.autograd.grad(z, [x, w, w2, gw, gb, gw2, gb2], torch.randn_like(z))
    K.sync_side_stream(); torch.cuda.synchronize()
    print(N, cin, cout, H, W, h(y), h(z), " ".join(h(g) for g in gs))
# the direct split kernels: 1x1, stride-2 Downsample, Upsample (phase convs), attention core
torch.manual_seed(3)
x = torch.randn(2, 128, 32, 32, device=d).requires_grad_(True)
for name, cfgk, wshape in [("1x1", K.ConvCfg(1, 1, 1, 0), (256, 128, 1, 1)), ("down", K.ConvCfg(3, 3, 2, 0, pad_br=1), (128, 128, 3, 3)),
                           ("up", K.ConvCfg(3, 3, 1, 1, upsample=True), (128, 128, 3, 3)), ("3x3 cout 192", K.ConvCfg(3, 3, 1, 1), (192, 128, 3, 3))]:
    w = (torch.randn(*wshape, device=d) * 0.03).requires_grad_(True)
    b = torch.randn(wshape[0], device=d) * 0.1
    y = K.fused_conv(x, w, b, None, None, None, cfgk)
    gs = torch.autograd.grad(y, [x, w], torch.randn_like(y))
    K.sync_side_stream(); torch.cuda.synchronize()
    print(name, h(y), " ".join(h(g) for g in gs))
qkv = torch.randn(2, 3 * 128, 16, 16, device=d).requires_grad_(True)
o = K.mha_core(qkv, 1, 0.0, False) if hasattr(K, "mha_core") else None
if o is not None:
    g, = torch.autograd.grad(o, [qkv], torch.randn_like(o))
    torch.cuda.synchronize()
    print("attn", h(o), h(g))
