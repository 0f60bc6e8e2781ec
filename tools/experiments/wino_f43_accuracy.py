#!/usr/bin/env python3
"""Accuracy gate for an F(4x4,3x3) Winograd variant of the split-precision conv (third avenue), on the CPU:
emulates the kernel's arithmetic -- fp32 input / weight transforms, operands split into two scaled fp16 planes (hi = rne(v), lo =
rne(v - hi)), three products hi.hi + hi.lo + lo.hi accumulated in fp32 over the input channels, fp32 output transform -- for
F(2x2,3x3) (the product kernel's algorithm) and F(4x4,3x3), against an fp64 direct convolution, next to torch's fp32 direct conv.
Prints max-rel / rms-rel error (relative to the output's max / rms), as tools/conv_accuracy.py does for the product kernels."""
import math, sys, torch, torch.nn.functional as F
torch.manual_seed(0)

BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                   [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def split(v, headroom):
    """Two fp16 planes of v at one power-of-two scale per tensor (absmax -> 2^15 / headroom), as the kernels do."""
    s = 2.0 ** math.floor(math.log2(32768.0 / headroom / float(v.abs().max())))
    t = v * s
    hi = t.half()
    lo = (t - hi.float()).half()
    return hi.float(), lo.float(), s


def wino(x, w, BT, G, AT, m, exact_products=False):
    """x (N,C,H,H) fp32, w (Co,C,3,3) fp32, pad 1.  m = output tile edge (2 or 4)."""
    N, C, H, _ = x.shape
    Co = w.shape[0]
    a = m + 2
    BT, G, AT = BT.float(), G.float(), AT.float()
    xp = F.pad(x, (1, 1, 1, 1))
    t = xp.unfold(2, a, m).unfold(3, a, m)                                   # N C th tw a a
    th = t.shape[2]
    V = torch.einsum("ij,nchwjk,lk->nchwil", BT, t, BT)                      # fp32 transforms
    U = torch.einsum("ij,ocjk,lk->ocil", G, w, G)
    if exact_products:
        M = torch.einsum("nchwil,ocil->nohwil", V.double(), U.double()).float()
    else:
        vh, vl, sv = split(V, 1)
        uh, ul, su = split(U, 1)
        M = torch.zeros(N, Co, th, th, a, a)
        for (p, q) in ((vh, uh), (vh, ul), (vl, uh)):
            M += torch.einsum("nchwil,ocil->nohwil", p, q)                   # fp32 accumulation (values are exact fp16 products)
        M = M / (sv * su)
    Y = torch.einsum("ij,nohwjk,lk->nohwil", AT, M, AT)                      # N Co th tw m m
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, Co, H, H)


def err(a, r):
    d = a.double() - r
    return "%.2e/%.2e" % (float(d.abs().max() / r.abs().max()), float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()))


for (N, C, Co, H, spread) in [(1, 128, 128, 64, 0), (1, 128, 128, 64, 12), (1, 512, 512, 16, 0)]:
    x = torch.randn(N, C, H, H)
    if spread:
        x = x * torch.exp2(-torch.randint(0, spread + 1, x.shape).float())
    w = torch.randn(Co, C, 3, 3) * math.sqrt(1.0 / (C * 9))
    ref = F.conv2d(x.double(), w.double(), padding=1)
    print(f"C={C} H={H} spread 2^-{spread}: max/rms  torch fp32 direct {err(F.conv2d(x, w, padding=1), ref)}"
          f" | F(2,3) h3 {err(wino(x, w, BT2, G2, AT2, 2), ref)}  fp32 transforms only {err(wino(x, w, BT2, G2, AT2, 2, True), ref)}"
          f" | F(4,3) h3 {err(wino(x, w, BT4, G4, AT4, 4), ref)}  fp32 transforms only {err(wino(x, w, BT4, G4, AT4, 4, True), ref)}", flush=True)
