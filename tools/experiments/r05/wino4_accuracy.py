"""F(4x4,3x3) and F(2x2,3x3) kernels against an fp64 convolution (plain conv, no fused GroupNorm): max / rms error relative to the output's max / rms."""
import os, sys, math, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import favae_hip as H
from favae_hip import ops as K
d = torch.device("cuda:0")
torch.manual_seed(0)
def err(a, r):
    e = a.cpu().double() - r
    return "%.2e/%.2e" % (float(e.abs().max() / r.abs().max()), float(e.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()))
for (N, C, Co, Hh, W, spread) in [(1, 128, 128, 64, 64, 0), (1, 128, 128, 64, 64, 12), (2, 256, 256, 32, 32, 0), (1, 64, 64, 32, 64, 0)]:
    x = torch.randn(N, C, Hh, W)
    if spread:
        x = x * torch.exp2(-torch.randint(0, spread + 1, x.shape).float())
    w = torch.randn(Co, C, 3, 3) * math.sqrt(1.0 / (C * 9))
    gy = torch.randn(N, Co, Hh, W)
    xr = x.double().requires_grad_(True)
    yr = F.conv2d(xr, w.double(), padding=1)
    dxr, = torch.autograd.grad(yr, (xr,), gy.double())
    out = []
    for mode in ("0", "2"):
        K.set_wino4(mode)
        with K.wino4_forward(True):
            xg = x.to(d).requires_grad_(True)
            y = K.fused_conv(xg, w.to(d), None, None, None, None, K.ConvCfg(3, 3, 1, 1))
            dx, = torch.autograd.grad(y, (xg,), gy.to(d))
            K.sync_side_stream(); torch.cuda.synchronize()
        out.append("y %s dx %s" % (err(y.detach(), yr.detach()), err(dx, dxr)))
    print(f"C={C}->{Co} {Hh}x{W} spread 2^-{spread}: max/rms vs fp64 | F(2x2) {out[0]} | F(4x4) {out[1]} | torch fp32 y {err(F.conv2d(x, w, padding=1), yr.detach())}", flush=True)
K.set_wino4("1")
print("-- which plane loses the bits: exactly representable operand on one side")
for case in ("x in {-1,0,1} (V exact in the hi plane), w real", "w in 576{-1,0,1} (U exact in the hi plane), x real"):
    N, C, Co, Hh, W = 1, 128, 128, 32, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-1, 2, (N, C, Hh, W), generator=g).float() if case.startswith("x in") else torch.randn(N, C, Hh, W)
    w = torch.randn(Co, C, 3, 3) * 0.03 if case.startswith("x in") else 576.0 * torch.randint(-1, 2, (Co, C, 3, 3), generator=g).float()
    yr = F.conv2d(x.double(), w.double(), padding=1)
    out = []
    for mode in ("0", "2"):
        K.set_wino4(mode)
        with K.wino4_forward(True), torch.no_grad():
            y = K.fused_conv(x.to(d), w.to(d), None, None, None, None, K.ConvCfg(3, 3, 1, 1))
            torch.cuda.synchronize()
        out.append(err(y, yr))
    print(f"{case}: F(2x2) {out[0]} | F(4x4) {out[1]}", flush=True)
K.set_wino4("1")
