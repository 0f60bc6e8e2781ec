import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import favae_hip as H
from favae_hip import ops as K
d = torch.device("cuda:0")
N, cin, cout, Hh, W = 1, 64, 64, 16, 32
g = torch.Generator().manual_seed(7)
x = torch.randint(-1, 2, (N, cin, Hh, W), generator=g).float().to(d)
w = (576.0 * torch.randint(-1, 2, (cout, cin, 3, 3), generator=g).float()).to(d)
for mode in ("0", "2"):
    K.set_wino4(mode)
    with K.wino4_forward(True), torch.no_grad():
        y = K.fused_conv(x, w, None, None, None, None, K.ConvCfg(3, 3, 1, 1))
    torch.cuda.synchronize()
    yr = F.conv2d(x.cpu().double(), w.cpu().double(), padding=1)
    e = (y.cpu().double() - yr).abs()
    print("mode", mode, "max err", float(e.max()), "ref max", float(yr.abs().max()), "nonzero frac", float((e > 0).float().mean()))
    if float(e.max()) > 0:
        em = e[0].amax(0)                      # (H, W) max over channels
        print(" err by (y%4, x%4):")
        for i in range(4):
            print("   ", ["%.3g" % float(em[i::4, j::4].max()) for j in range(4)])
        print(" err by tile row/col (max):")
        for ty in range(Hh // 4):
            print("   ", ["%.3g" % float(em[4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4].max()) for tx in range(W // 4)])
        ec = e[0].amax((1, 2))
        print(" err by channel (first 16):", ["%.3g" % float(v) for v in ec[:16]], " max over ch 32..:", float(ec[32:].max()))
        # single-channel / delta probes
# delta probe: one input pixel, one channel pair
for (cy, cx) in ((5, 9), (0, 0), (15, 31)):
    x2 = torch.zeros(N, cin, Hh, W, device=d); x2[0, 3, cy, cx] = 1.0
    w2 = torch.zeros(cout, cin, 3, 3, device=d); w2[5, 3] = 576.0 * torch.arange(1, 10, device=d).float().view(3, 3)
    K.set_wino4("2")
    with K.wino4_forward(True), torch.no_grad():
        y = K.fused_conv(x2, w2, None, None, None, None, K.ConvCfg(3, 3, 1, 1))
    yr = F.conv2d(x2.cpu().double(), w2.cpu().double(), padding=1)
    e = (y.cpu().double() - yr).abs()
    print("delta at", (cy, cx), "max err", float(e.max()))
    if float(e.max()) > 0:
        ys, xs = max(0, cy - 2), max(0, cx - 2)
        print(" got\n", y[0, 5, ys:cy + 3, xs:cx + 3].cpu() / 576, "\n ref\n", yr[0, 5, ys:cy + 3, xs:cx + 3] / 576)
K.set_wino4("1")
