#!/usr/bin/env python3
"""Stand-alone repro attempt: the GroupNorm-backward apply pass (favae_gn_act_bwd: rows kernel WITHOUT the column-sum epilogue;
favae_gn_act_bwd_colsum: with it) on the main stream while nine-tap weight gradients run on a second stream of the same process.
Every result is compared with the one computed on a quiet GPU."""
import os, sys
from ctypes import byref
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H
from favae_hip import ops as K
dev = torch.device("cuda", 0)
torch.manual_seed(2)
N, C, Hh, W, G = 4, 128, 64, 64, 32
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
aggr = sys.argv[2] if len(sys.argv) > 2 else "wgrad"
x = (torch.randn(N, C, Hh, W, device=dev) * 1.5).contiguous(memory_format=torch.channels_last)
da = (torch.randn(N, C, Hh, W, device=dev) * 1e-5).contiguous(memory_format=torch.channels_last)
gw, gb = 1 + 0.2 * torch.randn(C, device=dev), 0.2 * torch.randn(C, device=dev)
mean, rstd, scale, shift, xb = K.gn_stats(x, gw, gb, G, with_bound=True)
nws = H.query("favae_gn_workspace", N, Hh * W, C)
nb = H.query("favae_gn_bwd_colsum_blocks", N, Hh * W, C)


def apply(colsum):
    dx = K.new_cl(N, C, Hh, W, dev)
    dx.fill_(777.0)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    dg, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    if colsum:
        cp = torch.empty(nb * C, device=dev)
        ca = torch.empty(1, device=dev)
        H.call("favae_gn_act_bwd_colsum", H.ptr(da), H.ptr(x), H.ptr(gw), H.ptr(gb), H.ptr(mean), H.ptr(rstd), N, Hh * W, C, G, 1, None,
               H.ptr(dx), H.ptr(dg), H.ptr(dbt), 0, 0, H.ptr(ws), ws.numel(), H.ptr(cp), H.ptr(ca))
    else:
        H.call("favae_gn_act_bwd", H.ptr(da), H.ptr(x), H.ptr(gw), H.ptr(gb), H.ptr(mean), H.ptr(rstd), N, Hh * W, C, G, 1, None,
               H.ptr(dx), H.ptr(dg), H.ptr(dbt), 0, H.ptr(ws), ws.numel())
    return dx


ref = {c: apply(c) for c in (False, True)}
torch.cuda.synchronize()
assert torch.equal(ref[False], ref[True]) or float((ref[False] - ref[True]).abs().max()) < 1e-9

# aggressor on a second stream
side = torch.cuda.Stream()
if aggr == "wgrad":
    NB = 32
    xa = torch.randn(NB, C, Hh, W, device=dev).contiguous(memory_format=torch.channels_last)
    ya = (torch.randn(NB, C, Hh, W, device=dev) * 1e-3).contiguous(memory_format=torch.channels_last)
    m2, r2, sc2, sh2, xb2 = K.gn_stats(xa, gw, gb, G, with_bound=True)
    yb2 = K.absmax(ya)
    d = H.make_conv_desc(NB, Hh, W, C, Hh, W, C, 3, 3, 1, 1, 0, H.ACT_SILU, 1)
    wws = H.workspace(H.query("favae_conv_wgrad_workspace", byref(d)), dev)
    dw = torch.empty(C, 3, 3, C, device=dev)

    def kick():
        with torch.cuda.stream(side):
            H.call("favae_conv_wgrad", byref(d), H.ptr(xa), H.ptr(ya), H.ptr(sc2), H.ptr(sh2), H.ptr(xb2), H.ptr(yb2), H.ptr(dw), 0, H.ptr(wws), wws.numel())
elif aggr == "conv":                      # the Winograd forward conv (GN + SiLU on load) as the second-stream kernel
    NB = 32
    xa = torch.randn(NB, C, Hh, W, device=dev).contiguous(memory_format=torch.channels_last)
    wa = (torch.randn(C, C, 3, 3, device=dev) * 0.03).contiguous(memory_format=torch.channels_last)
    ba = torch.zeros(C, device=dev)
    m2, r2, sc2, sh2, xb2 = K.gn_stats(xa, gw, gb, G, with_bound=True)
    ya = K.new_cl(NB, C, Hh, W, dev)
    d = H.make_conv_desc(NB, Hh, W, C, Hh, W, C, 3, 3, 1, 1, 0, H.ACT_SILU, 1)
    K._conv_launch(d, xa, wa, ba, None, sc2, sh2, ya, xb2)

    def kick():
        with torch.cuda.stream(side):
            K._conv_launch(d, xa, wa, ba, None, sc2, sh2, ya, xb2)
elif aggr == "mm":
    ma = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
    mc = torch.empty(4096, 4096, device=dev, dtype=torch.float16)

    def kick():
        with torch.cuda.stream(side):
            torch.mm(ma, ma, out=mc)
else:
    def kick():
        pass
torch.cuda.synchronize()
bad = {False: 0, True: 0}
bad_t = 0
ref_t = (x * 1.5 + da).clone()
for r in range(reps):
    for _ in range(2):
        kick()
    yt = x * 1.5 + da                          # an ATen elementwise kernel as a third victim
    kick()
    torch.cuda.synchronize()
    bad_t += int(not torch.equal(yt, ref_t))
    for _ in range(3):
        kick()
    for c in (False, True):
        dx = apply(c)
        kick()
        torch.cuda.synchronize()
        if not torch.equal(dx, ref[c]):
            bad[c] += 1
            if bad[c] <= 2 and not c:
                # which value did the kernel use for d (= da) where it went wrong?  dx = rs (d act'(y) ga - k1 - xh k2), k terms negligible here
                w0 = (dx != ref[c])
                xm, dm = x.permute(0, 2, 3, 1), dx.permute(0, 2, 3, 1)
                idx = w0.permute(0, 2, 3, 1).nonzero()[:6]
                g_of = torch.arange(C, device=dev) // (C // G)
                for n_, h_, w_, c_ in idx.tolist():
                    rs_, mu_ = float(rstd[n_ * G + g_of[c_]]) if rstd.dim() == 1 else float(rstd.reshape(-1)[n_ * G + int(g_of[c_])]), float(mean.reshape(-1)[n_ * G + int(g_of[c_])])
                    xv = float(xm[n_, h_, w_, c_])
                    xh = (xv - mu_) * rs_
                    yv = xh * float(gw[c_]) + float(gb[c_])
                    sg = 1.0 / (1.0 + torch.exp(torch.tensor(-yv)).item())
                    ag = sg * (1.0 + yv * (1.0 - sg))
                    d_eff = float(dm[n_, h_, w_, c_]) / (rs_ * float(gw[c_]) * ag)
                    print("   wrong at (n %d, y %d, x %d, c %d): dx %.4e -> the kernel used d = %.5f there; x there = %.5f; x one row up/down = %.5f / %.5f; da there = %.3e"
                          % (n_, h_, w_, c_, float(dm[n_, h_, w_, c_]), d_eff, xv, float(xm[n_, max(h_ - 1, 0), w_, c_]), float(xm[n_, min(h_ + 1, Hh - 1), w_, c_]),
                             float(da.permute(0, 2, 3, 1)[n_, h_, w_, c_])), flush=True)
            if bad[c] <= 3:
                w = dx != ref[c]
                print("rep %d colsum=%s: %d wrong elements, of them still the fill value %d, max |wrong| %.3e (|right| max %.3e)"
                      % (r, c, int(w.sum()), int((dx[w] == 777.0).sum()), float(dx[w].abs().max()), float(ref[c].abs().max())), flush=True)
print("aggressor %s: ATen elementwise kernel wrong in %d of %d" % (aggr, bad_t, reps))
print("aggressor %s: apply WITHOUT the epilogue wrong in %d of %d, WITH it in %d of %d" % (aggr, bad[False], reps, bad[True], reps))
