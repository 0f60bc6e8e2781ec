#!/usr/bin/env python3
"""(FFL_INPROC=1: the aggressor is in THIS process: fp16 GEMMs (rocBLAS MFMA kernels) queued on a second stream.)
favae_ffl_bwd twice on the same saved spectrum into separate outputs / workspaces, repeated while another process trains:
which of its two FFT passes (H-axis complex -> scratch, W-axis half spectrum -> real) is not reproducible?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H
from favae_hip import ops as K
tag, reps = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)
torch.manual_seed(3)
N, C, Hh, W = 4, 128, 64, 64
p = torch.randn((N, C, Hh, W), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
t = torch.randn((N, C, Hh, W), device=dev).contiguous(memory_format=torch.channels_last)
l = K.focal_frequency_loss(p, t, 1.0)
spec = l.grad_fn.saved_tensors[0]
g = torch.ones((), device=dev)
nws = H.query("favae_ffl_workspace", N, Hh, W, C)


def bwd():
    gp = K.new_cl(N, C, Hh, W, dev)
    ws = torch.zeros(nws // 4 + 64, dtype=torch.float32, device=dev)
    H.call("favae_ffl_bwd", H.ptr(spec), H.ptr(g), N, Hh, W, C, H.ptr(gp), None, H.ptr(ws), ws.numel() * 4)
    return gp, ws


gp0, ws0 = bwd()
torch.cuda.synchronize()
for _ in range(20):                       # the reference must come from a quiet GPU: it has to reproduce itself
    gp1, ws1 = bwd()
    torch.cuda.synchronize()
    assert torch.equal(gp1, gp0) and torch.equal(ws1, ws0), "the reference run is not reproducible: start this process before the other"
import time
time.sleep(float(os.environ.get("FFL_WAIT", "0")))
badA = badB = 0
inproc = os.environ.get("FFL_INPROC") == "1"
if os.environ.get("FFL_INPROC") == "wgrad":          # the aggressor is this library's nine-tap weight gradient on a second stream
    from ctypes import byref
    side = torch.cuda.Stream()
    NB, Cw, Hw = 32, 128, 64
    gww, gbw = torch.ones(Cw, device=dev), torch.zeros(Cw, device=dev)
    xa = torch.randn(NB, Cw, Hw, Hw, device=dev).contiguous(memory_format=torch.channels_last)
    ya = (torch.randn(NB, Cw, Hw, Hw, device=dev) * 1e-3).contiguous(memory_format=torch.channels_last)
    m2, r2, sc2, sh2, xb2 = K.gn_stats(xa, gww, gbw, 32, with_bound=True)
    yb2 = K.absmax(ya)
    cd = H.make_conv_desc(NB, Hw, Hw, Cw, Hw, Hw, Cw, 3, 3, 1, 1, 0, H.ACT_SILU, 1)
    wws = H.workspace(H.query("favae_conv_wgrad_workspace", byref(cd)), dev)
    dwt = torch.empty(Cw, 3, 3, Cw, device=dev)
    pend = []
    inproc = False
    wg_aggr = True
else:
    wg_aggr = False
if inproc:
    side = torch.cuda.Stream()
    ma = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    mb = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    mc = torch.empty(8192, 8192, device=dev, dtype=torch.float16)
    pend = []
for r in range(reps):
    if wg_aggr and r % 8 == 0:
        pend = [e for e in pend if not e.query()]
        while len(pend) < 6:
            with torch.cuda.stream(side):
                H.call("favae_conv_wgrad", byref(cd), H.ptr(xa), H.ptr(ya), H.ptr(sc2), H.ptr(sh2), H.ptr(xb2), H.ptr(yb2), H.ptr(dwt), 0,
                       H.ptr(wws), wws.numel())
                e = torch.cuda.Event()
                e.record(side)
            pend.append(e)
    if inproc and r % 16 == 0:
        pend = [e for e in pend if not e.query()]
        while len(pend) < 4:                      # keep a few GEMMs (about 1.5 ms each) queued on the side stream
            with torch.cuda.stream(side):
                torch.mm(ma, mb, out=mc)
                e = torch.cuda.Event()
                e.record(side)
            pend.append(e)
    gp, ws = bwd()
    torch.cuda.synchronize()
    a, b = not torch.equal(ws, ws0), not torch.equal(gp, gp0)
    if a or b:
        badA += a
        badB += b
        if badA + badB <= 10:
            ix = (ws != ws0).nonzero().flatten()
            if ix.numel():
                i0 = int(ix[0]) & ~1
                inner = 33 * C
                print("[%s]    scratch [n][h][k][c][2]: first differing complex element %d = (n %d, h %d, k %d, c %d); got %s want %s; runs of differing indices: %s"
                      % (tag, i0 // 2, i0 // 2 // (Hh * inner), (i0 // 2 // inner) % Hh, (i0 // 2 // C) % 33, (i0 // 2) % C,
                         ws[i0:i0 + 8].tolist(), ws0[i0:i0 + 8].tolist(),
                         [(int(a), int(b)) for a, b in zip(ix[:1].tolist() + ix[1:][(ix[1:] - ix[:-1]) > 2].tolist(), ix[:-1][(ix[1:] - ix[:-1]) > 2].tolist() + ix[-1:].tolist())][:6]), flush=True)
            print("[%s] rep %d: scratch (pass A output) differs %s (%d elements, first %s), gradient differs %s (%d elements)"
                  % (tag, r, a, ix.numel(), ix[:8].tolist(), b, int((gp != gp0).sum())), flush=True)
print("[%s] done: pass A output differed %d times, final gradient %d times of %d" % (tag, badA, badB, reps), flush=True)
