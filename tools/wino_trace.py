#!/usr/bin/env python3
"""Where does a Winograd workgroup's time go?  Runs conv3x3_wino_sp_kernel (forward with GroupNorm+SiLU fused, and the data gradient) from the
trace build (tools/wino_trace.sh -> tools/experiments/libfavae_trace.so) on one layer shape and prints, per phase of a K chunk, the cycles
lane 0 of every wave spent between the stamps (mean over 64 sampled workgroups x 8 waves x the steady-state chunks).
usage: python tools/wino_trace.py [cin cout hw batch]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "experiments", "libfavae_trace.so")
os.environ["FAVAE_HIP_LIB"] = LIB
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H
from favae_hip import ops as K

cin, cout, hw, B = (int(v) for v in (sys.argv[1:5] + ["128", "128", "256", "32"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
raw = ctypes.CDLL(LIB)
raw.favae_debug_wino_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
N = 64 * 8 * 48 * 8


def read():
    buf = np.zeros(N, dtype=np.uint64)
    rc = raw.favae_debug_wino_trace(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    return buf.reshape(64, 8, 48, 8).astype(np.int64)


def report(name, t, KC):
    ok = t[:, :, 0, 0] > 0
    print("== %s: %d sampled wave traces" % (name, int(ok.sum())))
    if not ok.any():
        return
    tt = t[ok]                                   # [traces][48][8]
    total = tt[:, 47, 4] - tt[:, 0, 0]
    pro = tt[:, 0, 1] - tt[:, 0, 0]
    epi = tt[:, 47, 4] - tt[:, 47, 0]
    print("  workgroup (wave) lifetime %.0f cycles: prologue %.0f, K loop %.0f, epilogue %.0f (exchange stores %.0f, barrier %.0f, finish+stores %.0f, stats %.0f)"
          % (total.mean(), pro.mean(), (tt[:, 47, 0] - tt[:, 0, 1]).mean(), epi.mean(), (tt[:, 47, 1] - tt[:, 47, 0]).mean(),
             (tt[:, 47, 2] - tt[:, 47, 1]).mean(), (tt[:, 47, 3] - tt[:, 47, 2]).mean(), (tt[:, 47, 4] - tt[:, 47, 3]).mean()))
    if (tt[:, 0, 2] > 0).all():                  # finer prologue stamps (WIDE build)
        names = ["start", "end", "loads issued, (scale, shift) staged", "halo 0 arrived + staged", "halo 1 staged", "barrier", "patch + transform 0"]
        order = [0, 2, 3, 4, 5, 6, 1]
        prev = tt[:, 0, 0]
        for k in order[1:]:
            print("     prologue: -> %-40s %6.0f cycles" % (names[k] if k != 1 else "last barrier", (tt[:, 0, k] - prev).mean()))
            prev = tt[:, 0, k]
    rows = [r for r in range(3, KC)]             # steady-state periods (generic copies of the loop body): chunk index r - 1
    if not rows:
        rows = [1]
    if True:
        groups = [("all waves", slice(0, 8), ["read_patch + mma(0) + load_b", "barrier 1 wait", "store_raw + load_raw", "transform + split + V stores",
                                             "mma(1..3) + load_b", "barrier 2 wait"])]
    else:
        groups = [("G0 (waves 0-3)", slice(0, 4), ["read patch + transform", "barrier X wait", "stage (halo store, loads, weight DMA issue)",
                                                    "B-fragment reads + 24 MFMAs", "barrier Y wait"]),
                  ("G1 (waves 4-7)", slice(4, 8), ["B-fragment reads + 24 MFMAs (previous chunk)", "read patch + barrier X wait",
                                                    "stage (halo store, loads, weight DMA issue)", "transform", "barrier Y wait"])]
    okg = (t[:, :, 0, 0] > 0).all(axis=1)
    for gname, sl, names in groups:
        c = t[okg][:, sl][:, :, rows, :]             # [wg][wave][rows][8]
        n = len(names)
        tot = (c[..., n] - c[..., 0]).mean()
        print("  -- %s: period %.0f cycles" % (gname, tot))
        for i, nm in enumerate(names):
            d = c[..., i + 1] - c[..., i]
            print("     %-46s %7.0f cycles  %5.1f %%   (p10 %6.0f  p90 %6.0f)" % (nm, d.mean(), 100 * d.mean() / tot, np.percentile(d, 10), np.percentile(d, 90)))
    print("  24 MFMAs = 768 pipe cycles per wave, 1536 per SIMD and chunk; %d chunks" % KC)


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
b = torch.zeros(cout, device=dev)
gw, gb = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
mean, rstd, scale, shift, xb = K.gn_stats(x, gw, gb, 32, with_bound=True)
y = K.new_cl(B, cout, hw, hw, dev)
d = H.make_conv_desc(B, hw, hw, cin, hw, hw, cout, 3, 3, 1, 1, 0, H.ACT_SILU, 1)
fwd = lambda: K._conv_launch(d, x, w, b, None, scale, shift, y, xb)
ms = timeit(fwd)
read()
fwd(); t = read()
print("forward %d->%d @%d batch %d: %.3f ms (with the stamps compiled in)" % (cin, cout, hw, B, ms))
report("forward <2,false,false>", t, cin // 16)
yb = K.absmax(y)
d2 = H.make_conv_desc(B, hw, hw, cout, hw, hw, cin, 3, 3, 1, 1, 0, 0, 1)
dx = K.new_cl(B, cin, hw, hw, dev)
wt = torch.empty(cin, 3, 3, cout, device=dev)
H.call("favae_weight_flip", H.ptr(w), H.ptr(wt), cout, 3, 3, cin)
dg = lambda: K._conv_launch(d2, y, wt, None, None, None, None, dx, yb)
ms = timeit(dg)
read()
dg(); t = read()
print("data gradient: %.3f ms" % ms)
report("data gradient <0,false,false>", t, cout // 16)
