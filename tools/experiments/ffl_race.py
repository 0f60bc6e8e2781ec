#!/usr/bin/env python3
"""Is the FFL forward / backward bit-reproducible while another process uses the GPU?  (tools/race_probe.py localised a schedule-dependent
difference to FFLFnBackward.)  Repeats focal_frequency_loss + backward on fixed inputs of the probe's five FFL sites and compares loss,
saved spectrum and gradients bit for bit with the first repetition.  Run two copies at once (tools/r04_ffl_race.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
from favae_hip import ops as K
tag, reps = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)
torch.manual_seed(3)
shapes = [(4, 3, 64, 64), (4, 128, 64, 64), (4, 128, 32, 32), (4, 256, 16, 16), (4, 512, 4, 4), (32, 3, 256, 256)]
if os.environ.get("FFL_SHAPES"):
    shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ["FFL_SHAPES"].split(",")]
cases = []
for shp in shapes:
    p = torch.randn(shp, device=dev).contiguous(memory_format=torch.channels_last)
    t = torch.randn(shp, device=dev).contiguous(memory_format=torch.channels_last)
    cases.append((p, t))
hold = []


def run(p, t):
    pp = p.clone().requires_grad_(True)
    tt = t.clone().requires_grad_(True)
    l = K.focal_frequency_loss(pp, tt, 1.0)
    spec = l.grad_fn.saved_tensors[0].clone()
    l.backward()
    torch.cuda.synchronize()
    return l.detach().clone(), spec, pp.grad.clone(), tt.grad.clone()


base = [run(p, t) for p, t in cases]
bad = 0
for r in range(reps):
    for ci, (p, t) in enumerate(cases):
        o = run(p, t)
        d = [not torch.equal(a, b) for a, b in zip(o, base[ci])]
        if any(d):
            bad += 1
            if bad <= 12:
                for nm, a, b in zip(("loss", "spec", "gpred", "gtarget"), o, base[ci]):
                    if nm in ("gpred", "spec") and not torch.equal(a, b):
                        # gpred is NHWC in memory (channels_last): flat memory index = ((n*H + h)*W + w)*C + c; spec = [N][H][Wh][C][2]
                        am = a.permute(0, 2, 3, 1).reshape(-1) if a.dim() == 4 else a
                        bm = b.permute(0, 2, 3, 1).reshape(-1) if b.dim() == 4 else b
                        ix = (am != bm).nonzero().flatten()
                        N_, C_, H_, W_ = p.shape
                        if nm == "gpred":
                            dec = [((int(i) // (C_ * W_ * H_)), (int(i) // (C_ * W_)) % H_, (int(i) // C_) % W_, int(i) % C_) for i in ix[:6]]
                        else:
                            Wh = W_ // 2 + 1
                            dec = [((int(i) // (2 * C_ * Wh * H_)), (int(i) // (2 * C_ * Wh)) % H_, (int(i) // (2 * C_)) % Wh, (int(i) // 2) % C_, int(i) % 2) for i in ix[:6]]
                        print("[%s]      %s: %d differ, first memory indices %s = %s ... last %d; values %s vs %s" % (tag, nm, ix.numel(), ix[:6].tolist(), dec, int(ix[-1]), am[ix[:3]].tolist(), bm[ix[:3]].tolist()), flush=True)
                nd = [int((a != b).sum()) for a, b in zip(o, base[ci])]
                print("[%s] rep %d shape %s differs: loss %s spec %s gpred %s gtarget %s; differing elements %s; max |dspec| %.3e of %.3e"
                      % (tag, r, tuple(p.shape), *d, nd, float((o[1] - base[ci][1]).abs().max()), float(base[ci][1].abs().max())), flush=True)
print("[%s] done: %d differing (repetition, shape) pairs of %d" % (tag, bad, reps * len(cases)), flush=True)
