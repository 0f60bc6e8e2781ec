#!/usr/bin/env python3
"""FAVAE_DYCS_FUSE=0 in a whole model: which parameters' gradients differ from the default path, and is the arm itself reproducible?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import favae_oracle as O
from favae_hip import ops as K
from favae_step import TrainStep
from models.vqgan_fcm import VQGANFCM
dev = torch.device("cuda", 0)
mk = dict(codebook_size=256, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True, use_l2_quantizer=True,
          kernel_size=3, dsl_init_sigma=3.0, use_gauss_resblock=True)
cfg = O.OracleConfig(codebook_size=256, variant="gauss_resblock", kernel_size=3)
state = O.det_state(cfg, with_disc=True)
xg = O.det_input(4, 64, 64, 5).to(dev)


def grads(dycs):
    model = VQGANFCM(**mk, sync_codebook=False, device=dev)
    model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    model = model.to(dev)
    K._DYCS_FUSE = dycs
    ts = TrainStep(model, lr=1e-4)
    model.train()
    ts.gflat.zero_()
    out = ts.losses(xg)
    ts.backward(out)
    K.sync_side_stream()
    torch.cuda.synchronize()
    K._DYCS_FUSE = True
    return model, ts, ts.gflat.clone()


_, _, g1 = grads(True)
model, ts, g0 = grads(False)
_, _, g0b = grads(False)
scale = float(g1.abs().max())
print("DYCS off vs on: %.3e of the max; off vs off again: %.3e" % (float((g0 - g1).abs().max()) / scale, float((g0 - g0b).abs().max()) / scale))
names = {id(p): n for n, p in model.named_parameters()}
off, rows = 0, []
for p in ts.params:
    n = p.numel()
    rows.append((float((g0[off:off + n] - g1[off:off + n]).abs().max()) / scale, names.get(id(p), "?"), tuple(p.shape)))
    off += n
for e, nm, shp in sorted(rows, reverse=True)[:14]:
    print("  %-50s %-22s %.3e" % (nm, shp, e))
print("parameters within 2e-5:", sum(1 for r in rows if r[0] < 2e-5), "of", len(rows))

# ---- where does it start?  post-hooks on every autograd node, default arm vs FAVAE_DYCS_FUSE=0, first node whose result differs
def trace(dycs):
    model = VQGANFCM(**mk, sync_codebook=False, device=dev)
    model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    model = model.to(dev)
    K._DYCS_FUSE = dycs
    ts = TrainStep(model, lr=1e-4)
    model.train()
    ts.gflat.zero_()
    out = ts.losses(xg)
    rec = []
    seen, stack = set(), [v.grad_fn for v in out.values() if torch.is_tensor(v) and v.grad_fn is not None]
    while stack:
        nd = stack.pop()
        if nd is None or nd in seen:
            continue
        seen.add(nd)

        def post(gin, gout, nm=type(nd).__name__):
            rec.append((nm, [(tuple(g.shape), g.detach().double().abs().sum().item()) for g in gin if torch.is_tensor(g) and g.is_floating_point()],
                        [(tuple(g.shape), g.detach().double().abs().sum().item()) for g in gout if torch.is_tensor(g) and g.is_floating_point()]))
        nd.register_hook(post)
        for nx, _ in nd.next_functions:
            stack.append(nx)
    ts.backward(out)
    K.sync_side_stream()
    torch.cuda.synchronize()
    K._DYCS_FUSE = True
    return rec


ra, rb = trace(True), trace(False)
print("nodes:", len(ra), len(rb))
for i, (a, b) in enumerate(zip(ra, rb)):
    bad = a[0] != b[0] or len(a[1]) != len(b[1]) or any(abs(u[1] - v[1]) > 1e-3 * (abs(u[1]) + 1e-30) for u, v in zip(a[1], b[1]))
    if bad:
        print("first differing node %d: %s" % (i, a[0]))
        print("   default : incoming %s -> returned %s" % (a[2], a[1]))
        print("   dycs off: incoming %s -> returned %s" % (b[2], b[1]))
        print("   previous nodes:", [r[0] for r in ra[max(0, i - 5):i]])
        break

# ---- library calls of the first ~25 backward nodes in both arms
import favae_hip as H_


def calls_of(dycs, upto=24):
    model = VQGANFCM(**mk, sync_codebook=False, device=dev)
    model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    model = model.to(dev)
    K._DYCS_FUSE = dycs
    ts = TrainStep(model, lr=1e-4)
    model.train()
    ts.gflat.zero_()
    out = ts.losses(xg)
    log = []
    cnt = [0]
    seen, stack = set(), [v.grad_fn for v in out.values() if torch.is_tensor(v) and v.grad_fn is not None]
    while stack:
        nd = stack.pop()
        if nd is None or nd in seen:
            continue
        seen.add(nd)

        def post(gin, gout, nm=type(nd).__name__):
            cnt[0] += 1
            log.append("== end of node %d %s" % (cnt[0] - 1, nm))
        nd.register_hook(post)
        for nx, _ in nd.next_functions:
            stack.append(nx)

    def hook(name, args, launch):
        if cnt[0] <= upto:
            log.append("   %s(%s)" % (name, ", ".join(("%#x" % a if isinstance(a, int) and a > 1 << 20 else str(a))[:14] for a in args)))
        return launch()
    H_.set_call_hook(hook)
    ts.backward(out)
    K.sync_side_stream()
    torch.cuda.synchronize()
    H_.set_call_hook(None)
    K._DYCS_FUSE = True
    return log


la, lb = calls_of(True), calls_of(False)
import re
strip = lambda s: re.sub(r"0x[0-9a-f]+", "PTR", s)
print("---- default arm, nodes 19..22")
on = False
for l in la:
    if "end of node 18 " in l: on = True
    if on: print(strip(l)[:260])
    if "end of node 22 " in l: break
print("---- DYCS off arm, nodes 19..22")
on = False
for l in lb:
    if "end of node 18 " in l: on = True
    if on: print(strip(l)[:260])
    if "end of node 22 " in l: break
