#!/usr/bin/env python3
"""Stress for an intermittent gradient difference seen when two processes share one GPU (tests/dist_probe.py, round 4: the global-batch
reference gradient of ONE rank off by 1.1e-3 of the max once in ~70 runs).  No collectives here: this process repeats losses+backward
of a fresh non-distributed TrainStep on the same input and compares every flat gradient bit for bit with the first; run two copies
at the same time (tools/r04_race.sh).  On a difference: which parameters, how large."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import favae_oracle as O
from favae_hip import ops as K
from favae_step import TrainStep
from models.vqgan_fcm import VQGANFCM

tag = sys.argv[1] if len(sys.argv) > 1 else "p"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fresh = int(os.environ.get("RACE_FRESH", "1"))          # 1: a new model + TrainStep for every repetition (what the probe does)
dev = torch.device("cuda", 0)
VARIANT = os.environ.get("FAVAE_PROBE_VARIANT", "gauss_resblock")
flag = {"gauss_resblock": "use_gauss_resblock", "same_conv_gauss": "use_same_conv_gauss"}[VARIANT]
mk = dict(codebook_size=256, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True, use_l2_quantizer=True,
          kernel_size=3, dsl_init_sigma=3.0, **{flag: True})
cfg = O.OracleConfig(codebook_size=256, variant=VARIANT, kernel_size=3)
state = O.det_state(cfg, with_disc=True)
xg = O.det_input(4, 64, 64, 5).to(dev)


def make():
    model = VQGANFCM(**mk, sync_codebook=False, device=dev)
    model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    model = model.to(dev)
    return model, TrainStep(model, lr=1e-4, distributed=False)


FWD = {}


def grads(ts):
    ts.model.train()
    ts.gflat.zero_()
    GRADS.clear()
    FFLREC.clear()
    out = ts.losses(xg)
    FWD.clear()
    for k, v in out.items():
        if torch.is_tensor(v):
            FWD[k] = v.detach().double().sum().item() if v.numel() > 1 else float(v.detach())
    ts.backward(out)
    K.sync_side_stream()
    torch.cuda.synchronize()
    return ts.gflat.clone()


HOOK = {}


def _cs(prefix, v):
    if torch.is_tensor(v):
        HOOK[prefix] = (v.detach().double().sum().item(), v.detach().double().abs().sum().item())
    elif isinstance(v, (list, tuple)):
        for i, u in enumerate(v):
            _cs("%s.%d" % (prefix, i), u)


def hook(mod, inp, outp):              # checksums of (x_recon, loss_q, logits, z, enc_feats, dec_feats): where does a repetition first differ?
    HOOK.clear()
    _cs("out", outp)


GRADS = []                                # (module name, checksum of the gradient at its output), in backward order


def _grad_hooks(model):
    def fwd_hook(name):
        def h(mod, inp, outp):
            outs = outp if isinstance(outp, (list, tuple)) else (outp,)
            for j, o in enumerate(outs):
                if torch.is_tensor(o) and o.requires_grad and o.is_floating_point():
                    o.register_hook(lambda g, nm="%s#%d" % (name, j): GRADS.append((nm, g.detach().double().sum().item(), g.detach().double().abs().sum().item())))
        return h
    for name, m in model.named_modules():
        if name and os.environ.get("RACE_GRADHOOKS", "1") == "1":
            m.register_forward_hook(fwd_hook(name))


FFLREC = []
_ffl = K.focal_frequency_loss


def _ffl_rec(pred, target, loss_weight=1.0):
    l = _ffl(pred, target, loss_weight)
    cs = lambda t: (t.detach().double().sum().item(), t.detach().double().abs().sum().item())
    spec = l.grad_fn.saved_tensors[0]
    FFLREC.append((tuple(pred.shape), cs(pred), cs(target), float(l.detach()), cs(spec)))
    return l


K.focal_frequency_loss = _ffl_rec
import focal_frequency_loss as _fflmod
_fflmod._K.focal_frequency_loss = _ffl_rec
_losses = None


def grads(ts):
    ts.model.train()
    ts.gflat.zero_()
    GRADS.clear()
    FFLREC.clear()
    out = ts.losses(xg)
    FWD.clear()
    FWD.update(HOOK)
    for k, v in out.items():
        if torch.is_tensor(v):
            FWD[k] = v.detach().double().sum().item() if v.numel() > 1 else float(v.detach())
    if os.environ.get("RACE_NODEHOOKS", "1") == "1":          # a post-hook on EVERY autograd node: checksums of what its backward returned
        seen, stack = set(), [v.grad_fn for v in out.values() if torch.is_tensor(v) and v.grad_fn is not None]
        while stack:
            nd = stack.pop()
            if nd is None or nd in seen:
                continue
            seen.add(nd)

            def post(gin, gout, nm=type(nd).__name__):
                GRADS.append(("node:" + nm,) + tuple((g.detach().double().sum().item(), g.detach().double().abs().sum().item())
                                                     for g in gin if torch.is_tensor(g) and g.is_floating_point()))
            nd.register_hook(post)
            for nx, _ in nd.next_functions:
                stack.append(nx)
    ts.backward(out)
    K.sync_side_stream()
    torch.cuda.synchronize()
    return ts.gflat.clone()


model, ts = make()
model.register_forward_hook(hook)
_grad_hooks(model)
g0 = grads(ts)
f0 = dict(FWD)
gr0 = list(GRADS)
ffl0 = list(FFLREC)
scale = float(g0.abs().max())
ref_file = os.environ.get("RACE_REF")                  # compare the first gradient with the one another configuration saved
if ref_file:
    if os.path.exists(ref_file):
        gr = torch.load(ref_file).to(dev)
        print("[%s] first gradient vs %s: max diff %.3e of the max" % (tag, ref_file, float((g0 - gr).abs().max()) / float(gr.abs().max())), flush=True)
    else:
        torch.save(g0.cpu(), ref_file)
bad = 0
for r in range(reps):
    if fresh:
        model, ts = make()
        model.register_forward_hook(hook)
    else:
        model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)     # the EMA moved the codebook
    g = grads(ts)
    if not torch.equal(g, g0):
        bad += 1
        names = {id(p): n for n, p in model.named_parameters()}
        off, rows = 0, []
        for p in ts.params:
            n = p.numel()
            e = float((g[off:off + n] - g0[off:off + n]).abs().max()) / scale
            if e > 0:
                rows.append((e, names.get(id(p), "?"), off, n))
            off += n
        print("[%s] repetition %d differs: max %.3e of the max gradient in %d parameters; forward outputs that differ: %s"
              % (tag, r, max(x[0] for x in rows), len(rows), {k: (f0[k], FWD[k]) for k in f0 if FWD.get(k) != f0[k]} or "none"), flush=True)
        for i, (a, b) in enumerate(zip(ffl0, FFLREC)):
            if a != b:
                print("[%s]    FFL site %d %s: pred equal %s, target equal %s, loss equal %s (%r vs %r), spectrum equal %s (%r vs %r)"
                      % (tag, i, a[0], a[1] == b[1], a[2] == b[2], a[3] == b[3], a[3], b[3], a[4] == b[4], a[4], b[4]), flush=True)
        firsts = [(i, a[0], a[1:], b[1:]) for i, (a, b) in enumerate(zip(gr0, GRADS)) if a != b][:1]
        if firsts:
            i0 = firsts[0][0]
            print("[%s]    trail before the first difference: %s" % (tag, [g[0] for g in gr0[max(0, i0 - 6):i0]]), flush=True)
        print("[%s]    backward order: %d gradient hooks, first that differ: %s" % (tag, len(GRADS), firsts or "none"), flush=True)
        if firsts and firsts[0][0] > 0:
            print("[%s]    last equal hook before it: %s" % (tag, gr0[firsts[0][0] - 1][0]), flush=True)
        for e, nm, o, n in sorted(rows, reverse=True)[:2]:
            print("[%s]    %-52s [%9d, +%8d)  %.3e" % (tag, nm, o, n, e), flush=True)
print("[%s] done: %d of %d repetitions differ" % (tag, bad, reps), flush=True)
