"""Feasibility probe: forward (and forward+backward) of the headline model on one batch-32 stream vs two batch-16 halves on two
HIP streams (matrix-bound kernels of one half overlapping the HBM-bound kernels of the other)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import favae_oracle as O
from favae_hip import ops as K
from models.vqgan_fcm import VQGANFCM
from losses.vqgan_losses import recon_ffl_loss, recon_ffl_features_loss
from focal_frequency_loss import FocalFrequencyLoss

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = VQGANFCM(16384, 256, use_cosine_sim=True, use_l2_quantizer=True, commitment_weight=1.0, kernel_size=9, dsl_init_sigma=3.0,
                 device=dev, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_gauss_resblock=True).to(dev)
x = O.det_input(32, 256, 256, 1234).to(dev)
xa, xb = x[:16].contiguous(), x[16:].contiguous()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
ffl, dsl = FocalFrequencyLoss(loss_weight=1.0), FocalFrequencyLoss(loss_weight=0.01)
K._SIDE["on"] = os.environ.get("SIDE", "1") == "1"


def loss_of(xx):
    xx = K.to_cl(xx)
    xr, lq, _, _, ef, df = model(xx, stage=0)
    l = K.l1_loss(xx, xr) + lq + recon_ffl_loss(ffl, xx, xr)
    l2, _ = recon_ffl_features_loss(dsl, ef, df, dev)
    return l + l2


def full(bwd):
    l = loss_of(x)
    if bwd:
        l.sum().backward()
        K.sync_side_stream()


def split(bwd):
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur); sB.wait_stream(cur)
    with torch.cuda.stream(sA):
        la = loss_of(xa)
    with torch.cuda.stream(sB):
        lb = loss_of(xb)
    cur.wait_stream(sA); cur.wait_stream(sB)
    if bwd:
        (0.5 * (la + lb)).sum().backward()
        K.sync_side_stream()
        cur.wait_stream(sA); cur.wait_stream(sB)


def timeit(fn, bwd, n=4):
    model.train()
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        fn(bwd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model.zero_grad(set_to_none=True)
        fn(bwd)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for bwd in (False, True):
    with torch.set_grad_enabled(bwd):
        a = timeit(full, bwd)
        b = timeit(split, bwd)
    print("%s: one stream batch 32: %.1f ms   two streams 2 x 16: %.1f ms" % ("fwd+bwd" if bwd else "fwd    ", a, b), flush=True)
