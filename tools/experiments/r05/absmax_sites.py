#!/usr/bin/env python3
"""which call sites launch favae_absmax in one training step (shape, caller) -- candidates for a producer by-product"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
from favae_hip import ops as K

sites = collections.Counter()
import favae_hip as H_
orig_call = K.call
def spy_call(name, *args):
    if name in ("favae_absmax", "favae_gn_stats"):
        fr = [f for f in traceback.extract_stack()[:-1] if "favae" in f.filename or "models" in f.filename][-4:]
        sites[(name[6:], int(args[1]) if name == "favae_absmax" else tuple(int(v) for v in args[3:7]), " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr)))] += 1
    return orig_call(name, *args)
K.call = spy_call
import favae_hip; favae_hip.load()
from favae_step import TrainStep
from utils import synthetic_batch
from models.vqgan_fcm import VQGANFCM
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = VQGANFCM(16384, 256, use_cosine_sim=True, use_l2_quantizer=True, sync_codebook=False, commitment_weight=1.0, kernel_size=9,
                 dsl_init_sigma=3.0, device=dev, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_gauss_resblock=True).to(dev)
ts = TrainStep(model, lr=1e-4, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01, distributed=False, train_disc=False, lpips=None,
               perceptual_weight=1.0)
x = synthetic_batch(8, 256, 256, 1).to(dev)
import favae_step

ts.step(x)
sites.clear()
ts.step(x)
torch.cuda.synchronize()
for key, n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(n, *key)
print("total", sum(sites.values()))
