#!/usr/bin/env python3
"""host time to ENQUEUE one training step (python + ctypes + HIP launches, nothing waited for) against the GPU time of the step"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import favae_hip; favae_hip.load()
from favae_step import TrainStep
from utils import synthetic_batch
from models.vqgan_fcm import VQGANFCM
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = VQGANFCM(16384, 256, use_cosine_sim=True, use_l2_quantizer=True, sync_codebook=False, commitment_weight=1.0, kernel_size=9,
                 dsl_init_sigma=3.0, device=dev, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_gauss_resblock=True).to(dev)
ts = TrainStep(model, lr=1e-4, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01, distributed=False, train_disc=False, lpips=None,
               perceptual_weight=1.0)
x = synthetic_batch(B, 256, 256, 1).to(dev)
for _ in range(3):
    ts.step(x)
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    enq = []
    for _ in range(5):
        a = time.perf_counter()
        ts.step(x)
        enq.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("batch %d: enqueue per step %s ms; 5 steps enqueued in %.1f ms, finished in %.1f ms (%.1f ms/step)" % (
        B, " ".join("%.1f" % (1e3 * e) for e in enq), 1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e3 * (t2 - t0) / 5))
