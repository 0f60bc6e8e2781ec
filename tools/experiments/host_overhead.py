"""Host-side cost of one training step: wall time for Python to ENQUEUE a step (no device sync) vs the device time of the step.
If enqueue << device time the step is GPU-bound and launch overhead is hidden (also with 8 ranks sharing the host's cores)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import favae_oracle as O
from favae_step import TrainStep
from models.vqgan_fcm import VQGANFCM

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = VQGANFCM(16384, 256, use_cosine_sim=True, use_l2_quantizer=True, commitment_weight=1.0, kernel_size=9, dsl_init_sigma=3.0,
                 device=dev, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_gauss_resblock=True).to(dev)
ts = TrainStep(model, lr=1e-4)
x = O.det_input(32, 256, 256, 1234).to(dev)
for _ in range(2):
    ts.step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
enq = []
for _ in range(4):
    a = time.perf_counter()
    ts.step(x)
    enq.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue per step (ms):", ["%.1f" % (1e3 * e) for e in enq], " total enqueue %.1f ms, device done after %.1f ms (%.1f ms/step)"
      % (1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e3 * (t2 - t0) / 4))
import resource
print("process CPU time so far: user %.1f s, sys %.1f s" % resource.getrusage(resource.RUSAGE_SELF)[:2])
