import os, sys, collections
sys.path.insert(0, "/root/repo/fa-vae_amd")
import torch
import favae_hip
from favae_hip import ops as K
from favae_step import TrainStep
from utils import synthetic_batch
from models.vqgan_fcm import VQGANFCM
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = VQGANFCM(16384, 256, use_cosine_sim=True, use_l2_quantizer=True, commitment_weight=1.0, kernel_size=9, dsl_init_sigma=3.0, device=dev,
                 ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_gauss_resblock=True).to(dev)
ts = TrainStep(model, lr=1e-4)
x = synthetic_batch(8, 256, 256, 1234).to(dev)
ts.step(x)
cnt = collections.Counter(); am = collections.Counter()
orig_call = K.call
def spy(name, *a):
    if name == "favae_gn_stats": cnt[tuple(a[3:7])] += 1
    if name == "favae_absmax": am[a[1]] += 1
    return orig_call(name, *a)
K.call = spy
ts.step(x)
torch.cuda.synchronize()
print("gn_stats streaming passes (N, HW, C, G):")
for k, v in sorted(cnt.items(), key=lambda kv: -kv[0][1] * kv[0][2]): print("  ", k, v)
print("absmax calls (numel):")
for k, v in sorted(am.items(), key=lambda kv: -kv[0]): print("  ", k, v)
