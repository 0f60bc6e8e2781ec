import sys, os, collections, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "fa-vae_amd"))
import torch, favae_hip
from favae_step import TrainStep
from utils import synthetic_batch
from models.vqgan_fcm import VQGANFCM
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = VQGANFCM(16384, 256, use_cosine_sim=True, use_l2_quantizer=True, commitment_weight=1.0, kernel_size=9, dsl_init_sigma=3.0,
                 device=dev, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_gauss_resblock=True).to(dev)
ts = TrainStep(model, lr=1e-4)
x = synthetic_batch(4, 256, 256, 1).to(dev)
ts.step(x)
sites = collections.Counter()
def hook(name, args, launch):
    if name == "favae_absmax":
        st = traceback.extract_stack()[:-2]
        key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[-4:])
        sites[(key, args[1])] += 1
    return launch()
favae_hip.set_call_hook(hook)
ts.step(x)
torch.cuda.synchronize()
for (k, n), c in sorted(sites.items(), key=lambda kv: -kv[1] * kv[0][1])[:25]:
    print(c, n, k)
print("total", sum(sites.values()))
