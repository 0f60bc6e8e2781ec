"""Per-tensor error of the stage-0 generator gradients of the GAN golden (tests/golden/gan_128.npz) on the HIP path."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import favae_oracle as O
from favae_step import TrainStep
from models.vqgan_fcm import VQGANFCM

g = np.load(os.path.join(ROOT, "tests", "golden", "gan_128.npz"))
B, H, W, seed = [int(v) for v in g["gan_128.shape"]]
lr, disc_w = [float(v) for v in g["gan_128.hyper"]]
mk = dict(codebook_size=512, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True,
          use_l2_quantizer=True, kernel_size=9, dsl_init_sigma=3.0, use_same_conv_gauss=True, num_groups=32)
cfg = O.OracleConfig(codebook_size=512, variant="same_conv_gauss", kernel_size=9, num_groups=32)
model = VQGANFCM(**mk, device="cuda:0")
model.load_state_dict(O.det_state(cfg, with_disc=True), strict=True)
model = model.to("cuda:0")
ts = TrainStep(model, lr=lr, train_disc=True, disc_weight=disc_w)
x = O.det_input(B, H, W, seed).to("cuda:0")
model.train()
ts.gflat.zero_()
out = ts.losses(x)
full = os.environ.get("FULL_BWD") == "1"
if full:
    out.pop("_bwd")
    out["loss_g"].sum().backward()
else:
    ts.backward(out)
torch.cuda.synchronize()
print("weight_d", float(out["weight_d"]), "golden", float(g["gan_128.weight_d"]), "full_bwd", full)
for k, p in model.named_parameters():
    key = "gan_128.g." + k + ".head"
    if key in g.files:
        a = p.grad.detach().cpu().reshape(-1)[:16].double().numpy()
        b = g[key].astype(np.float64)
        print("%-45s max-rel %.3e   |g|max %.3e" % (k, np.abs(a - b).max() / (np.abs(b).max() + 1e-30), np.abs(b).max()))
