"""Per-step losses of the headline config under the fp32-grade (h3) and the mixed-precision (h1) conv modes, same initial
state and batches: shows how far the 16-bit mode drifts from the fp32-grade trajectory.  python tools/precision_trace.py [steps] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import favae_oracle as O  # noqa: E402
from favae_hip import ops as K  # noqa: E402
from favae_step import TrainStep  # noqa: E402
from models.vqgan_fcm import VQGANFCM  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
xs = [O.det_input(B, 256, 256, 1234 + i).to(dev) for i in range(2)]


def run(mode):
    K.set_conv_mode(mode)
    torch.manual_seed(0)
    model = VQGANFCM(16384, 256, use_cosine_sim=True, use_l2_quantizer=True, commitment_weight=1.0, kernel_size=9, dsl_init_sigma=3.0,
                     device=dev, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_gauss_resblock=True).to(dev)
    ts = TrainStep(model, lr=4.5e-6 * 32, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01)
    rows = []
    for i in range(steps):
        out = ts.step(xs[i % 2])
        rows.append([float(out[k].reshape(-1)[0]) for k in ("loss_g", "loss_l1", "loss_quant", "loss_ffl", "loss_dsl")])
    return rows


a, b = run("h3"), run("h1")
print("step  mode  loss_g    l1        quant     ffl       dsl")
for i, (ra, rb) in enumerate(zip(a, b)):
    print("%4d  h3   " % i + "  ".join("%.6f" % v for v in ra))
    print("%4d  h1   " % i + "  ".join("%.6f" % v for v in rb))
