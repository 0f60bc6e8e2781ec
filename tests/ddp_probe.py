"""Child process of tests/test_gpu_dist.py::test_model_under_torch_ddp: the drop-in modules wrapped the way the reference wraps them
(accelerate.prepare -> torch DDP with find_unused_parameters=True, favae_scripts/train_favae.py:238-243,309), one stage-0 iteration
with torch.optim.Adam, against the same iteration without DDP.  world_size 1 (one MI355X per box), backend nccl = RCCL."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

import favae_oracle as O
from focal_frequency_loss import FocalFrequencyLoss
from losses.vqgan_losses import recon_ffl_features_loss, recon_ffl_loss
from models.vqgan_fcm import VQGANFCM

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
mk = dict(codebook_size=256, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True, use_l2_quantizer=True,
          kernel_size=3, dsl_init_sigma=3.0, use_gauss_resblock=True, sync_codebook=True)
cfg = O.OracleConfig(codebook_size=256, variant="gauss_resblock", kernel_size=3)
state = O.det_state(cfg, with_disc=True)
x = O.det_input(2, 64, 64, 5).to(dev)
ffl, dsl = FocalFrequencyLoss(loss_weight=1.0, alpha=1.0), FocalFrequencyLoss(loss_weight=0.01, alpha=1.0)


def iteration(wrap):
    model = VQGANFCM(**mk, device=dev)
    model.load_state_dict(state, strict=True)
    model = model.to(dev)
    net = DDP(model, device_ids=[0], find_unused_parameters=True, broadcast_buffers=True) if wrap else model
    inner = net.module if wrap else net
    opt = torch.optim.Adam(list(inner.encoder.parameters()) + list(inner.decoder.parameters()) + list(inner.quantizer.parameters()),
                           lr=1e-4, betas=(0.5, 0.9))                                        # train_favae.py:292-302
    net.train()
    opt.zero_grad()
    x_recon, loss_q, logits_fake, _, enc_feats, dec_feats = net(x, stage=0)                  # train_favae.py:75
    loss = (x - x_recon).abs().mean() + loss_q + recon_ffl_loss(ffl, x, x_recon)
    l2, _ = recon_ffl_features_loss(dsl, enc_feats, dec_feats, dev)
    loss = loss + l2
    loss.sum().backward()
    opt.step()
    torch.cuda.synchronize()
    return float(loss.sum()), {k: v.detach().clone() for k, v in inner.state_dict().items()}


la, sa = iteration(False)
lb, sb = iteration(True)
assert abs(la - lb) <= 1e-6 * abs(la), (la, lb)
worst = 0.0
for k in sa:
    if sa[k].dtype.is_floating_point:
        worst = max(worst, float((sa[k] - sb[k]).abs().max()))
assert worst <= 2.1e-4, worst            # Adam's first step is +-lr: identical gradients -> identical parameters (bound: 2 lr)
print("loss %.6f / %.6f, max parameter difference after one Adam step %.3e" % (la, lb, worst))
dist.destroy_process_group()
print("DDP PROBE OK")
