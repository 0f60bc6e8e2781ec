"""One FA-VAE training step (stage 0) on the MI355X-native kernels -- restates the hot loop of the reference's
train() (favae_scripts/train_favae.py:68-106) for the BASELINE configurations (LPIPS and discriminator *training* off,
FFL + 4-level DSL on), with the optimizer of :292-301 (Adam, betas (0.5, 0.9), lr = base_lr * batch * world).

MI355X-first choices:
  * all trainable parameters, their gradients and both Adam moments live in four flat fp32 buffers; parameters are
    views into them, so the optimizer is ONE fused HIP kernel launch over the whole model and the data-parallel
    gradient exchange is ONE RCCL all-reduce over the flat gradient buffer (342 MB -> a few ms over xGMI);
  * the input batch is converted to channels-last once; every activation stays NHWC in HBM;
  * no host synchronisation inside the step: losses stay on the device (the reference's ten .item() calls per step,
    train_favae.py:118-119, are left to the caller's logging cadence).
"""
import torch
import torch.distributed as dist

from favae_hip import ops as K
from focal_frequency_loss import FocalFrequencyLoss
from losses.vqgan_losses import recon_ffl_features_loss, recon_ffl_loss


class TrainStep:
    def __init__(self, model, lr, betas=(0.5, 0.9), eps=1e-8, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01,
                 sigma_lr=2.0e-7, distributed=False):
        self.model = model
        self.lr, self.betas, self.eps, self.sigma_lr = lr, betas, eps, sigma_lr
        self.cw = codebook_weight
        self.ffl = FocalFrequencyLoss(loss_weight=ffl_weight, alpha=1.0) if ffl_weight > 0 else None
        self.dsl = FocalFrequencyLoss(loss_weight=dsl_weight, alpha=1.0) if dsl_weight > 0 else None
        self.distributed = distributed and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.distributed else 1
        self.t = 0
        # opt_g parameter set: encoder + decoder + quantizer (+ pair-wise model.sigmas at its own lr)
        main = list(model.encoder.parameters()) + list(model.decoder.parameters()) + list(model.quantizer.parameters())
        extra = [model.sigmas] if hasattr(model, "sigmas") else []
        self.params = main + extra
        self.n_main = sum(p.numel() for p in main)
        total = self.n_main + sum(p.numel() for p in extra)
        dev = main[0].device
        self.pflat = torch.empty(total, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.mflat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.vflat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.pflat[off:off + n].as_strided(p.shape, p.stride())
            view.copy_(p.data)
            p.data = view
            p.grad = self.gflat[off:off + n].as_strided(p.shape, p.stride())
            p._favae_flat = True          # ops._direct_grad: reductions accumulate straight into this (zeroed) view
            off += n

    def losses(self, x):
        """Forward + loss assembly of train() :75-102 (perceptual / adversarial terms off)."""
        m = self.model
        x = K.to_cl(x)
        x_recon, loss_q, _logits_fake, _z, enc_feats, dec_feats = m(x, stage=0)
        out = {"loss_l1": K.l1_loss(x, x_recon), "loss_quant": loss_q}
        loss_g = out["loss_l1"] + self.cw * loss_q
        if self.ffl is not None:
            out["loss_ffl"] = recon_ffl_loss(self.ffl, x, x_recon)
            loss_g = loss_g + out["loss_ffl"]
        if self.dsl is not None:
            out["loss_dsl"], out["loss_dsl_levels"] = recon_ffl_features_loss(self.dsl, enc_feats, dec_feats, x.device)
            loss_g = loss_g + out["loss_dsl"]
        out["loss_g"] = loss_g
        out["x_recon"] = x_recon
        return out

    def step(self, x):
        self.model.train()
        self.gflat.zero_()
        out = self.losses(x)
        out["loss_g"].sum().backward()
        if self.distributed:
            dist.all_reduce(self.gflat)                      # RCCL over xGMI; averaged inside the Adam kernel
        self.t += 1
        gs = 1.0 / self.world
        nm = self.n_main
        K.adam_step(self.pflat[:nm], self.gflat[:nm], self.mflat[:nm], self.vflat[:nm], self.t, self.lr, self.betas, self.eps, gs)
        if self.pflat.numel() > nm:
            K.adam_step(self.pflat[nm:], self.gflat[nm:], self.mflat[nm:], self.vflat[nm:], self.t, self.sigma_lr, self.betas,
                        self.eps, gs)
        return out
