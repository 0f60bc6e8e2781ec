"""One FA-VAE training iteration on the MI355X-native kernels -- restates the hot loop of the reference's train()
(favae_scripts/train_favae.py:68-116): stage 0 (encoder + decoder + quantizer: L1 + commit + FFL + 4-level DSL, optionally
the hinge generator term with the adaptive weight of :32-39) and, with train_disc, stage 1 (discriminator, hinge loss), with
the optimizers of :292-305 (Adam, betas (0.5, 0.9), lr = base_lr * batch * world).  The perceptual term
`perceptual_weight * lpips(x, x_recon).mean()` (:77-79) is on when an LPIPS module (losses/lpips.py) is passed.

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
from losses.hinge import hinge_d_loss, hinge_g_loss
from losses.vqgan_losses import recon_ffl_features_loss, recon_ffl_loss, recon_sl_gaussian_features_loss


def _flatten(params, dev):
    """parameters, gradients and Adam moments of `params` as views into four flat fp32 buffers"""
    total = sum(p.numel() for p in params)
    pflat = torch.empty(total, dtype=torch.float32, device=dev)
    gflat = torch.zeros(total, dtype=torch.float32, device=dev)
    mflat = torch.zeros(total, dtype=torch.float32, device=dev)
    vflat = torch.zeros(total, dtype=torch.float32, device=dev)
    off = 0
    for p in params:
        n = p.numel()
        view = pflat[off:off + n].as_strided(p.shape, p.stride())
        view.copy_(p.data)
        p.data = view
        p.grad = gflat[off:off + n].as_strided(p.shape, p.stride())
        p._favae_flat = True          # ops._direct_grad: reductions accumulate straight into this (zeroed) view
        off += n
    return pflat, gflat, mflat, vflat


class TrainStep:
    def __init__(self, model, lr, betas=(0.5, 0.9), eps=1e-8, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01,
                 sigma_lr=2.0e-7, distributed=False, train_disc=False, disc_weight=0.75, lpips=None, perceptual_weight=1.0,
                 sl_weight=0.0, gaussian_kernel=None, gaussian_sigma=None):
        self.model = model
        self.lpips = lpips                                   # losses.lpips.LPIPS in eval mode (train_favae.py:308) or None
        self.pw = perceptual_weight if lpips is not None else 0.0
        self.lr, self.betas, self.eps, self.sigma_lr = lr, betas, eps, sigma_lr
        self.cw = codebook_weight
        self.train_disc, self.disc_weight = train_disc, disc_weight
        self.ffl = FocalFrequencyLoss(loss_weight=ffl_weight, alpha=1.0) if ffl_weight > 0 else None
        self.dsl = FocalFrequencyLoss(loss_weight=dsl_weight, alpha=1.0) if dsl_weight > 0 else None
        # --SL_weight / --gaussian_kernel / --gaussian_sigma: fixed-sigma Spectrum Loss (train_favae.py:101-103,324-326)
        self.sl = FocalFrequencyLoss(loss_weight=sl_weight, alpha=1.0) if sl_weight > 0 else None
        self.sl_kernel, self.sl_sigma = gaussian_kernel, gaussian_sigma
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
        self.pflat, self.gflat, self.mflat, self.vflat = _flatten(self.params, dev)
        if train_disc:                 # opt_d (train_favae.py:304-305)
            self.dparams = list(model.discriminator.parameters())
            self.dpflat, self.dgflat, self.dmflat, self.dvflat = _flatten(self.dparams, dev)

    # ------------------------------------------------------------------------------------------------------------
    # checkpoint wire format of the reference (train_favae.py:366-379): {"model", "opt_g", "opt_d", "epoch", "step", "loss_recon"}
    # with opt_* = torch.optim.Adam.state_dict().  The flat moment buffers are exported / imported in that layout, so a
    # checkpoint written here loads into the reference's optimizers (and vice versa).
    # ------------------------------------------------------------------------------------------------------------
    def _adam_state_dict(self, groups, mflat, vflat, step):
        state, pgroups, idx, off = {}, [], 0, 0
        for params, lr in groups:
            ids = []
            for p in params:
                n = p.numel()
                if step > 0:
                    state[idx] = {"step": torch.tensor(float(step)),
                                  "exp_avg": mflat[off:off + n].as_strided(p.shape, p.stride()).clone(),
                                  "exp_avg_sq": vflat[off:off + n].as_strided(p.shape, p.stride()).clone()}
                ids.append(idx)
                idx += 1
                off += n
            pgroups.append({"lr": lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                            "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                            "decoupled_weight_decay": False, "params": ids})
        return {"state": state, "param_groups": pgroups}

    def _load_adam_state(self, sd, params, mflat, vflat):
        off, step = 0, 0
        for i, p in enumerate(params):
            n = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                mflat[off:off + n].as_strided(p.shape, p.stride()).copy_(st["exp_avg"])
                vflat[off:off + n].as_strided(p.shape, p.stride()).copy_(st["exp_avg_sq"])
                step = max(step, int(st["step"]))
            else:
                mflat[off:off + n].zero_()
                vflat[off:off + n].zero_()
            off += n
        return step

    def _g_groups(self):
        n_main_params = len(self.params) - (1 if hasattr(self.model, "sigmas") else 0)
        groups = [(self.params[:n_main_params], self.lr)]
        if n_main_params < len(self.params):
            groups.append((self.params[n_main_params:], self.sigma_lr))       # train_favae.py:296-299
        return groups

    def opt_g_state_dict(self):
        return self._adam_state_dict(self._g_groups(), self.mflat, self.vflat, self.t)

    def opt_d_state_dict(self):
        if not self.train_disc:              # the reference creates (and saves) opt_d even when it never steps: empty state
            return self._adam_state_dict([(list(self.model.discriminator.parameters()), self.lr)], None, None, 0)
        return self._adam_state_dict([(self.dparams, self.lr)], self.dmflat, self.dvflat, self.t)

    def load_opt_state_dicts(self, opt_g=None, opt_d=None):
        """Resume the optimizer moments and step count from torch.optim.Adam state dicts (the reference saves them but only
        restores the model on --resume, train_favae.py:335-341; restoring them is optional here too)."""
        if opt_g is not None:
            self.t = self._load_adam_state(opt_g, self.params, self.mflat, self.vflat)
        if opt_d is not None and self.train_disc:
            self.t = max(self.t, self._load_adam_state(opt_d, self.dparams, self.dmflat, self.dvflat))

    def checkpoint(self, epoch, step, loss_recon=None):
        """the dict the reference hands to utils.save_model (train_favae.py:366-374)"""
        state = {"model": self.model.state_dict(), "opt_g": self.opt_g_state_dict(), "opt_d": self.opt_d_state_dict(),
                 "epoch": epoch, "step": step}
        if loss_recon is not None:
            state["loss_recon"] = loss_recon
        return state

    def losses(self, x):
        """Forward + loss assembly of train() :75-102."""
        m = self.model
        x = K.to_cl(x)
        x_recon, loss_q, _logits_fake, _z, enc_feats, dec_feats = m(x, stage=0)
        out = {"loss_l1": K.l1_loss(x, x_recon), "loss_quant": loss_q}
        loss_recon = out["loss_l1"]
        if self.pw > 0:                                      # train_favae.py:77-79
            out["loss_perceptual"] = self.lpips(x, x_recon).mean()
            loss_recon = loss_recon + self.pw * out["loss_perceptual"]
        out["loss_recon"] = loss_recon
        loss_g = loss_recon + self.cw * loss_q
        if self.ffl is not None:
            out["loss_ffl"] = recon_ffl_loss(self.ffl, x, x_recon)
            loss_g = loss_g + out["loss_ffl"]
        if self.dsl is not None:
            out["loss_dsl"], out["loss_dsl_levels"] = recon_ffl_features_loss(self.dsl, enc_feats, dec_feats, x.device)
            loss_g = loss_g + out["loss_dsl"]
        if self.sl is not None:                              # reverses dec_feats in place once more, like the reference
            out["loss_sl"], out["loss_sl_levels"] = recon_sl_gaussian_features_loss(self.sl, self.sl_kernel, self.sl_sigma,
                                                                                    enc_feats, dec_feats, x.device)
            loss_g = loss_g + out["loss_sl"]
        if self.train_disc:                                  # train_favae.py:82-88
            out["loss_disc"] = hinge_g_loss(_logits_fake)
            out["weight_d"], g_recon, g_disc = self.adaptive_weight(loss_recon, out["loss_disc"], x_recon)
            # loss_g.backward() would walk the LPIPS stack and the discriminator a second time to reach x_recon; their
            # gradients at x_recon are already known from the adaptive weight, so step() back-propagates
            #   rest = loss_g - loss_recon - w_d * disc_weight * loss_disc      (everything else, through the graph)
            #   x_recon <- g_recon + w_d * disc_weight * g_disc                 (handed in at x_recon)
            # -- the same total gradient for encoder / decoder / quantizer.  The discriminator's own stage-0 gradients, which
            # the reference computes here and discards at opt_d.zero_grad() (train_favae.py:109), are not computed at all.
            out["_bwd"] = (loss_g - loss_recon, x_recon, g_recon + (out["weight_d"] * self.disc_weight) * g_disc)
            loss_g = loss_g + out["weight_d"] * self.disc_weight * out["loss_disc"]
        out["logits_fake"] = _logits_fake
        out["loss_g"] = loss_g
        out["x_recon"] = x_recon
        return out

    def adaptive_weight(self, loss_recon, loss_disc, x_recon):
        """compute_adaptive_weight (train_favae.py:32-39): ratio of the gradient norms at decoder.final[2].weight, clamped to
        [0, 1e4] and detached.  Kept as a device scalar (the reference syncs with .item() here).
        d loss / d last = (d loss / d x_recon) pulled back through the final conv only, so each loss is back-propagated to
        x_recon ONCE (data gradients only: the LPIPS stack is frozen, the discriminator's stage-0 weight gradients are never
        used) and only the final conv's weight gradient is evaluated on top; the two x_recon gradients are returned for reuse
        by the main backward.  Returns (weight, d loss_recon / d x_recon, d loss_disc / d x_recon)."""
        last = self.model.decoder.final[2].weight
        with K.no_direct_grad():
            with K.grad_only("data"):
                g_recon = torch.autograd.grad(loss_recon, x_recon, retain_graph=True)[0]
                g_disc = torch.autograd.grad(loss_disc, x_recon, retain_graph=True)[0]
            with K.grad_only("param"):
                grad_recon = torch.autograd.grad(x_recon, last, g_recon, retain_graph=True)[0]
                grad_disc = torch.autograd.grad(x_recon, last, g_disc, retain_graph=True)[0]
        w = torch.norm(grad_recon) / (torch.norm(grad_disc) + 1e-4)
        return torch.clamp(w, 0.0, 1e4).detach(), g_recon.detach(), g_disc.detach()

    def backward(self, out):
        """loss_g.backward() of train_favae.py:105 on the dict losses() returned.  With discriminator training the gradients of
        loss_recon and loss_disc at x_recon were already computed for the adaptive weight and are handed in there (see losses())."""
        bwd = out.pop("_bwd", None)
        if bwd is None:
            out["loss_g"].sum().backward()
        else:
            rest, x_recon, g_x = bwd
            torch.autograd.backward([rest.sum(), x_recon], [None, g_x])

    def disc_step(self, x):
        """Stage 1 (train_favae.py:108-116): discriminator update on (x, x_recon.detach()); model(x, stage=1) recomputes the
        reconstruction under no_grad in train mode (second EMA codebook update of the iteration, as in the reference)."""
        self.dgflat.zero_()                                  # opt_d.zero_grad(): also drops what stage 0 left here
        logits_real, logits_fake = self.model(K.to_cl(x), stage=1)
        loss_d = hinge_d_loss(logits_real, logits_fake)
        loss_d.backward()
        K.sync_side_stream()
        if self.distributed:
            dist.all_reduce(self.dgflat)
        K.adam_step(self.dpflat, self.dgflat, self.dmflat, self.dvflat, self.t, self.lr, self.betas, self.eps, 1.0 / self.world)
        return {"loss_d": loss_d, "logits_real": logits_real, "logits_fake_d": logits_fake}

    def step(self, x):
        self.model.train()
        self.gflat.zero_()
        K.set_dropout_seed(self.t + 1)                       # dropout sites (attention FCM only): fresh masks every step
        out = self.losses(x)
        self.backward(out)
        K.sync_side_stream()                                 # weight gradients run on a second stream (ops._SIDE)
        if self.distributed:
            dist.all_reduce(self.gflat)                      # RCCL over xGMI; averaged inside the Adam kernel
        self.t += 1
        gs = 1.0 / self.world
        nm = self.n_main
        K.adam_step(self.pflat[:nm], self.gflat[:nm], self.mflat[:nm], self.vflat[:nm], self.t, self.lr, self.betas, self.eps, gs)
        if self.pflat.numel() > nm:
            K.adam_step(self.pflat[nm:], self.gflat[nm:], self.mflat[nm:], self.vflat[nm:], self.t, self.sigma_lr, self.betas,
                        self.eps, gs)
        if self.train_disc:
            out.update(self.disc_step(x))
        return out
