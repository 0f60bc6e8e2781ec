"""Frequency-loss glue of FA-VAE (reference losses/vqgan_losses.py:13-30), same function names and return values.

`ffl` is any callable (pred, target) -> scalar; with `focal_frequency_loss.FocalFrequencyLoss` from this tree each
call is one fused HIP FFT-loss.  The reference evaluates every feature term twice (once for the sum, once for the
logging list, vqgan_losses.py:25-26); both evaluations are identical, so the value is computed once and reused.
"""
import torch


def recon_ffl_loss(ffl, x, x_recon):
    return ffl(x_recon, x)                                   # (pred, target) order, vqgan_losses.py:14


def recon_ffl_features_loss(ffl, en_feat, de_feat, device):
    de_feat.reverse()                                        # in place, like the reference (vqgan_losses.py:20)
    loss = torch.zeros(1, device=device)
    losses = []
    for i in range(len(en_feat)):
        li = ffl(de_feat[i], en_feat[i])
        loss = loss + li
        losses.append(li)
    return loss / len(en_feat), losses


def recon_sl_gaussian_features_loss(ffl, gaussian_kernel, gaussian_sigma, en_feat, de_feat, device):
    raise NotImplementedError("fixed-sigma Spectrum Loss (SL, torchvision GaussianBlur) is outside the accelerated hot "
                              "path (SURVEY 2.1: only train_favae_celeba.sh experiment 3 uses it)")
