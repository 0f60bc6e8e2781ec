"""Frequency-loss glue of FA-VAE (reference losses/vqgan_losses.py:13-50), same function names and return values.

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
    """Spectrum Loss with a fixed sigma (vqgan_losses.py:34-50).  The reference blurs every feature with
    torchvision.transforms.GaussianBlur((k, k), sigma): kernel1d = normalised exp(-0.5 (x / sigma)^2) on
    linspace(-(k-1)/2, (k-1)/2, k), outer product, reflect padding k//2, depthwise conv -- the same operator as the codec's
    own _gaussian_blur (models/codec.py:255-277), so the same HIP blur kernel runs with a constant sigma (no sigma gradient).
    A (min, max) sigma range would be sampled per call by torchvision; the reference passes one float (--gaussian_sigma)."""
    from favae_hip import ops as K
    if isinstance(gaussian_sigma, (tuple, list)):
        if float(gaussian_sigma[0]) != float(gaussian_sigma[1]):
            raise NotImplementedError("a sigma range is sampled randomly by torchvision; pass a single sigma")
        gaussian_sigma = gaussian_sigma[0]
    sigma = torch.full((1,), float(gaussian_sigma), dtype=torch.float32, device=device)
    de_feat.reverse()                                        # in place (vqgan_losses.py:37)
    en_b = [K.gaussian_blur(f, sigma, 0, gaussian_kernel) for f in en_feat]
    de_b = [K.gaussian_blur(f, sigma, 0, gaussian_kernel) for f in de_feat]
    loss = torch.zeros(1, device=device)
    losses = []
    for i in range(len(en_b)):
        li = ffl(de_b[i], en_b[i])
        loss = loss + li
        losses.append(li)
    return loss / len(en_b), losses
