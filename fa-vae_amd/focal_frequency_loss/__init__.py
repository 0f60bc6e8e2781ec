"""Drop-in for the pip package `focal-frequency-loss==0.3.0` (imported by the reference at
favae_scripts/train_favae.py:27 as `from focal_frequency_loss import FocalFrequencyLoss as FFL`), computed by the
hand-written gfx950 FFT kernels (fa-vae_amd/csrc/ffl.hip).

Only the configuration FA-VAE uses is accelerated: alpha=1.0, patch_factor=1, ave_spectrum=False, log_matrix=False,
batch_matrix=False, no external weight matrix (call sites train_favae.py:313,318,326).  Anything else raises.
Parity note: upstream source is not available offline -> semantics follow the published v0.3.0 algorithm and are
pinned by analytic known-answer tests only ("parity unpinned").
"""
import torch.nn as nn

from favae_hip import ops as _K


class FocalFrequencyLoss(nn.Module):
    def __init__(self, loss_weight=1.0, alpha=1.0, patch_factor=1, ave_spectrum=False, log_matrix=False, batch_matrix=False):
        super().__init__()
        if alpha != 1.0 or patch_factor != 1 or ave_spectrum or log_matrix or batch_matrix:
            raise NotImplementedError("only FocalFrequencyLoss(loss_weight, alpha=1.0) with package defaults is accelerated")
        self.loss_weight = loss_weight
        self.alpha = alpha
        self.patch_factor = patch_factor
        self.ave_spectrum = ave_spectrum
        self.log_matrix = log_matrix
        self.batch_matrix = batch_matrix

    def forward(self, pred, target, matrix=None, **kwargs):
        if matrix is not None:
            raise NotImplementedError("external spectrum weight matrices are not used by FA-VAE")
        return _K.focal_frequency_loss(pred, target, self.loss_weight)


__all__ = ["FocalFrequencyLoss"]
