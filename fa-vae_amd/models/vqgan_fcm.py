"""VQGANFCM -- top-level FA-VAE model on the MI355X-native kernels.

Drop-in for the reference's `models/vqgan_fcm.py` (constructor :44-48, variant selection :58-96, encode :112,
decode :120, two-stage forward :124-149): same kwargs, same attributes (`encoder`, `decoder`, `quantizer`,
`discriminator`, optional `sigmas`), same return tuples, same state_dict keys.
"""
import torch
import torch.nn as nn

from favae_hip import ops as K

from .codec import (DecoderFcm, DecoderFcmAttnGauss, DecoderFcmGauss, DecoderFcmGaussSame, DecoderFcmGaussSameResblock,
                    DecoderFcmResGauss, Encoder, EncoderGauss)
from .discriminator import Discriminator, PatchDiscriminator


class VQGANFCM(nn.Module):
    def __init__(self, codebook_size, n_embed, double_z=False, ch_mult=(1, 2, 4, 8), attn_resolutions=[], use_cosine_sim=False,
                 codebook_dim=None, orthogonal_reg_weight=0, orthogonal_reg_max_codes=None,
                 orthogonal_reg_active_codes_only=False, use_l2_quantizer=False, sync_codebook=False, commitment_weight=1.0,
                 kernel_size=0, dsl_init_sigma=None, use_non_pair_conv=False, device=None, use_gauss_resblock=False,
                 use_gauss_attn=False, use_same_conv_gauss=False, use_same_gauss_resblock=False, use_ffl_with_fcm=False,
                 inference=False, num_groups=32, use_patch_discriminator=False, disc_n_layers=None):
        super().__init__()
        self.inference = inference
        self.use_same_gauss = bool(use_same_conv_gauss or use_same_gauss_resblock)
        self.device = device
        enc_kw = dict(z_channels=n_embed, double_z=double_z, ch_mult=ch_mult, attn_resolutions=attn_resolutions)
        dec_kw = dict(z_channels=n_embed, ch_mult=ch_mult, attn_resolutions=attn_resolutions)
        gauss_kw = dict(kernel_size=kernel_size, dsl_init_sigma=dsl_init_sigma, device=device)

        # Variant table (reference models/vqgan_fcm.py:58-96, first matching flag wins, in the reference's order):
        #   flag -> (encoder class, decoder class, model-level pair-wise sigmas?, decoder takes the gaussian kwargs?)
        variants = (
            (use_non_pair_conv, EncoderGauss, DecoderFcmGauss, False, True),             # non pair-wise DSL, convolutional FCM
            (use_same_conv_gauss, Encoder, DecoderFcmGaussSame, True, False),            # pair-wise DSL, convolutional FCM
            (use_same_gauss_resblock, Encoder, DecoderFcmGaussSameResblock, True, False),  # pair-wise DSL, residual FCM
            (use_gauss_resblock, EncoderGauss, DecoderFcmResGauss, False, True),         # non pair-wise DSL, residual FCM (BASELINE configs 1-3)
            (use_gauss_attn, EncoderGauss, DecoderFcmAttnGauss, False, True),            # non pair-wise DSL, attention FCM
            (use_ffl_with_fcm, Encoder, DecoderFcm, False, False),                       # convolutional FCM + FFL, no blur
        )
        for flag, enc_cls, dec_cls, pairwise, dec_gauss in variants:
            if not flag:
                continue
            if use_non_pair_conv:
                self.gauss_kernels = None                    # attribute the reference sets for this variant only
            if pairwise:                                     # one sigma per (encoder tap, decoder tap) pair, owned by the model
                self.sigmas = nn.Parameter(torch.tensor([dsl_init_sigma] * 4), requires_grad=True)
                self.kernel_size = kernel_size
                self.padding = [kernel_size // 2] * 4
            self.encoder = enc_cls(**enc_kw, **(gauss_kw if enc_cls is EncoderGauss else {}))
            if dec_gauss:
                self.decoder = dec_cls(**dec_kw, **gauss_kw)
            elif dec_cls is DecoderFcm:
                self.decoder = dec_cls(**dec_kw)
            elif dec_cls is DecoderFcmGaussSame:
                self.decoder = dec_cls(**dec_kw, kernel_size=kernel_size, device=device, num_groups=num_groups)
            else:
                self.decoder = dec_cls(**dec_kw, kernel_size=kernel_size, device=device)
            break

        self.use_l2_quantizer = use_l2_quantizer
        if use_l2_quantizer:
            from .l2_quantize import VectorQuantize
            self.quantizer = VectorQuantize(codebook_size=codebook_size, dim=n_embed, accept_image_fmap=True,
                                            use_cosine_sim=use_cosine_sim, codebook_dim=codebook_dim,
                                            orthogonal_reg_weight=orthogonal_reg_weight,
                                            orthogonal_reg_max_codes=orthogonal_reg_max_codes,
                                            orthogonal_reg_active_codes_only=orthogonal_reg_active_codes_only,
                                            sync_codebook=sync_codebook, commitment_weight=commitment_weight)
        self.discriminator = PatchDiscriminator(n_layers=disc_n_layers) if use_patch_discriminator else Discriminator()

    def _gaussian_blur(self, x, i, device=None):
        return K.gaussian_blur(x, self.sigmas, i, self.kernel_size)

    def encode(self, x):
        z, enc_feats = self.encoder(x, inference=self.inference)
        z_q, indices, loss_q = self.quantizer(z)
        return z_q, loss_q, indices, enc_feats

    def decode(self, z):
        return self.decoder(z, inference=self.inference)

    def forward(self, x, stage=0, inference=False):
        if stage == 0:                                              # train E + G + Q
            # a loop with ordinary gradient tensors (not TrainStep / FlatAdam's direct accumulation): the dense conv weights go through
            # identity nodes created HERE, before anything else of the pass, so that their gradients can be formed on the second stream
            # and are delivered at the end of backward (favae_hip/ops.py, _LateGradFn)
            if getattr(self, "_late_params", None) is None:
                self._late_params = [p for m in (self.encoder, self.decoder) for p in m.parameters() if p.dim() == 4]
            K.late_weights(self._late_params if self.training and torch.is_grad_enabled() else ())
            z, loss_q, _, enc_feats = self.encode(x)
            x_recon, dec_feats = self.decode(z)
            logits_fake = self.discriminator(x_recon)
            if self.use_same_gauss and not inference:
                for i in range(len(enc_feats)):
                    enc_feats[i] = self._gaussian_blur(enc_feats[i], i)
                    dec_feats[3 - i] = self._gaussian_blur(dec_feats[3 - i], 3 - i)
            return x_recon, loss_q, logits_fake, z, enc_feats, dec_feats
        elif stage == 1:                                            # train D
            with torch.no_grad():
                z, loss_q, _, _ = self.encode(x)
                x_recon, _ = self.decode(z)
            logits_real = self.discriminator(x)
            logits_fake = self.discriminator(x_recon.detach())
            return logits_real, logits_fake
        raise ValueError(f"Invalid stage: {stage}")
