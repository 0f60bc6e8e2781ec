"""NLayer PatchGAN discriminator of FA-VAE (reference models/discriminator.py:193-218) on the HIP kernels.

VQGANFCM.forward(stage=0) always runs the discriminator on x_recon (reference models/vqgan_fcm.py:129), so its forward is
part of every training step; with discriminator training on (BASELINE config 5) its backward carries the hinge generator
term into the decoder (stage 0) and the hinge discriminator loss into its own parameters (stage 1).

Every layer is one fused conv (ops.FusedConvFn): the BatchNorm(batch statistics)+LeakyReLU(0.2) -- or the bare LeakyReLU after
the first conv -- of the previous layer is folded into the 4x4 conv's input load, BatchNorm statistics reuse the GroupNorm
kernels with one channel per group and the batch folded into the pixel dimension (forward and backward), running statistics
are updated in train mode exactly like nn.BatchNorm2d (momentum 0.1, unbiased variance).
"""
import torch
import torch.nn as nn

import favae_hip as H
from favae_hip import ops as K


class Discriminator(nn.Module):
    def __init__(self, in_channel=3, channel=64, num_layer=3):
        super().__init__()
        modules = [nn.Conv2d(in_channel, channel, kernel_size=4, stride=2, padding=1), nn.LeakyReLU(0.2, True)]
        chs = [channel * min(2 ** i, 8) for i in range(num_layer + 1)]
        for i in range(1, num_layer + 1):
            stride = 2 if i != num_layer else 1
            modules += [nn.Conv2d(chs[i - 1], chs[i], kernel_size=4, stride=stride, padding=1, bias=False),
                        nn.BatchNorm2d(chs[i]), nn.LeakyReLU(0.2, True)]
        self.features = nn.Sequential(*modules)
        self.head = nn.Conv2d(chs[-1], 1, kernel_size=4, stride=1, padding=1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)

    @staticmethod
    def _cfg(conv, norm):
        return K.ConvCfg(conv.kernel_size[0], conv.kernel_size[1], conv.stride[0], conv.padding[0], act=H.ACT_LEAKY02, norm=norm)

    def forward(self, x):
        f = self.features
        h = K.fused_conv(x, f[0].weight, f[0].bias, cfg=self._cfg(f[0], "group"))        # no input transform (gn_w is None)
        prev_bn = None
        i = 2
        convs = []
        while i < len(f):
            convs.append((f[i], f[i + 1]))
            i += 3
        convs.append((self.head, None))
        for conv, bn in convs:
            if prev_bn is None:                                    # LeakyReLU of the first conv's output, no normalisation
                h = K.fused_conv(h, conv.weight, conv.bias, cfg=self._cfg(conv, "act"))
            else:
                stats = K.bn_batch_stats(h, prev_bn, self.training)
                h = K.fused_conv(h, conv.weight, conv.bias, prev_bn.weight, prev_bn.bias, None, self._cfg(conv, "batch"),
                                 False, stats)
            prev_bn = bn
        return h


class PatchDiscriminator(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("PatchDiscriminator cannot be constructed in the reference either "
                                  "(kwarg mismatch models/vqgan_fcm.py:108 vs models/discriminator.py:142)")
