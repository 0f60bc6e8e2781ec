"""NLayer PatchGAN discriminator of FA-VAE (reference models/discriminator.py:193-218) -- forward on the HIP kernels.

VQGANFCM.forward(stage=0) always runs the discriminator on x_recon (reference models/vqgan_fcm.py:129) even while
it is not being trained, so its FORWARD is part of every training step: 4x4 stride-2/1 convolutions with the
BatchNorm(batch statistics)+LeakyReLU(0.2) of the previous layer folded into the conv's input load.  BatchNorm batch
statistics reuse the GroupNorm kernel with G == C and the batch folded into the pixel dimension; running statistics are
updated in train mode exactly like nn.BatchNorm2d (momentum 0.1, unbiased variance).

Training THROUGH the discriminator (BASELINE config 5: hinge losses, adaptive weight) is the next SURVEY 8(f) row:
the output therefore carries a grad_fn that raises instead of silently producing no gradient.
"""
from ctypes import byref

import torch
import torch.nn as nn

import favae_hip as H
from favae_hip import ops as K


class _DiscForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, disc, *params):
        return disc._forward_impl(x.detach())

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError("gradient through the discriminator (hinge/adaptive-weight terms, BASELINE config 5) "
                                  "is not accelerated yet -- SURVEY 8(f) item 1")


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

    def _conv(self, x, conv, scale, shift, act):
        N, Cin, Hin, Win = x.shape
        s, p, k = conv.stride[0], conv.padding[0], conv.kernel_size[0]
        Ho, Wo = (Hin + 2 * p - k) // s + 1, (Win + 2 * p - k) // s + 1
        y = K.new_cl(N, conv.out_channels, Ho, Wo, x.device)
        d = H.make_conv_desc(N, Hin, Win, Cin, Ho, Wo, conv.out_channels, k, k, s, p, H.GATHER_PLAIN, act, 0)
        H.call("favae_conv_fwd", byref(d), H.ptr(x), H.ptr(K.weight_ohwi(conv.weight)), H.ptr(conv.bias), None, H.ptr(scale),
               H.ptr(shift), H.ptr(y))
        return y

    @torch.no_grad()
    def _forward_impl(self, x):
        x = K.to_cl(x)
        dev = x.device
        f = self.features
        h = self._conv(x, f[0], None, None, H.ACT_NONE)
        # LeakyReLU after the first conv: identity affine + activation folded into the next conv's load
        scale = torch.ones(f[0].out_channels, device=dev)
        shift = torch.zeros(f[0].out_channels, device=dev)
        i = 2
        while i < len(f):
            conv, bn = f[i], f[i + 1]
            h = self._conv(h, conv, scale, shift, H.ACT_LEAKY02)
            N, C, Hh, Ww = h.shape
            if self.training or not bn.track_running_stats:
                mean = torch.empty((1, C), device=dev)
                rstd = torch.empty_like(mean)
                scale = torch.empty((1, C), device=dev)
                shift = torch.empty_like(scale)
                ws = H.workspace(H.query("favae_gn_workspace", 1, N * Hh * Ww, C), dev)
                H.call("favae_gn_stats", H.ptr(h), H.ptr(bn.weight), H.ptr(bn.bias), 1, N * Hh * Ww, C, C, bn.eps, H.ptr(mean),
                       H.ptr(rstd), H.ptr(scale), H.ptr(shift), None, H.ptr(ws), ws.numel())
                if self.training and bn.track_running_stats:
                    H.call("favae_bn_update_running", H.ptr(mean), H.ptr(rstd), C, N * Hh * Ww, bn.eps, bn.momentum,
                           H.ptr(bn.running_mean), H.ptr(bn.running_var))
                    bn.num_batches_tracked += 1
            else:
                inv = torch.rsqrt(bn.running_var + bn.eps)
                scale = (bn.weight * inv).contiguous()
                shift = (bn.bias - bn.running_mean * bn.weight * inv).contiguous()
            i += 3
        return self._conv(h, self.head, scale, shift, H.ACT_LEAKY02)

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _DiscForwardFn.apply(x, self, *list(self.parameters()))
        return self._forward_impl(x)


class PatchDiscriminator(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("PatchDiscriminator cannot be constructed in the reference either "
                                  "(kwarg mismatch models/vqgan_fcm.py:108 vs models/discriminator.py:142)")
