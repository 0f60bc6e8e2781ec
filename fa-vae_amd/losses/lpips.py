"""LPIPS perceptual loss of FA-VAE (reference losses/lpips.py:17-110) on the HIP kernels.

Same constructor, sub-module names and state_dict keys as the reference (`scaling_layer.{shift,scale}`,
`net.slice{1..5}.<torchvision features index>.{weight,bias}`, `lin{0..4}.model.1.weight`), so `vgg16_lpips.pt` loads with
`load_state_dict(strict=True)`.  The reference builds the VGG16 stack from `torchvision.models.vgg16().features`; torchvision
is not a dependency here, the topology (configuration "D": 2-2-3-3-3 3x3 convs, 64..512 channels, ReLU, 2x2 max pooling
between the blocks) is written out below with torchvision's layer indices.

Data path: the 13 convs are `ops.fused_conv` sites with ReLU applied on the operand load of the next conv, so every feature
tensor holds pre-activation values; pooling (`ops.MaxPool2Fn`) and the level distance (`ops.LpipsLevelFn`: ReLU, channel
normalisation, squared difference, 1x1 `lin`, spatial mean in one pass over the two feature tensors) account for that.
All parameters are frozen (losses/lpips.py:30-31): no weight gradients are computed; in the training step
`lpips(x, x_recon)` (favae_scripts/train_favae.py:77) the first argument is evaluated without a graph and only the second
carries a gradient.
"""
from pathlib import Path

import torch
import torch.nn as nn

import favae_hip as H
from favae_hip import ops as K

LIPIPS_PATH = Path(__file__).parent / "vgg16_lpips.pt"

# torchvision vgg16 "D": features index -> (Cin, Cout) of the 3x3 convs; ReLU follows each conv, 'M' = MaxPool2d(2, 2)
_VGG16_FEATURES = [(3, 64), (64, 64), "M", (64, 128), (128, 128), "M", (128, 256), (256, 256), (256, 256), "M",
                   (256, 512), (512, 512), (512, 512), "M", (512, 512), (512, 512), (512, 512)]
_SLICES = [(0, 4), (4, 9), (9, 16), (16, 23), (23, 30)]          # losses/lpips.py:88-96


def _vgg16_feature_layers():
    layers = []
    for item in _VGG16_FEATURES:
        if item == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(item[0], item[1], kernel_size=3, padding=1), nn.ReLU(inplace=True)]
    return layers                                                # 30 entries (the final pool, index 30, is not used)


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.Tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, inp):
        return K.ChannelAffineFn.apply(inp, self.shift.reshape(-1).contiguous(), self.scale.reshape(-1).contiguous())


class NetLinLayer(nn.Module):
    """A single linear layer which does a 1x1 conv (consumed inside ops.LpipsLevelFn)."""

    def __init__(self, chn_in, chn_out=1, use_dropout=False):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)


class vgg16(nn.Module):
    """Five slices of VGG16 features; forward returns the PRE-activation tensors whose ReLU the reference calls
    relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 (the consumers apply the ReLU)."""

    _C_FIRST = K.ConvCfg(3, 3, 1, 1, act=H.ACT_NONE)
    _C_RELU = K.ConvCfg(3, 3, 1, 1, act=H.ACT_RELU, norm="act")

    def __init__(self):
        super().__init__()
        feats = _vgg16_feature_layers()
        self.N_slices = 5
        for k, (lo, hi) in enumerate(_SLICES):
            sl = nn.Sequential()
            for i in range(lo, hi):
                sl.add_module(str(i), feats[i])
            setattr(self, "slice%d" % (k + 1), sl)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):                         # torchvision's initialisation of vgg16(pretrained=False)
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, X):
        h, outs, first = X, [], True
        for k in range(5):
            for layer in getattr(self, "slice%d" % (k + 1)):
                if isinstance(layer, nn.MaxPool2d):
                    h = K.MaxPool2Fn.apply(h)
                elif isinstance(layer, nn.Conv2d):
                    h = K.fused_conv(h, layer.weight, layer.bias, cfg=self._C_FIRST if first else self._C_RELU)
                    first = False
            outs.append(h)
        return outs


class LPIPS(nn.Module):
    """Learned perceptual metric.  `pretrained=False` (not in the reference, which always loads the file) keeps the random
    initialisation: the weight file is not redistributable with this repo, so benchmarks time the path on random weights."""

    def __init__(self, use_dropout=True, pretrained=True):
        super().__init__()
        self.scaling_layer = ScalingLayer()
        self.chns = [64, 128, 256, 512, 512]
        self.net = vgg16()
        for k, c in enumerate(self.chns):
            setattr(self, "lin%d" % k, NetLinLayer(c, use_dropout=use_dropout))
        if pretrained:
            self.load_from_pretrained()
        for param in self.parameters():
            param.requires_grad = False

    def load_from_pretrained(self):
        state = torch.load(LIPIPS_PATH, map_location="cpu")      # raises FileNotFoundError like the reference when absent
        self.load_state_dict(state)
        print("loaded pretrained VGG16 LPIPS loss from {}".format(LIPIPS_PATH))

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        for m in self.net.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        return out

    def forward(self, input, target):
        if self.training and any(isinstance(m, nn.Dropout) for m in self.lin0.model):
            raise NotImplementedError("LPIPS runs in eval mode in the reference (train_favae.py:308 `LPIPS().cuda().eval()`); "
                                      "the Dropout of the lin layers in train mode is not part of the accelerated path")
        if torch.is_tensor(input) and input.requires_grad:
            raise NotImplementedError("LPIPS: only the second argument carries a gradient (train_favae.py:77 lpips(x, x_recon))")
        with torch.no_grad():
            outs0 = self.net(self.scaling_layer(input))
        outs1 = self.net(self.scaling_layer(target))
        lins = [self.lin0, self.lin1, self.lin2, self.lin3, self.lin4]
        val = None
        for kk in range(len(self.chns)):
            val = K.LpipsLevelFn.apply(outs0[kk], outs1[kk], lins[kk].model[-1].weight, val)
        return val
