"""Encoder / decoder zoo of FA-VAE on the MI355X-native kernels.

Drop-in for the reference's `models/codec.py`: same class names, constructor signatures, attribute paths and
state_dict keys (so `train_favae.py`, checkpoints and stage-2 code keep working), but every forward runs through
libfavae_hip (fa-vae_amd/csrc): GroupNorm statistics + SiLU are folded into the implicit-GEMM convolution that
consumes them, the residual add into its epilogue, nearest-upsampling / the asymmetric stride-2 padding into its
gather, and the learnable-sigma blur is one LDS-tiled stencil.  nn.Conv2d / nn.GroupNorm / nn.MultiheadAttention
objects are kept purely as parameter containers (=> identical default initialisation and key names).

Reference map (file:line in the reference repo):
  Upsample codec.py:11-18 | Downsample :21-31 | ResnetBlock :34-57 | NonResnetBlock :62-84 | AttnBlock :87-102
  Encoder :125-188 | EncoderGauss :193-314 | Decoder :400-466 | DecoderFcm :471-550 | DecoderFcmGauss :557-693
  DecoderFcmGaussSame :700-788 | DecoderFcmGaussSameResblock :794-876 | DecoderFcmResGauss :882-1004
  TransEncoderBlock :108-122 | DecoderFcmAttnGauss :1011-1129
"""
import torch
import torch.nn as nn

from favae_hip import ACT_NONE, ACT_SILU
from favae_hip import ops as K

_CL = torch.channels_last


def _conv_weights_channels_last(module):
    """Store 4-D conv weights as OHWI memory (logical shape stays (Cout,Cin,KH,KW)) so kernels read them zero-copy."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=_CL)
    return module


_C3 = K.ConvCfg(3, 3, 1, 1)                                   # GN+SiLU -> conv3x3 s1 p1
_C1 = K.ConvCfg(1, 1, 1, 0)
_CDOWN = K.ConvCfg(3, 3, 2, 0, pad_br=1)                      # F.pad(0,1,0,1) + conv s2 p0
_CUP = K.ConvCfg(3, 3, 1, 1, upsample=True)                   # nearest x2 + conv s1 p1


def _cfg_gn(groups, act=ACT_SILU, k=3):
    return K.ConvCfg(k, k, 1, k // 2, act=act, groups=groups)


class Upsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        return K.fused_conv(x, self.conv.weight, self.conv.bias, cfg=_CUP)


class Downsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, kernel_size=3, stride=2, padding=0)

    def forward(self, x):
        return K.fused_conv(x, self.conv.weight, self.conv.bias, cfg=_CDOWN)


class _NormActConvBlock(nn.Module):
    """GN-SiLU-Conv3 -GN-SiLU-Drop-Conv3 body shared by ResnetBlock and NonResnetBlock (keys block.{0,2,3,6})."""

    def __init__(self, in_c, out_c, dropout, num_groups=32):
        super().__init__()
        self.block = nn.Sequential(
            nn.GroupNorm(num_groups, in_c), nn.SiLU(),
            nn.Conv2d(in_c, out_c, kernel_size=3, stride=1, padding=1),
            nn.GroupNorm(num_groups, out_c), nn.SiLU(), nn.Dropout(dropout),
            nn.Conv2d(out_c, out_c, kernel_size=3, stride=1, padding=1),
        )
        self.has_shortcut = in_c != out_c
        if self.has_shortcut:
            self.shortcut = nn.Conv2d(in_c, out_c, kernel_size=1, stride=1, padding=0)
        self._cfg = _cfg_gn(num_groups)

    def _conv1(self, x, pass_input=False):
        b = self.block
        return K.fused_conv(x, b[2].weight, b[2].bias, b[0].weight, b[0].bias, None, self._cfg, pass_input)

    def _conv2(self, h, resid):
        b = self.block
        if self.training and b[5].p > 0.0:
            # Dropout between SiLU and the conv (only DecoderFcmAttnGauss.fcm_4 has p = 0.1, codec.py:1069): the activated
            # tensor has to exist to be masked, so this one site runs GroupNorm+SiLU materialised -> dropout -> plain conv
            t = K.gn_apply(h, b[3].weight, b[3].bias, self._cfg.groups, self._cfg.eps, ACT_SILU)
            t = K.dropout(t, b[5].p, True)
            return K.fused_conv(t, b[6].weight, b[6].bias, resid=resid, cfg=_C3)
        return K.fused_conv(h, b[6].weight, b[6].bias, b[3].weight, b[3].bias, resid, self._cfg)


class ResnetBlock(_NormActConvBlock):
    def __init__(self, in_c, out_c, dropout):
        super().__init__(in_c, out_c, dropout)

    def forward(self, x):
        # the skip path reads the alias of x returned by the first conv: its gradient is then added inside that conv's
        # GroupNorm-backward kernel (ops.FusedConvFn, pass_input) instead of by an autograd accumulation kernel
        h, xa = self._conv1(x, pass_input=True)
        skip = K.fused_conv(xa, self.shortcut.weight, self.shortcut.bias, cfg=_C1) if self.has_shortcut else xa
        return self._conv2(h, skip)                               # x + h, add fused in the second conv's epilogue


class NonResnetBlock(_NormActConvBlock):
    """FCM block with convolutional architecture: returns h only (no skip)."""

    def __init__(self, in_c, out_c, dropout, num_groups=32):
        super().__init__(in_c, out_c, dropout, num_groups)

    def forward(self, x):
        return self._conv2(self._conv1(x), None)


class AttnBlock(nn.Module):
    def __init__(self, in_c):
        super().__init__()
        self.norm = nn.GroupNorm(32, in_c)
        self.attn = nn.MultiheadAttention(in_c, num_heads=1, batch_first=True)
        self._cfg_in = K.ConvCfg(1, 1, 1, 0, act=ACT_NONE, groups=32)

    def forward(self, x):
        a = self.attn
        qkv, xa = K.fused_conv(x, a.in_proj_weight, a.in_proj_bias, self.norm.weight, self.norm.bias, None, self._cfg_in, True)
        o = K.AttnCoreFn.apply(qkv)
        return K.fused_conv(o, a.out_proj.weight, a.out_proj.bias, resid=xa, cfg=_C1)


class TransEncoderBlock(nn.Module):
    """FCM with attention architecture (codec.py:108-122): GroupNorm(32) -> nn.TransformerEncoderLayer(in_c, nhead=8,
    batch_first=True) on the (B, HW, C) tokens -> back to (B, C, H, W).  The layer's defaults are what the reference gets:
    dim_feedforward 2048, dropout 0.1, ReLU, post-norm, LayerNorm eps 1e-5.  The nn modules are parameter containers (same
    state_dict keys: norm.*, attn.self_attn.in_proj_*, attn.self_attn.out_proj.*, attn.linear1/2.*, attn.norm1/2.*); tokens are the
    channels-last memory of the 4-D tensor, so the linear layers are 1x1 convs on the split-precision matrix path."""

    def __init__(self, in_c):
        super().__init__()
        self.norm = nn.GroupNorm(32, in_c)
        self.attn = nn.TransformerEncoderLayer(in_c, nhead=8, batch_first=True)

    def forward(self, x):
        a = self.attn
        sa = a.self_attn
        tr = self.training
        h = K.gn_apply(x, self.norm.weight, self.norm.bias, 32, self.norm.eps)             # the residual branch is GN(x)
        # x = norm1(x + dropout1(self_attn(x)))   (TransformerEncoderLayer._sa_block, norm_first=False)
        qkv = K.fused_conv(h, sa.in_proj_weight, sa.in_proj_bias, cfg=_C1)
        o = K.mha_core(qkv, sa.num_heads, sa.dropout, tr)
        s = K.fused_conv(o, sa.out_proj.weight, sa.out_proj.bias, cfg=_C1)
        s = K.dropout(s, a.dropout1.p, tr)
        x1 = K.layer_norm(K.add(h, s), a.norm1.weight, a.norm1.bias, a.norm1.eps)
        # x = norm2(x + dropout2(linear2(dropout(relu(linear1(x))))))   (_ff_block)
        f = K.fused_conv(x1, a.linear1.weight, a.linear1.bias, cfg=_C1)
        f = K.dropout(f, a.dropout.p, tr, relu=True)
        f = K.fused_conv(f, a.linear2.weight, a.linear2.bias, cfg=_C1)
        f = K.dropout(f, a.dropout2.p, tr)
        return K.layer_norm(K.add(x1, f), a.norm2.weight, a.norm2.bias, a.norm2.eps)


def _final(block_in, out_ch, with_quant_conv=None):
    layers = [nn.GroupNorm(32, block_in), nn.SiLU(), nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)]
    if with_quant_conv is not None:
        layers.append(nn.Conv2d(with_quant_conv, with_quant_conv, kernel_size=1))
    return nn.Sequential(*layers)


class _BlurMixin:
    """Learnable-sigma Gaussian blur shared by the *Gauss* classes (codec.py:255-277)."""

    def _gaussian_blur(self, x, i):
        return K.gaussian_blur(x, self.sigmas, i, self.kernel_size)

    def _blur_tap(self, h, i):
        """(tensor to continue the trunk on, blurred tap): the trunk continues on an alias of h handed out by the blur node, so the
        trunk's gradient is added inside the blur-backward kernel (ops.BlurTapFn).  The tile statistics the producing conv left on h
        (ops.gn_stats) stay valid for the alias: same memory, same version counter."""
        if not (K._BLUR_TAP and torch.is_grad_enabled() and h.requires_grad):
            return h, self._gaussian_blur(h, i)
        st, am = getattr(h, "_favae_gnstats", None), getattr(h, "_favae_amax", None)
        h2, f = K.blur_tap(h, self.sigmas, i, self.kernel_size)
        if st is not None:
            h2._favae_gnstats = st
        if am is not None:
            h2._favae_amax = am
        return h2, f


class Encoder(nn.Module):
    def __init__(self, in_c=3, ch=128, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0,
                 resolution=256, z_channels=256, double_z=True):
        super().__init__()
        self.conv_in = nn.Conv2d(in_c, ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        blocks = []
        for level in range(len(ch_mult)):
            block_in = ch * in_ch_mult[level]
            block_out = ch * ch_mult[level]
            for _ in range(num_res_blocks):
                blocks.append(ResnetBlock(block_in, block_out, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    blocks.append(AttnBlock(block_in))
            if level != len(ch_mult) - 1:
                blocks.append(Downsample(block_in))
                curr_res = curr_res // 2
        self.down = nn.Sequential(*blocks)
        self.mid = nn.Sequential(ResnetBlock(block_in, block_in, dropout=dropout), AttnBlock(block_in),
                                 ResnetBlock(block_in, block_in, dropout=dropout))
        self.final = _final(block_in, 2 * z_channels if double_z else z_channels, with_quant_conv=z_channels)
        _conv_weights_channels_last(self)

    def _tap(self, h, i, inference):
        """-> (h to continue the trunk on, the tap appended to the feature list)"""
        return h, h

    def _final_fwd(self, h):
        f = self.final
        h = K.fused_conv(h, f[2].weight, f[2].bias, f[0].weight, f[0].bias, None, _C3)
        return K.fused_conv(h, f[3].weight, f[3].bias, cfg=_C1)

    def forward(self, x, inference=False):
        feats = []
        h = K.fused_conv(K.as_cl(x), self.conv_in.weight, self.conv_in.bias, cfg=_C3)
        h, f = self._tap(h, 0, inference)
        feats.append(f)
        h, f = self._tap(self.down(h), 1, inference)
        feats.append(f)
        h, f = self._tap(self.mid(h), 2, inference)
        feats.append(f)
        h, f = self._tap(self._final_fwd(h), 3, inference)
        feats.append(f)
        return h, feats


class EncoderGauss(Encoder, _BlurMixin):
    """Encoder whose four taps are blurred with its own learnable sigmas (non pair-wise DSL)."""

    def __init__(self, in_c=3, ch=128, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0,
                 resolution=256, z_channels=256, double_z=True, kernel_size=3, dsl_init_sigma=None, device=None):
        super().__init__(in_c, ch, ch_mult, num_res_blocks, attn_resolutions, dropout, resolution, z_channels, double_z)
        self.kernel_size = kernel_size
        self.device = device
        self.sigmas = nn.Parameter(torch.tensor([dsl_init_sigma] * 4), requires_grad=True)
        self.padding = [kernel_size // 2] * 4

    def _tap(self, h, i, inference):
        return (h, h) if inference else self._blur_tap(h, i)


class Decoder(nn.Module):
    """Plain decoder (no FCM); kept for API completeness (not used by VQGANFCM)."""

    def __init__(self, ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0,
                 resolution=256, z_channels=256):
        super().__init__()
        block_in = ch * ch_mult[len(ch_mult) - 1]
        self.quant_conv_in = nn.Conv2d(z_channels, z_channels, kernel_size=1)
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Sequential(ResnetBlock(block_in, block_in, dropout=dropout), AttnBlock(block_in),
                                 ResnetBlock(block_in, block_in, dropout=dropout))
        self.up, block_in = _make_up(ch, ch_mult, num_res_blocks, attn_resolutions, dropout, resolution, block_in)
        self.final = _final(block_in, out_ch)
        _conv_weights_channels_last(self)

    def forward(self, z):
        feats = []
        h = K.fused_conv(z, self.quant_conv_in.weight, self.quant_conv_in.bias, cfg=_C1)
        feats.append(h)
        h = K.fused_conv(h, self.conv_in.weight, self.conv_in.bias, cfg=_C3)
        feats.append(h)
        h = self.mid(h)
        feats.append(h)
        h = self.up(h)
        feats.append(h)
        f = self.final
        h = K.fused_conv(h, f[2].weight, f[2].bias, f[0].weight, f[0].bias, None, _C3)
        return h, feats


def _make_up(ch, ch_mult, num_res_blocks, attn_resolutions, dropout, resolution, block_in):
    blocks = []
    curr_res = resolution // 2 ** (len(ch_mult) - 1)
    for level in reversed(range(len(ch_mult))):
        block_out = ch * ch_mult[level]
        for _ in range(num_res_blocks + 1):
            blocks.append(ResnetBlock(block_in, block_out, dropout=dropout))
            if curr_res in attn_resolutions:
                blocks.append(AttnBlock(block_out))
            block_in = block_out
        if level != 0:
            blocks.append(Upsample(block_out))
            curr_res = curr_res * 2
    return nn.Sequential(*blocks), block_in


class _DecoderFcmBase(nn.Module, _BlurMixin):
    """All FCM decoders: fcm_1 -> conv_in -> fcm_2 -> mid -> fcm_3 -> up -> fcm_4 -> final.

    RES_FCM : FCMs are ResnetBlocks applied in sequence (codec.py:972-1004, 857-876)
              else NonResnetBlocks added back to the trunk, h_ = h_ + fcm(h_) (codec.py:528-550, 650-693, 764-788)
    OWN_SIGMAS : the decoder owns `sigmas` and blurs its taps itself (non pair-wise DSL)."""
    RES_FCM = False
    OWN_SIGMAS = False

    def __init__(self, ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0,
                 resolution=256, z_channels=256, kernel_size=0, dsl_init_sigma=None, device=None, num_groups=32):
        super().__init__()
        if self.OWN_SIGMAS:
            self.sigmas = nn.Parameter(torch.tensor([dsl_init_sigma] * 4), requires_grad=True)
        self.padding = [kernel_size // 2] * 4
        self.device = device
        self.kernel_size = kernel_size
        block_in = ch * ch_mult[len(ch_mult) - 1]
        self.fcm_1 = self._make_fcm(z_channels, 1, dropout, num_groups)
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.fcm_2 = self._make_fcm(block_in, 2, dropout, 32)
        self.mid = nn.Sequential(ResnetBlock(block_in, block_in, dropout=dropout), AttnBlock(block_in),
                                 ResnetBlock(block_in, block_in, dropout=dropout))
        self.fcm_3 = self._make_fcm(block_in, 3, dropout, 32)
        self.up, block_in = _make_up(ch, ch_mult, num_res_blocks, attn_resolutions, dropout, resolution, block_in)
        self.fcm_4 = self._make_fcm(block_in, 4, dropout, 32)
        self.final = _final(block_in, out_ch)
        _conv_weights_channels_last(self)

    def _make_fcm(self, c, index, dropout, num_groups):
        if self.RES_FCM:
            return ResnetBlock(c, c, dropout=dropout)
        return NonResnetBlock(c, c, dropout=dropout, num_groups=num_groups)

    def _tap(self, h, i, inference):
        """-> (h to continue on, the tap appended to the feature list)"""
        if not self.OWN_SIGMAS:
            return h, h
        if not inference:
            return self._blur_tap(h, i)
        return h, (None if self.RES_FCM else h)     # DecoderFcmResGauss appends None under inference (codec.py:973-1000)

    def forward(self, z, inference=False):
        # behind the quantizer no codebook index depends on a conv: with FAVAE_WINO4=2 the 256^2 layers may take the F(4x4, 3x3) kernel
        with K.wino4_forward(True):
            return self._forward(z, inference)

    def _forward(self, z, inference=False):
        feats = []
        conv_in = lambda t: K.fused_conv(t, self.conv_in.weight, self.conv_in.bias, cfg=_C3)
        z = K.as_cl(z)
        if self.RES_FCM:
            h, f = self._tap(self.fcm_1(z), 0, inference)
            feats.append(f)
            h, f = self._tap(self.fcm_2(conv_in(h)), 1, inference)
            feats.append(f)
            h, f = self._tap(self.fcm_3(self.mid(h)), 2, inference)
            feats.append(f)
            h, f = self._tap(self.fcm_4(self.up(h)), 3, inference)
            feats.append(f)
            trunk = h
        else:
            h, f = self._tap(self.fcm_1(z), 0, inference)
            feats.append(f)
            trunk = conv_in(K.add(h, z))
            h, f = self._tap(self.fcm_2(trunk), 1, inference)
            feats.append(f)
            trunk = self.mid(K.add(trunk, h))
            h, f = self._tap(self.fcm_3(trunk), 2, inference)
            feats.append(f)
            trunk = self.up(K.add(trunk, h))
            h, f = self._tap(self.fcm_4(trunk), 3, inference)
            feats.append(f)
            trunk = K.add(trunk, h)
        f = self.final
        out = K.fused_conv(trunk, f[2].weight, f[2].bias, f[0].weight, f[0].bias, None, _C3)
        return out, feats


class DecoderFcm(_DecoderFcmBase):
    def __init__(self, ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0,
                 resolution=256, z_channels=256):
        super().__init__(ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, dropout, resolution, z_channels)


class DecoderFcmGauss(_DecoderFcmBase):
    OWN_SIGMAS = True

    def __init__(self, ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0,
                 resolution=256, z_channels=256, kernel_size=0, dsl_init_sigma=None, device=None):
        super().__init__(ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, dropout, resolution, z_channels, kernel_size,
                         dsl_init_sigma, device)


class DecoderFcmGaussSame(_DecoderFcmBase):
    def __init__(self, ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0,
                 resolution=256, z_channels=256, kernel_size=0, device=None, num_groups=32):
        super().__init__(ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, dropout, resolution, z_channels, kernel_size,
                         None, device, num_groups)


class DecoderFcmGaussSameResblock(_DecoderFcmBase):
    RES_FCM = True

    def __init__(self, ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0,
                 resolution=256, z_channels=256, kernel_size=0, device=None):
        super().__init__(ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, dropout, resolution, z_channels, kernel_size,
                         None, device)


class DecoderFcmResGauss(_DecoderFcmBase):
    RES_FCM = True
    OWN_SIGMAS = True

    def __init__(self, ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0,
                 resolution=256, z_channels=256, kernel_size=0, dsl_init_sigma=None, device=None):
        super().__init__(ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, dropout, resolution, z_channels, kernel_size,
                         dsl_init_sigma, device)


class DecoderFcmAttnGauss(_DecoderFcmBase):
    """--use_gauss_attn (codec.py:1011-1129, paper Table 2 row 9): fcm_1..3 are TransEncoderBlocks applied in sequence, fcm_4 a
    ResnetBlock with dropout 0.1; own sigmas; under `inference` the taps are None (codec.py:1101-1125)."""
    RES_FCM = True
    OWN_SIGMAS = True

    def __init__(self, ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0,
                 resolution=256, z_channels=256, kernel_size=0, dsl_init_sigma=None, device=None):
        super().__init__(ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, dropout, resolution, z_channels, kernel_size,
                         dsl_init_sigma, device)

    def _make_fcm(self, c, index, dropout, num_groups):
        return ResnetBlock(c, c, dropout=0.1) if index == 4 else TransEncoderBlock(c)
