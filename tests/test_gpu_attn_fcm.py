"""GPU parity of the attention FCM (--use_gauss_attn): token-wise kernels (LayerNorm, materialised GroupNorm,
counter-based dropout, 8-head attention core), TransEncoderBlock / DecoderFcmAttnGauss against the reference golden
(tests/golden/attn_fcm.npz: dropout probabilities 0, and eval mode) and against the CPU oracle with dropout ON through the
shared counter-based mask (torch's generator stream cannot be reproduced on the device)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import favae_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rnd(shape, seed, scale=1.0):
    n = int(np.prod(shape))
    return (scale * (2 * O._hash_uniform(n, seed).reshape(shape) - 1)).float()


@pytest.mark.parametrize("shape", [(2, 64, 6, 5), (1, 512, 16, 16), (3, 2048, 2, 3), (2, 36, 4, 4)])
def test_layer_norm_fwd_bwd(shape):
    from favae_hip import ops as K
    N, C, H, W = shape
    x = rnd(shape, 1, 2.0).requires_grad_(True)
    w = (1 + rnd((C,), 2, 0.3)).requires_grad_(True)
    b = rnd((C,), 3, 0.2).requires_grad_(True)
    y = F.layer_norm(x.permute(0, 2, 3, 1), (C,), w, b, 1e-5).permute(0, 3, 1, 2)
    gy = rnd(shape, 4)
    (y * gy).sum().backward()
    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = K.layer_norm(xd, wd, bd, 1e-5)
    (yd * gy.to(DEV)).sum().backward()
    assert rel(yd, y) < 5e-6
    assert rel(xd.grad, x.grad) < 2e-5
    assert rel(wd.grad, w.grad) < 2e-5 and rel(bd.grad, b.grad) < 2e-5


@pytest.mark.parametrize("act", ["none", "silu"])
def test_gn_apply_fwd_bwd(act):
    from favae_hip import ACT_NONE, ACT_SILU
    from favae_hip import ops as K
    shape = (2, 64, 9, 7)
    x = rnd(shape, 5, 1.5).requires_grad_(True)
    w = (1 + rnd((64,), 6, 0.3)).requires_grad_(True)
    b = rnd((64,), 7, 0.2).requires_grad_(True)
    y = F.group_norm(x, 32, w, b, 1e-5)
    if act == "silu":
        y = F.silu(y)
    gy = rnd(shape, 8)
    (y * gy).sum().backward()
    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = K.gn_apply(xd, wd, bd, 32, 1e-5, ACT_SILU if act == "silu" else ACT_NONE)
    (yd * gy.to(DEV)).sum().backward()
    assert rel(yd, y) < 5e-6
    assert rel(xd.grad, x.grad) < 5e-5
    assert rel(wd.grad, w.grad) < 1e-4 and rel(bd.grad, b.grad) < 1e-4


@pytest.mark.parametrize("relu", [False, True])
def test_dropout_matches_oracle_mask_bit_exact(relu):
    """favae_dropout against the oracle's restatement of the same counter-based mask: forward and backward are bit-identical
    (one fp32 multiply), ~10 % of the elements are dropped, a different seed gives a different mask."""
    from favae_hip import ops as K
    shape = (2, 40, 5, 6)
    x = rnd(shape, 9, 2.0)
    gy = rnd(shape, 10)
    K.set_dropout_seed(17)
    xd = x.to(DEV).requires_grad_(True)
    yd = K.dropout(xd, 0.1, True, relu=relu)
    (yd * gy.to(DEV)).sum().backward()
    xo = x.clone().requires_grad_(True)
    yo = O.dropout_like_hip(F.relu(xo) if relu else xo, 0.1, O.DropoutState(17), True)
    (yo * gy).sum().backward()
    assert torch.equal(yd.detach().cpu(), yo.detach())
    assert torch.equal(xd.grad.cpu(), xo.grad)
    if not relu:
        assert abs(float((yd == 0).float().mean()) - 0.1) < 0.03
    K.set_dropout_seed(18)
    assert not torch.equal(K.dropout(xd.detach(), 0.1, True, relu=relu).cpu(), yd.detach().cpu())
    # eval mode / p = 0: identity (or the plain ReLU), no seed consumed
    assert torch.equal(K.dropout(xd.detach(), 0.1, False).cpu(), x)
    assert torch.equal(K.dropout(xd.detach(), 0.1, False, relu=True).cpu(), F.relu(x))


@pytest.mark.parametrize("C,heads,hw", [(64, 8, (6, 5)), (256, 8, (16, 16)), (512, 8, (4, 8))])
def test_mha_core_vs_torch(C, heads, hw):
    from favae_hip import ops as K
    N, (H, W) = 2, hw
    L, dh = H * W, C // heads
    qkv = rnd((N, 3 * C, H, W), 11).requires_grad_(True)
    t = qkv.permute(0, 2, 3, 1).reshape(N, L, 3 * C)
    q, k, v = (u.view(N, L, heads, dh).transpose(1, 2) for u in t.split(C, dim=-1))
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(N, H, W, C).permute(0, 3, 1, 2)
    go = rnd((N, C, H, W), 12)
    (o * go).sum().backward()
    qd = qkv.detach().to(DEV).requires_grad_(True)
    od = K.mha_core(qd, heads, 0.1, False)
    (od * go.to(DEV)).sum().backward()
    assert rel(od, o) < 1e-5
    assert rel(qd.grad, qkv.grad) < 5e-5


def _trans_module(C, golden=None):
    from models.codec import TransEncoderBlock
    mod = TransEncoderBlock(C)
    sd = {k: O.det_value("blk." + k, tuple(v.shape)) for k, v in mod.state_dict().items()}
    mod.load_state_dict(sd, strict=True)
    return mod.to(DEV), {"blk." + k: v for k, v in sd.items()}


def _zero_dropout(m):
    for s in m.modules():
        if isinstance(s, torch.nn.Dropout):
            s.p = 0.0
        if isinstance(s, torch.nn.MultiheadAttention):
            s.dropout = 0.0


@pytest.mark.parametrize("name,C", [("trans64", 64), ("trans256", 256)])
def test_trans_encoder_block_against_reference_golden(golden_dir, name, C):
    g = np.load(os.path.join(golden_dir, "attn_fcm.npz"))
    mod, _ = _trans_module(C)
    x = torch.from_numpy(g[name + ".x"]).to(DEV).requires_grad_(True)
    mod.eval()
    with torch.no_grad():
        assert rel(mod(x), g[name + ".y_eval"]) < 2e-5
    _zero_dropout(mod)
    mod.train()
    y = mod(x)
    (y * torch.from_numpy(g[name + ".gy"]).to(DEV)).sum().backward()
    assert rel(y, g[name + ".y"]) < 2e-5
    assert rel(x.grad, g[name + ".gx"]) < 1e-4
    from test_oracle_golden import _check_grad_fixture
    for k, p in mod.named_parameters():
        _check_grad_fixture(g, name, k, p.grad, rel, 2e-4)


def test_trans_encoder_block_train_mode_dropout_vs_oracle():
    """dropout 0.1 ON at all four sites of the encoder layer (attention probabilities, dropout1, FFN dropout, dropout2)."""
    from favae_hip import ops as K
    C = 64
    mod, P = _trans_module(C)
    mod.train()
    x = rnd((2, C, 6, 5), 21)
    gy = rnd((2, C, 6, 5), 22)
    K.set_dropout_seed(5)
    xd = x.to(DEV).requires_grad_(True)
    y = mod(xd)
    (y * gy.to(DEV)).sum().backward()
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xo = x.clone().requires_grad_(True)
    yo = O.trans_encoder_block(Po, "blk", xo, training=True, drop=O.DropoutState(5))
    (yo * gy).sum().backward()
    assert rel(y, yo) < 2e-5
    assert rel(xd.grad, xo.grad) < 1e-4
    for k, p in mod.named_parameters():
        assert rel(p.grad, Po["blk." + k].grad) < 2e-4, k
    mod.eval()
    with torch.no_grad():
        ye = mod(xd.detach())
    assert rel(ye, O.trans_encoder_block(P, "blk", x, training=False)) < 2e-5
    assert rel(ye, yo) > 1e-3                      # the masks really were applied in train mode


def test_resnet_block_with_dropout_vs_oracle():
    """ResnetBlock(dropout=0.1) (fcm_4 of DecoderFcmAttnGauss, codec.py:1069): the mask sits between SiLU and the second conv."""
    from favae_hip import ops as K
    from models.codec import ResnetBlock
    C = 128
    mod = ResnetBlock(C, C, dropout=0.1)
    sd = {k: O.det_value("blk." + k, tuple(v.shape)) for k, v in mod.state_dict().items()}
    mod.load_state_dict(sd, strict=True)
    mod = mod.to(DEV).train()
    P = {"blk." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = rnd((2, C, 16, 16), 31)
    gy = rnd((2, C, 16, 16), 32)
    K.set_dropout_seed(9)
    xd = x.to(DEV).requires_grad_(True)
    y = mod(xd)
    (y * gy.to(DEV)).sum().backward()
    xo = x.clone().requires_grad_(True)
    yo = O.resnet_block(P, "blk", xo, dropout=0.1, training=True, drop=O.DropoutState(9))
    (yo * gy).sum().backward()
    assert rel(y, yo) < 2e-5
    assert rel(xd.grad, xo.grad) < 1e-4
    for k, p in mod.named_parameters():
        assert rel(p.grad, P["blk." + k].grad) < 2e-4, k
    mod.eval()
    with torch.no_grad():
        assert rel(mod(xd.detach()), O.resnet_block({k: v.detach() for k, v in P.items()}, "blk", x)) < 2e-5


MK = dict(codebook_size=256, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True,
          use_l2_quantizer=True, kernel_size=3, dsl_init_sigma=3.0, use_gauss_attn=True)
OK = dict(codebook_size=256, variant="gauss_attn", kernel_size=3)


def _model(**extra):
    from models.vqgan_fcm import VQGANFCM
    cfg = O.OracleConfig(**OK)
    state = O.det_state(cfg, with_disc=True)
    model = VQGANFCM(**MK, device=DEV, **extra)
    model.load_state_dict(state, strict=True)
    return model.to(DEV), cfg, state


def test_gauss_attn_model_against_reference_golden(golden_dir):
    from favae_step import TrainStep
    g = np.load(os.path.join(golden_dir, "attn_fcm.npz"))
    t = "gauss_attn_64."
    B, H, W, seed = (int(v) for v in g[t + "shape"])
    model, cfg, state = _model()
    _zero_dropout(model)
    ts = TrainStep(model, lr=1e-4, dsl_weight=0.01)
    x = O.det_input(B, H, W, seed)
    model.train()
    ts.gflat.zero_()
    out = ts.losses(x.to(DEV))
    out["loss_g"].sum().backward()
    torch.cuda.synchronize()
    xr = out["x_recon"]
    assert abs(float(xr.double().sum()) - float(g[t + "x_recon_sum"])) < 1e-4 * float(g[t + "x_recon_abs"])
    for k, gk in (("loss_l1", "loss_l1"), ("loss_ffl", "loss_ffl"), ("loss_dsl", "loss_dsl"), ("loss_g", "loss_g"), ("loss_quant", "loss_q")):
        assert rel(out[k].reshape(-1), g[t + gk]) < 1e-4, k
    named = dict(model.named_parameters())
    for k in ("decoder.fcm_1.attn.self_attn.in_proj_weight", "decoder.fcm_2.attn.linear1.weight", "decoder.fcm_3.attn.norm2.weight",
              "decoder.fcm_3.norm.weight", "decoder.fcm_4.block.6.weight", "encoder.conv_in.weight", "decoder.final.2.weight"):
        gr = named[k].grad
        assert abs(float(gr.double().sum()) - float(g[t + "g." + k + ".sum"])) < 5e-3 * float(g[t + "g." + k + ".abs"]), k
        assert rel(gr.reshape(-1)[:16], g[t + "g." + k + ".head"]) < 5e-2 or float(np.abs(g[t + "g." + k + ".head"]).max()) < 1e-6, k
    # codebook indices: a fresh model in train mode (the golden's indices are those of the first forward)
    model2, _, _ = _model()
    model2.train()
    with torch.no_grad():
        _, _, ind, _ = model2.encode(x.to(DEV))
    assert np.array_equal(ind.cpu().numpy().reshape(-1), g[t + "indices"].reshape(-1))
    # inference surface (eval, inference=True): taps are None, indices identical
    mi, _, _ = _model(inference=True)
    mi.eval()
    with torch.no_grad():
        zq, lq, ind, ef = mi.encode(x.to(DEV))
        xr, df = mi.decode(zq)
    assert all(f is None for f in df)
    assert np.array_equal(ind.cpu().numpy().reshape(-1), g[t + "inf.indices"].reshape(-1))
    assert abs(float(xr.double().sum()) - float(g[t + "inf.x_recon_sum"])) < 1e-4 * float(g[t + "inf.x_recon_abs"])


def test_gauss_attn_two_train_steps_with_dropout_vs_oracle():
    """Two full training steps, dropout ON (TrainStep seeds the masks from its step count, OracleTrainer likewise)."""
    from favae_step import TrainStep
    model, cfg, state = _model()
    ts = TrainStep(model, lr=1e-4, dsl_weight=0.01)
    orc = O.OracleTrainer(cfg, O.StepConfig(lr=1e-4, dsl_weight=0.01, with_disc_forward=True), state)
    for step in range(2):
        x = O.det_input(1, 64, 64, 300 + step)
        ro = orc.step(x)
        out = ts.step(x.to(DEV))
        for k in ("loss_l1", "loss_quant", "loss_ffl", "loss_dsl", "loss_g"):
            assert rel(out[k].reshape(-1), ro[k].reshape(-1)) < (3e-4 if step else 1e-4), (step, k)
        assert rel(out["x_recon"], ro["out"]["x_recon"]) < (3e-4 if step else 1e-4), step
    named = dict(model.named_parameters())
    worst = 0.0
    for k in orc.keys:
        go = orc.P[k].grad
        if go is not None and float(go.abs().max()) > 1e-6:
            worst = max(worst, rel(named[k].grad, go))
    assert worst < 3e-3, worst
