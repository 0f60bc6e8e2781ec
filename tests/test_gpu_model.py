"""GPU parity of the whole model / training step (HIP path through the drop-in modules) against
 (a) the golden vectors captured from the reference implementation and (b) the CPU oracle on the same seeded inputs.
BASELINE bar: codebook indices bit-exact (near-ties with a reference top-2 gap < 1e-6 are reported separately),
reconstruction and FFL within 1e-4 relative (fp32)."""
import os

import numpy as np
import pytest
import torch

import favae_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"

MODEL_KW = {
    "cfg1": (dict(codebook_size=1024, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True,
                  use_l2_quantizer=True, kernel_size=9, dsl_init_sigma=3.0, use_gauss_resblock=True),
             dict(codebook_size=1024, variant="gauss_resblock", kernel_size=9)),
    "cfg1_k3": (dict(codebook_size=1024, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True,
                     use_l2_quantizer=True, kernel_size=3, dsl_init_sigma=3.0, use_gauss_resblock=True),
                dict(codebook_size=1024, variant="gauss_resblock", kernel_size=3)),
    # BASELINE configs[1] wiring at its codebook size
    "cfg2": (dict(codebook_size=16384, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True,
                  use_l2_quantizer=True, kernel_size=9, dsl_init_sigma=3.0, use_gauss_resblock=True),
             dict(codebook_size=16384, variant="gauss_resblock", kernel_size=9)),
    # BASELINE configs[3] model: f=4, embed_dim 3 -> codebook_dim 256, codebook 8192, conv FCM with one sigma per pair, k=9
    "f4_full": (dict(codebook_size=8192, n_embed=3, ch_mult=(1, 2, 4), attn_resolutions=[], use_cosine_sim=True, codebook_dim=256,
                     use_l2_quantizer=True, kernel_size=9, dsl_init_sigma=3.0, use_same_conv_gauss=True, num_groups=3),
                dict(codebook_size=8192, n_embed=3, ch_mult=(1, 2, 4), attn_resolutions=(), codebook_dim=256, kernel_size=9,
                     variant="same_conv_gauss", num_groups=3)),
    "f4_same_conv": (dict(codebook_size=512, n_embed=3, ch_mult=(1, 2, 4), attn_resolutions=[], use_cosine_sim=True,
                          codebook_dim=32, use_l2_quantizer=True, kernel_size=3, dsl_init_sigma=3.0, use_same_conv_gauss=True,
                          num_groups=3),
                     dict(codebook_size=512, n_embed=3, ch_mult=(1, 2, 4), attn_resolutions=(), codebook_dim=32, kernel_size=3,
                          variant="same_conv_gauss", num_groups=3)),
    "nonpair_conv": (dict(codebook_size=256, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True,
                          use_l2_quantizer=True, kernel_size=5, dsl_init_sigma=2.0, use_non_pair_conv=True),
                     dict(codebook_size=256, variant="non_pair_conv", kernel_size=5, dsl_init_sigma=2.0)),
    "same_resblock": (dict(codebook_size=256, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True,
                           use_l2_quantizer=True, kernel_size=3, dsl_init_sigma=3.0, use_same_gauss_resblock=True),
                      dict(codebook_size=256, variant="same_gauss_resblock", kernel_size=3)),
    "ffl_with_fcm": (dict(codebook_size=256, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True,
                          use_l2_quantizer=True, use_ffl_with_fcm=True),
                     dict(codebook_size=256, variant="ffl_with_fcm")),
}


import margins  # noqa: E402  (achieved errors of every bar -> gpurun_out/parity_margins.txt)


def build(tag):
    from models.vqgan_fcm import VQGANFCM
    mk, ok = MODEL_KW[tag]
    cfg = O.OracleConfig(**ok)
    state = O.det_state(cfg, with_disc=True)
    model = VQGANFCM(**mk, device=DEV)
    model.load_state_dict(state, strict=True)
    return model.to(DEV), cfg, state


def rel(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def close(a, b, tol, name):
    e = rel(a, b)
    margins.record(name, e, tol)
    assert e < tol, f"{name}: max-rel {e:.3e} >= {tol}"


def check_indices(got, ref, gap, name="indices"):
    got, ref, gap = np.asarray(got), np.asarray(ref), np.asarray(gap)
    mism = got != ref
    assert not (mism & (gap > 1e-6)).any(), f"{name}: mismatch outside flagged near-ties"
    return int(mism.sum())


# --------------------------------------------------------------------------------------------------------------
# (a) golden vectors from the reference
# --------------------------------------------------------------------------------------------------------------
GOLDEN_FWD = [("cfg1_96", "cfg1"), ("f4_same_conv_32", "f4_same_conv"), ("nonpair_conv_80", "nonpair_conv")]


@pytest.mark.parametrize("gtag,mtag", GOLDEN_FWD)
def test_forward_against_reference_golden(golden_dir, gtag, mtag):
    from favae_hip import ops as K
    g = np.load(os.path.join(golden_dir, "models.npz"))
    model, cfg, _ = build(mtag)
    B, H, W, seed = [int(v) for v in g[gtag + ".shape"]]
    x = O.det_input(B, H, W, seed).to(DEV)
    model.train()
    x_recon, loss_q, logits_fake, z, enc_feats, dec_feats = model(x, stage=0)
    p = gtag + "."
    close(loss_q, g[p + "loss_q"], LOSS_TOL, "loss_q")
    xr = x_recon.detach().cpu()
    close(xr[:, :, ::max(1, H // 8), ::max(1, W // 8)], g[p + "x_recon_slice"], XRECON_TOL, "x_recon")
    assert abs(float(xr.double().abs().sum()) - float(g[p + "x_recon_abs"])) < 1e-4 * float(g[p + "x_recon_abs"])
    close(K.l1_loss(x, x_recon), g[p + "loss_l1"], LOSS_TOL, "loss_l1")
    for i in range(4):
        s = float(enc_feats[i].detach().double().abs().sum())
        assert abs(s - float(g[p + f"enc_feat{i}_abs"])) < 1e-4 * float(g[p + f"enc_feat{i}_abs"]), f"enc_feat{i}"
        s = float(dec_feats[i].detach().double().abs().sum())
        assert abs(s - float(g[p + f"dec_feat{i}_abs"])) < 1e-4 * float(g[p + f"dec_feat{i}_abs"]), f"dec_feat{i}"
    close(model.quantizer._codebook.cluster_size, g[p + "cluster_after"], 1e-5, "cluster (=> indices histogram)")
    assert abs(float(model.quantizer._codebook.embed.double().abs().sum()) - float(g[p + "embed_after_abs"])) < 1e-5 * float(g[p + "embed_after_abs"])


from test_oracle_golden import large_close  # noqa: E402  (fixture summaries of large tensors: hashed positions + channel sums)


def test_indices_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "models.npz"))
    for gtag, mtag in GOLDEN_FWD:
        model, cfg, _ = build(mtag)
        B, H, W, seed = [int(v) for v in g[gtag + ".shape"]]
        model.train()
        with torch.no_grad():
            _, _, ind, _ = model.encode(O.det_input(B, H, W, seed).to(DEV))
        flips = check_indices(ind.cpu().numpy(), g[gtag + ".indices"], g[gtag + ".index_gap"], gtag)
        assert flips == 0, f"{gtag}: {flips} near-tie flips"


# Stage-0 bars (round 6): within 10 x of the worst error achieved on MI355X by the quantity's family over all fixtures
# (profiles/r06_parity_margins.txt, checked by tests/test_margins.py) and never above north_star's 1e-4:
#   losses 1.3e-6 -> 1e-5; x_recon 5.1e-6 -> 5e-5; gradients (max error / tensor max) 1.3e-5 -> 1e-4;
#   sigma gradients 1.2e-4 -> 1e-3 (one scalar at the end of a k x k-tap reduction over a whole feature tensor with heavy
#   cancellation: the only family above 1e-4; the oracle itself differs from the reference by 3e-6 there).
# Rounds 1-5 carried 5e-3 / 2e-2 here: three orders above what the kernels deliver.
LOSS_TOL, XRECON_TOL, GRAD_TOL, SIGMA_GRAD_TOL = 1e-5, 5e-5, 1e-4, 1e-3


def _golden_step(g, gtag, mtag, grad_tol=GRAD_TOL):
    from favae_step import TrainStep
    model, cfg, _ = build(mtag)
    B, H, W, seed = [int(v) for v in g[gtag + ".shape"]]
    x = O.det_input(B, H, W, seed).to(DEV)
    ts = TrainStep(model, lr=4.5e-6 * B, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01)
    model.train()
    ts.gflat.zero_()
    out = ts.losses(x)
    out["loss_g"].sum().backward()
    p = gtag + "."
    for k, gk in (("loss_quant", "loss_q"), ("loss_l1", "loss_l1"), ("loss_ffl", "loss_ffl"), ("loss_dsl", "loss_dsl"), ("loss_g", "loss_g")):
        close(out[k].reshape(-1), g[p + gk], LOSS_TOL, gk)
    close(torch.stack([v.reshape(()) for v in out["loss_dsl_levels"]]), g[p + "loss_dsl_levels"], LOSS_TOL, "dsl levels")
    xr = out["x_recon"].detach().cpu()
    close(xr[:, :, ::max(1, H // 8), ::max(1, W // 8)], g[p + "x_recon_slice"], XRECON_TOL, "x_recon")
    # the slice above sits on the corners of the conv kernels' 16 x 16-pixel workgroup tiles; these cover every element: the sum and
    # abs-sum over the whole tensor, and (fixtures of the BASELINE sizes) 32768 hashed positions + per-channel sums / sums of squares
    xd = xr.double()
    assert abs(float(xd.sum()) - float(g[p + "x_recon_sum"])) < 1e-4 * float(g[p + "x_recon_abs"]), "x_recon sum"
    assert abs(float(xd.abs().sum()) - float(g[p + "x_recon_abs"])) < 1e-4 * float(g[p + "x_recon_abs"]), "x_recon abs-sum"
    if p + "x_recon.at" in g.files:
        large_close(g, p + "x_recon", xr, XRECON_TOL, "x_recon")
    named = dict(model.named_parameters())
    n = n_full = 0
    for k, prm in named.items():
        key = p + "g." + k + ".head"
        if key in g.files:
            gr = prm.grad.detach().cpu().contiguous()
            tol = SIGMA_GRAD_TOL if k.endswith("sigmas") else grad_tol
            scale = float(np.abs(g[key]).max()) + 1e-30
            err = float(np.abs(gr.reshape(-1)[:16].numpy() - g[key]).max()) / scale
            margins.record("grad head " + k, err, tol)
            assert err < tol, f"grad head {k}: {err:.3e}"
            ref_abs = float(g[p + "g." + k + ".abs"])
            margins.record("grad abs-sum " + k, abs(float(gr.double().abs().sum()) - ref_abs) / ref_abs, tol)
            assert abs(float(gr.double().abs().sum()) - ref_abs) < tol * ref_abs, f"grad abs-sum {k}"
            # whole tensor (fixtures of the BASELINE sizes): every element of the small ones, hashed positions + channel sums of the large
            if p + "g." + k + ".full" in g.files:
                ref = g[p + "g." + k + ".full"]
                err = float(np.abs(gr.numpy().reshape(ref.shape) - ref).max()) / (float(np.abs(ref).max()) + 1e-30)
                margins.record("grad (every element) " + k, err, tol)
                assert err < tol, f"grad (every element) {k}: {err:.3e}"
                n_full += 1
            elif p + "g." + k + ".at" in g.files:
                large_close(g, p + "g." + k, gr, tol, "grad " + k)
                n_full += 1
            n += 1
    assert n >= 8
    if p + "x_recon.at" in g.files:
        assert n_full >= 8, "the full-size fixtures carry whole-tensor gradients"
    return model, ts, out


def test_train_step_f4_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "models.npz"))
    _golden_step(g, "f4_same_conv_32", "f4_same_conv")


@pytest.mark.parametrize("gtag,mtag", [("cfg1_96", "cfg1"), ("nonpair_conv_80", "nonpair_conv")])
def test_train_step_non_pow2_resolution_against_reference_golden(golden_dir, gtag, mtag):
    """Full step (losses incl. FFL / DSL, gradients) at resolutions that are not powers of two -- 96x96 (6x6 latents) and 80x80
    (5x5 latents, 5-tap blur on a 5-pixel map) -- against the reference's outputs: the focal-frequency loss runs the any-length
    DFT fallback here, the blur its generic tile kernel."""
    g = np.load(os.path.join(golden_dir, "models.npz"))
    _golden_step(g, gtag, mtag)


@pytest.mark.parametrize("gtag,mtag", [("same_resblock_64", "same_resblock"), ("ffl_with_fcm_64", "ffl_with_fcm")])
def test_train_step_variants_against_reference_golden(golden_dir, gtag, mtag):
    """The two remaining FCM / DSL wirings (pair-wise sigmas + residual FCMs; conv FCM + FFL without blur) against the reference."""
    g = np.load(os.path.join(golden_dir, "models_variants.npz"))
    _golden_step(g, gtag, mtag)


def test_train_step_cfg1_256_against_reference_golden(golden_dir):
    """BASELINE config 1 at full size: f=16, codebook 1024, 256x256, batch 2, FFL + DSL on."""
    g = np.load(os.path.join(golden_dir, "cfg1_256.npz"))
    model, ts, out = _golden_step(g, "cfg1_256", "cfg1")
    with torch.no_grad():
        m2, _, _ = build("cfg1")
        m2.train()
        _, _, ind, _ = m2.encode(O.det_input(2, 256, 256, 1234).to(DEV))
    flips = check_indices(ind.cpu().numpy(), g["cfg1_256.indices"], g["cfg1_256.index_gap"])
    assert flips == 0
    close(model.discriminator.features[3].running_mean, g["cfg1_256.bn_running_mean"], 1e-4, "disc BN running mean")
    # optimizer: one fused Adam step over the flat buffers vs torch.optim.Adam in the reference run
    ts.t += 1
    from favae_hip import ops as K
    K.adam_step(ts.pflat, ts.gflat, ts.mflat, ts.vflat, 1, ts.lr, ts.betas, ts.eps, 1.0)
    named = dict(model.named_parameters())
    for k in ("encoder.conv_in.weight", "decoder.final.2.weight", "encoder.sigmas", "decoder.sigmas"):
        got = named[k].detach().cpu().contiguous().reshape(-1)[:16]
        close(got, g[f"cfg1_256.adam.{k}.head"], 1e-6, "adam." + k)


def test_train_step_cfg1_256_with_f44_data_gradients_against_reference_golden(golden_dir):
    """The same reference step with the optional F(4x4, 3x3) kernel on every data gradient that tiles into it (FAVAE_WINO4=1: the
    default of round 5's first half, off since the wide F(2x2) tiling): losses, x_recon and every gradient hold the SAME bars."""
    from favae_hip import ops as K
    prev = K.set_wino4("1")
    try:
        g = np.load(os.path.join(golden_dir, "cfg1_256.npz"))
        _golden_step(g, "cfg1_256", "cfg1")
    finally:
        K.set_wino4(prev)


def test_winograd_tilings_agree_on_the_whole_model():
    """16 x 8 x 128 against 16 x 16 x 64 workgroups of the F(2x2) kernel (FAVAE_WINO_WIDE=1 / 0) on a whole training step at 128 x 128:
    the conv results are bit-identical per layer (tests/test_gpu_ops.py); through the model the per-tile partial sums of the statistics
    epilogues (a finer tile grid, fp64) may move a GroupNorm statistic by an fp32 ulp -- indices identical, reconstruction, losses and
    gradients equal to 1e-5 of their maxima."""
    from favae_hip import ops as K
    from favae_step import TrainStep
    x = O.det_input(2, 128, 128, 77).to(DEV)

    def run(wide):
        model, _, _ = build("cfg1")
        prev = K.set_wino_wide(wide)
        try:
            ts = TrainStep(model, lr=1e-4)
            model.eval()
            with torch.no_grad():
                _, _, ind, _ = model.encode(x)
            model.train()
            ts.gflat.zero_()
            out = ts.losses(x)
            ts.backward(out)
            K.sync_side_stream()
            torch.cuda.synchronize()
            return ind.clone(), out["x_recon"].detach().clone(), float(out["loss_g"]), ts.gflat.clone()
        finally:
            K.set_wino_wide(prev)
    i1, x1, l1, g1 = run(1)
    i0, x0, l0, g0 = run(0)
    assert torch.equal(i1, i0), "codebook indices differ between the two tilings"
    assert float((x1 - x0).abs().max()) <= 1e-5 * float(x0.abs().max())
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    assert float((g1 - g0).abs().max()) <= 1e-5 * float(g0.abs().max()), float((g1 - g0).abs().max()) / float(g0.abs().max())


@pytest.mark.parametrize("gtag,mtag", [("cfg2_256", "cfg2"), ("f4_256", "f4_full")])
def test_train_step_at_baseline_sizes_against_reference_golden(golden_dir, gtag, mtag):
    """The sizes that distinguish the BASELINE configs, one full reference step each (forward, every loss, backward, Adam):
    cfg2_256 = configs[1] wiring at codebook 16384 (256x256, batch 2 of the 32); f4_256 = the configs[3] model at 256x256 (batch 1
    of the 16): f=4, gaussian_kernel 9, num_groups 3, codebook 8192 behind Linear(3,256), 4096 tokens, the two L=4096 / d=512
    AttnBlocks of the mid stages (models/codec.py:87-102), 9-tap blurs on 64^2..256^2 maps."""
    g = np.load(os.path.join(golden_dir, gtag + ".npz"))
    model, ts, out = _golden_step(g, gtag, mtag)
    B, H, W, seed = [int(v) for v in g[gtag + ".shape"]]
    with torch.no_grad():
        m2, _, _ = build(mtag)
        m2.train()
        _, _, ind, _ = m2.encode(O.det_input(B, H, W, seed).to(DEV))
    gap = g[gtag + ".index_gap"]
    flips = check_indices(ind.cpu().numpy(), g[gtag + ".indices"], gap, gtag)
    print(f"\n[{gtag}] tokens {gap.size}, reference gaps < 1e-6: {int((gap < 1e-6).sum())}, flips: {flips}")
    assert flips == 0
    lf = out["logits_fake"].detach().double().abs().sum()
    assert abs(float(lf) - float(g[gtag + ".logits_fake_abs"])) < 1e-4 * float(g[gtag + ".logits_fake_abs"]), "logits_fake"
    close(model.discriminator.features[3].running_mean, g[gtag + ".bn_running_mean"], 1e-4, "disc BN running mean")
    assert abs(float(model.quantizer._codebook.embed.double().abs().sum()) - float(g[gtag + ".embed_after_abs"])) < 1e-5 * float(g[gtag + ".embed_after_abs"])
    close(model.quantizer._codebook.cluster_size, g[gtag + ".cluster_after"], 1e-6, "cluster sizes")
    from favae_hip import ops as K
    ts.t += 1
    nm = ts.n_main
    K.adam_step(ts.pflat[:nm], ts.gflat[:nm], ts.mflat[:nm], ts.vflat[:nm], 1, ts.lr, ts.betas, ts.eps, 1.0)
    if ts.pflat.numel() > nm:
        K.adam_step(ts.pflat[nm:], ts.gflat[nm:], ts.mflat[nm:], ts.vflat[nm:], 1, ts.sigma_lr, ts.betas, ts.eps, 1.0)
    named = dict(model.named_parameters())
    n = 0
    for k in ("encoder.conv_in.weight", "decoder.final.2.weight", "encoder.sigmas", "decoder.sigmas", "sigmas"):
        if f"{gtag}.adam.{k}.head" in g.files:
            close(named[k].detach().cpu().contiguous().reshape(-1)[:16], g[f"{gtag}.adam.{k}.head"], 1e-6, "adam." + k)
            n += 1
    assert n >= 3


def test_full_bench_batch_gives_the_reference_indices_on_its_first_images(golden_dir):
    """Parity evidence AT THE BENCHMARKED SIZE: BASELINE configs[1] runs batch 32; the reference fixture
    `cfg2_256` holds batch 2 of it (a CPU reference step at batch 32 takes minutes and 77 GB).  Every operator in front of the codebook
    lookup is per-image (GroupNorm statistics per image, attention per image; models/vqgan_fcm.py:112-122, models/codec.py:125-314), so a
    batch of 32 whose first two images are the fixture's must reproduce the REFERENCE's indices on those two -- bit for bit -- whatever
    the other thirty are (here: bench.py's randn(seed 1234).clamp(-1, 1) images), in train mode (the lookup precedes the EMA update,
    models/l2_quantize.py:403-438) and in eval mode; and the x_recon of the two images must be the batch-2 run's (decoder: per-image too)."""
    g = np.load(os.path.join(golden_dir, "cfg2_256.npz"))
    B, H, W, seed = [int(v) for v in g["cfg2_256.shape"]]
    gen = torch.Generator().manual_seed(1234)
    x32 = torch.randn(32, 3, H, W, generator=gen).clamp_(-1.0, 1.0)
    x32[:B] = O.det_input(B, H, W, seed)
    x32 = x32.to(DEV)
    gap = g["cfg2_256.index_gap"]
    for mode in ("train", "eval"):
        m, _, _ = build("cfg2")
        m.train() if mode == "train" else m.eval()
        with torch.no_grad():
            _, _, ind32, _ = m.encode(x32)
            m2, _, _ = build("cfg2")
            m2.train() if mode == "train" else m2.eval()
            _, _, ind2, _ = m2.encode(x32[:B].contiguous())
        assert tuple(ind32.shape) == (32, 16, 16) and ind32.dtype == torch.int64
        assert check_indices(ind32[:B].cpu().numpy(), g["cfg2_256.indices"], gap, "batch 32, images 0-1, " + mode) == 0
        assert torch.equal(ind32[:B], ind2), "indices of an image depend on its batch neighbours (%s)" % mode
        assert len(torch.unique(ind32[B:])) > 32            # the other thirty went through the lookup as well
    model, _, _ = build("cfg2")
    model.eval()
    with torch.no_grad():
        xr32 = model(x32, stage=0)[0]
        xr2 = model(x32[:B].contiguous(), stage=0)[0]
    # (not bit for bit: the fp16 operand planes are scaled by a power of two taken from max|x| over the WHOLE tensor, batch neighbours
    # included -- another rounding of the same values; the fixture's own x_recon bar applies)
    close(xr32[:B], xr2, XRECON_TOL, "x_recon of images 0-1: batch 32 vs batch 2")


@pytest.mark.parametrize("mtag,hw", [("nonpair_conv", 64), ("cfg1_k3", 64), ("f4_same_conv", 32)])
def test_side_stream_weight_gradients_are_race_free(mtag, hw):
    """Weight gradients run on a second HIP stream and read dy / x there.  The autograd engine may accumulate IN PLACE into a
    gradient tensor it holds the last reference to (the `g` that an add hands to both of its inputs; a `dres = dy` alias), on the
    main stream, with no ordering against the side stream.  With an artificially slow side stream (a busy-wait in front of every
    side launch) any such write would land before the weight-gradient kernel has read its operand: gradients must stay bit-identical
    to the single-stream run."""
    from favae_hip import ops as K
    from favae_step import TrainStep
    x = O.det_input(2, hw, hw, 31).to(DEV)

    def grads(side_on, delay, fuse=True):
        model, _, _ = build(mtag)
        ts = TrainStep(model, lr=1e-4)
        model.train()
        prev = K._SIDE["on"], K._SIDE["delay"], K._GNBWD_FUSE
        K._SIDE["on"], K._SIDE["delay"], K._GNBWD_FUSE = side_on, delay, fuse
        try:
            ts.gflat.zero_()
            out = ts.losses(x)
            ts.backward(out)
            K.sync_side_stream()
            torch.cuda.synchronize()
        finally:
            K._SIDE["on"], K._SIDE["delay"], K._GNBWD_FUSE = prev
        assert not K._SIDE["pending"]
        return ts.gflat.clone()
    ref = grads(False, 0)
    assert torch.isfinite(ref).all() and float(ref.abs().sum()) > 0
    got = grads(True, 400000)                    # ~0.2 ms in front of each of the ~150 side-stream launches
    assert torch.equal(got, ref), "two-stream gradients differ from the single-stream run: %g" % float((got - ref).abs().max())
    # the direct (non-Winograd) kernels with the streaming GroupNorm-backward pass instead of the data-gradient epilogue: race-free as
    # well (bit-identical between the two-stream and the single-stream run), and the two GroupNorm-backward formulations agree to rounding
    import favae_hip as H
    prev_w = H.query("favae_set_wino", 0)
    try:
        ref1 = grads(False, 0)
        ref2 = grads(False, 0, fuse=False)
        got2 = grads(True, 400000, fuse=False)
    finally:
        H.query("favae_set_wino", prev_w)
    assert torch.equal(got2, ref2), "direct-kernel gradients differ between one and two streams: %g" % float((got2 - ref2).abs().max())
    scale = float(ref2.abs().max())
    assert float((ref1 - ref2).abs().max()) < 2e-5 * scale, float((ref1 - ref2).abs().max()) / scale


@pytest.mark.parametrize("switch", ["_DYCS_FUSE", "_GNBWD_FUSE", "_DEFER_REDUCE", "_DIRECT_GRAD"])
def test_ab_switches_give_the_same_gradients(switch):
    """Every A/B arm of the backward pass (FAVAE_DYCS_FUSE / FAVAE_GNBWD_FUSE / FAVAE_DEFER_REDUCE / FAVAE_DIRECT_GRAD = 0) is a
    different schedule of the same sums: whole-model gradients must agree with the default path to rounding.  (Round 4: FAVAE_DYCS_FUSE=0
    had silently produced garbage gradients in a model while its op-level test stayed green.)"""
    from favae_hip import ops as K
    from favae_step import TrainStep
    if not hasattr(K, switch):
        pytest.skip("no such switch in this build")
    x = O.det_input(2, 64, 64, 31).to(DEV)

    def grads(value):
        model, _, _ = build("cfg1_k3")
        prev = getattr(K, switch)
        setattr(K, switch, value)
        try:
            ts = TrainStep(model, lr=1e-4)
            model.train()
            ts.gflat.zero_()
            out = ts.losses(x)
            ts.backward(out)
            K.sync_side_stream()
            torch.cuda.synchronize()
            return ts.gflat.clone()
        finally:
            setattr(K, switch, prev)
    ref, got = grads(True), grads(False)
    assert torch.isfinite(got).all()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 2e-5 * scale, "%s=0 changes the gradients by %g of the max" % (switch, float((got - ref).abs().max()) / scale)


def test_gan_stage1_discriminator_alone_against_reference_golden(golden_dir):
    """Stage 1 of train() without the chaotic generator step in front of it: the discriminator on (x, the reference's own stage-1
    reconstruction stored in the fixture): logits, hinge_d and every discriminator gradient against the reference, tight."""
    from models.vqgan_fcm import VQGANFCM
    from favae_step import TrainStep
    from losses.hinge import hinge_d_loss
    g = np.load(os.path.join(golden_dir, "gan_128.npz"))
    B, H, W, seed = [int(v) for v in g["gan_128.shape"]]
    mk = dict(codebook_size=512, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True,
              use_l2_quantizer=True, kernel_size=9, dsl_init_sigma=3.0, use_same_conv_gauss=True, num_groups=32)
    cfg = O.OracleConfig(codebook_size=512, variant="same_conv_gauss", kernel_size=9, num_groups=32)
    model = VQGANFCM(**mk, device=DEV)
    model.load_state_dict(O.det_state(cfg, with_disc=True), strict=True)
    model = model.to(DEV).train()
    ts = TrainStep(model, lr=1e-5, train_disc=True)
    ts.dgflat.zero_()
    x = O.det_input(B, H, W, seed).to(DEV)
    xr = torch.from_numpy(g["gan_128.x_recon_d"]).to(DEV)
    logits_real = model.discriminator(x)
    logits_fake = model.discriminator(xr)
    loss_d = hinge_d_loss(logits_real, logits_fake)
    loss_d.backward()
    torch.cuda.synchronize()
    close(logits_real, g["gan_128.logits_real"], 1e-4, "logits_real")
    close(logits_fake, g["gan_128.logits_fake_d"], 1e-4, "logits_fake_d")
    close(loss_d.reshape(-1), g["gan_128.loss_d"], 1e-4, "loss_d")
    named = dict(model.named_parameters())
    n = 0
    for k, p in named.items():
        if "gan_128.dg." + k + ".head" in g.files:
            gr = p.grad.detach().cpu().contiguous()
            ref_abs = float(g["gan_128.dg." + k + ".abs"])
            assert abs(float(gr.double().abs().sum()) - ref_abs) < 2e-3 * ref_abs, f"dgabs.{k}"
            if "gan_128.dgfull." + k in g.files:
                close(gr, g["gan_128.dgfull." + k], 2e-3, "dgfull." + k)
            n += 1
    assert n == 9


# --------------------------------------------------------------------------------------------------------------
# (b) CPU oracle on the same seeded inputs, every decoder/DSL variant, 2 full optimizer steps
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mtag,hw", [("cfg1_k3", 64), ("f4_same_conv", 32), ("same_resblock", 64), ("ffl_with_fcm", 64),
                                     ("nonpair_conv", 128)])
def test_two_train_steps_vs_oracle(mtag, hw):
    from favae_step import TrainStep
    model, cfg, state = build(mtag)
    dsl = 0.01
    ts = TrainStep(model, lr=1e-4, dsl_weight=dsl)
    orc = O.OracleTrainer(cfg, O.StepConfig(lr=1e-4, dsl_weight=dsl, with_disc_forward=True), state)
    for step in range(2):
        x = O.det_input(2, hw, hw, 100 + step)
        ro = orc.step(x)
        out = ts.step(x.to(DEV))
        top2 = ro["out"]["dist"].topk(2, dim=-1).values
        gap = (top2[..., 0] - top2[..., 1]).reshape(ro["out"]["indices"].shape).numpy()
        with torch.no_grad():
            pass
        for k in ("loss_l1", "loss_quant", "loss_ffl", "loss_dsl", "loss_g"):
            close(out[k].reshape(-1), ro[k].reshape(-1), 2e-4 if step else 1e-4, f"step{step}.{k}")
        close(out["x_recon"], ro["out"]["x_recon"], 2e-4 if step else 1e-4, f"step{step}.x_recon")
    # Gradients of the second step (they already depend on the first optimizer update).  Adam divides by sqrt(v): where a
    # gradient is pure rounding noise (e.g. a conv bias in front of a GroupNorm with one channel per group) the *update*
    # is +-lr whatever the kernel, so parameters are compared against Adam's hard bound and gradients per tensor.
    named = dict(model.named_parameters())
    worst_g, worst_p = 0.0, 0.0
    for k in orc.keys:
        go = orc.P[k].grad
        if go is None:
            continue
        if float(go.abs().max()) > 1e-6:
            worst_g = max(worst_g, rel(named[k].grad, go))
        worst_p = max(worst_p, float((named[k].detach().cpu() - orc.P[k].detach()).abs().max()))
    assert worst_g < 2e-3, f"step-2 gradients: worst per-tensor max-rel {worst_g:.3e}"
    assert worst_p <= 2 * 2 * 1e-4 * 1.01, f"parameters drifted by more than Adam's bound: {worst_p:.3e}"
    close(model.quantizer._codebook.embed, orc.P["quantizer._codebook.embed"], 1e-4, "EMA codebook after 2 steps")
    close(model.quantizer._codebook.cluster_size, orc.P["quantizer._codebook.cluster_size"], 1e-5, "cluster_size")


def test_eval_and_inference_surface():
    """encode()/decode()/get_codebook_entry() as used by the stage-2 caller (reference models/txt_cond_transformer.py:136,165)."""
    from models.vqgan_fcm import VQGANFCM
    mk, ok = MODEL_KW["cfg1_k3"]
    cfg = O.OracleConfig(**ok, inference=True)
    state = O.det_state(cfg, with_disc=True)
    model = VQGANFCM(**mk, device=DEV, inference=True)
    model.load_state_dict(state, strict=True)
    model.to(DEV).eval()
    x = O.det_input(1, 64, 64, 9)
    with torch.no_grad():
        zq, lq, ind, ef = model.encode(x.to(DEV))
        xr, df = model.decode(zq)
        ze = model.quantizer.get_codebook_entry(ind.reshape(1, -1), (1, 4, 4, 256))
    P = {k: v.clone() for k, v in state.items()}
    r = O.vqganfcm_forward(P, x, cfg, training=False)
    assert torch.equal(ind.cpu(), r["indices"])
    assert float(lq) == 0.0 and all(d is None for d in df)
    close(xr, r["x_recon"], 1e-4, "x_recon (eval)")
    close(ze, r["z_q"], 1e-6, "get_codebook_entry")
    close(ef[1], r["enc_feats"][1], 1e-4, "unblurred tap under inference")


# --------------------------------------------------------------------------------------------------------------
# (c) discriminator training (BASELINE config 5 wiring, perceptual term off)
# --------------------------------------------------------------------------------------------------------------
def test_discriminator_forward_backward_vs_oracle():
    """PatchGAN forward + full backward (input gradient, conv / BatchNorm parameter gradients, running statistics)."""
    from models.discriminator import Discriminator
    cfg = O.OracleConfig(codebook_size=256)
    P = {k: v.clone() for k, v in O.det_state(cfg, with_disc=True).items() if k.startswith("discriminator.")}
    for k in P:
        if not O.is_buffer(k):
            P[k].requires_grad_(True)
    # seed 19: no LeakyReLU input of this network lies within 1e-5 of zero (a pre-activation of 1e-7 flips sign between any
    # two fp32 implementations and switches that element's slope between 1 and 0.2 -- seen with seed 9)
    x = (2 * O._hash_uniform(2 * 3 * 64 * 64, 19).reshape(2, 3, 64, 64) - 1).float().requires_grad_(True)
    ref = O.discriminator_forward(P, x, True)
    gy = (2 * O._hash_uniform(ref.numel(), 10).reshape(ref.shape) - 1).float()
    (ref * gy).sum().backward()
    d = Discriminator().to(DEV)
    d.load_state_dict({k[len("discriminator."):]: v.detach() for k, v in O.det_state(cfg, with_disc=True).items()
                       if k.startswith("discriminator.")}, strict=True)
    d.train()
    xd = x.detach().to(DEV).requires_grad_(True)
    out = d(xd)
    (out * gy.to(DEV)).sum().backward()
    close(out, ref, 2e-5, "logits")
    close(xd.grad, x.grad, 2e-4, "dx")
    for name, p in d.named_parameters():
        close(p.grad, P["discriminator." + name].grad, 5e-4, "d" + name)
    close(d.features[3].running_mean, P["discriminator.features.3.running_mean"], 1e-5, "running_mean")
    close(d.features[9].running_var, P["discriminator.features.9.running_var"], 1e-5, "running_var")


def _gan_case(tag):
    """model / oracle configuration of a discriminator-training fixture (oracle/gen_golden.py GAN_CASES)"""
    csize = {"gan_128": 512, "cfg5_256": 2048}[tag]
    mk = dict(codebook_size=csize, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True,
              use_l2_quantizer=True, kernel_size=9, dsl_init_sigma=3.0, use_same_conv_gauss=True, num_groups=32)
    return mk, O.OracleConfig(codebook_size=csize, variant="same_conv_gauss", kernel_size=9, num_groups=32)


@pytest.mark.parametrize("tag", ["gan_128", "cfg5_256"])
def test_gan_iteration_against_reference_golden(golden_dir, tag):
    """One full train() iteration with discriminator training against the reference's outputs (tests/golden/gan_128.npz, and
    cfg5_256.npz = BASELINE configs[4] at its own size: 256x256, codebook 2048, use_same_conv_gauss, num_groups 32, k = 9):
    hinge generator term, adaptive weight, stage-1 discriminator update, two EMA codebook updates, three BN updates."""
    from test_oracle_golden import check_gan_golden
    from models.vqgan_fcm import VQGANFCM
    from favae_step import TrainStep
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    B, H, W, seed = [int(v) for v in g[tag + ".shape"]]
    lr, disc_w = [float(v) for v in g[tag + ".hyper"]]
    mk, cfg = _gan_case(tag)
    model = VQGANFCM(**mk, device=DEV)
    model.load_state_dict(O.det_state(cfg, with_disc=True), strict=True)
    model = model.to(DEV)
    ts = TrainStep(model, lr=lr, train_disc=True, disc_weight=disc_w)
    x = O.det_input(B, H, W, seed).to(DEV)
    model.train()
    # gradients must be read before Adam consumes them: run the pieces of TrainStep.step() by hand
    ts.gflat.zero_()
    out = ts.losses(x)
    ts.backward(out)
    named = dict(model.named_parameters())
    res = {"loss_disc": out["loss_disc"].detach().cpu(), "weight_d": float(out["weight_d"]), "loss_g": out["loss_g"].detach().cpu(),
           "logits_fake": out["logits_fake"].detach().cpu(), "x_recon": out["x_recon"].detach().cpu(),
           "grads": {k: p.grad.detach().clone().cpu() for k, p in named.items() if not k.startswith("discriminator.")}}
    ts.t += 1
    nm = ts.n_main
    from favae_hip import ops as K
    K.adam_step(ts.pflat[:nm], ts.gflat[:nm], ts.mflat[:nm], ts.vflat[:nm], ts.t, ts.lr, ts.betas, ts.eps, 1.0)
    if ts.pflat.numel() > nm:
        K.adam_step(ts.pflat[nm:], ts.gflat[nm:], ts.mflat[nm:], ts.vflat[nm:], ts.t, ts.sigma_lr, ts.betas, ts.eps, 1.0)
    # stage 1 by hand as well (gradients before the discriminator's Adam step)
    from losses.hinge import hinge_d_loss
    ts.dgflat.zero_()
    seen = []                                         # the reconstruction stage 1 feeds the discriminator with (post-Adam generator)
    hook = model.decoder.register_forward_hook(lambda m, i, o: seen.append(o[0].detach().cpu().contiguous()))
    logits_real, logits_fake = model(x, stage=1)
    hook.remove()
    # EVERY element of the stage-1 reconstruction against the reference's (x_recon_d is stored in full).  It sits behind the generator's
    # first Adam step (-lr * sign(g): noise-level gradient elements move their parameter by 2 lr in opposite directions in two fp32
    # implementations), the same perturbation the stage-1 logits carry (logits_fake_d: 8e-4 measured, bar 3e-3) -- hence that bar here;
    # the stage-0 reconstruction, in front of any optimizer step, holds 1e-4 on every element in check_gan_golden (x_recon.at / .csum).
    ref_d = torch.from_numpy(g[tag + ".x_recon_d"])
    err_d = float((seen[0] - ref_d).abs().max() / ref_d.abs().max())
    print(f"\n[{tag}] stage-1 x_recon vs reference, every element: max err / max {err_d:.2e}")
    assert err_d < 3e-3, err_d
    loss_d = hinge_d_loss(logits_real, logits_fake)
    loss_d.backward()
    res.update({"loss_d": loss_d.detach().cpu(), "logits_real": logits_real.detach().cpu(), "logits_fake_d": logits_fake.detach().cpu(),
                "dgrads": {k: p.grad.detach().clone().cpu() for k, p in named.items() if k.startswith("discriminator.")}})
    K.adam_step(ts.dpflat, ts.dgflat, ts.dmflat, ts.dvflat, ts.t, ts.lr, ts.betas, ts.eps, 1.0)
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}

    def close_fn(a, b, rtol, name):
        close(a, b, rtol, name)
    # Tolerances.  Stage 0 (before any optimizer step) keeps the BASELINE bars: 1e-4 on losses / logits, 5e-3 on gradients;
    # weight_d is a RATIO OF GRADIENT NORMS at the end of two long backward chains (8e-5 off here, bar 2e-3), and loss_g
    # carries weight_d * disc_weight * loss_disc, about -1.8x its own size (-> 1.5e-4, bar 5e-4).
    # Stage 1 is evaluated on the reconstruction of the POST-ADAM generator: the first Adam step is -lr*sign(g), so noise-level
    # gradient elements legitimately move their parameter by 2*lr in opposite directions in two fp32 implementations; the
    # perturbed x_recon (8e-4 on logits_fake_d) then switches individual LeakyReLU slopes in D.  Aggregates stay within 1e-2,
    # single gradient elements within 1e-1.  The tight check of the discriminator's backward itself (5e-4) is
    # test_discriminator_forward_backward_vs_oracle; logits_real / loss_d (unperturbed input) stay at 1e-4.
    # Generator gradients of stage 0 are g_recon + weight_d * disc_weight * g_disc with the two terms largely cancelling behind the
    # decoder (decoder tensors: <= 1e-3 of the tensor max here; encoder tensors, after the cancellation: 3e-3 .. 6e-3 with either
    # 2x2-phase kernel family, tools/experiments/gan_grad_err.py) -- bar 1e-2; the un-cancelled training step keeps 5e-3 (_golden_step).
    tols = dict(stage0=1e-4, weight_d=2e-3, loss_g=5e-4, grads=1e-2, logits_fake_d=3e-3, dgrad_head=1e-1, dgrad_abs=1e-2, bn=1e-3)
    if tag == "cfg5_256":      # 4x the LeakyReLU units: the slope-switching aggregates take the bars the oracle itself needed against the
        tols.update(dgrad_abs=6e-2, dgrad_head=2e-1)    # reference at this size (oracle/gen_golden.py GAN_CASES); stage 0 unchanged
    check_gan_golden(g, res, P, lr, close_fn=close_fn, tols=tols, tag=tag)


def test_cfg5_256_b1_index_flips_and_loss_deltas(golden_dir):
    """BASELINE configs[4] names bf16: the one-plane bf16 mode (b1) on the cfg5_256 state and input, REPORTED as what decides whether the
    mode is usable -- the number of codebook indices that differ from the reference's (fp32) indices and the loss deltas against the
    reference golden -- next to the fp32-grade mode on the same state (0 flips, 1e-4).  Two codebooks: the fixture's closed-form one
    (similarities nearly degenerate: the worst case) and a trained-like one whose codes ARE encoder outputs of other images (every token
    then has a well separated best code, as after EMA training).  bf16 operands carry 8 significand bits: the encoder output moves by
    ~1e-2 relative, so flips are inherent to the mode in any framework; the bars only catch a broken kernel."""
    from models.vqgan_fcm import VQGANFCM
    from favae_hip import ops as K
    tag = "cfg5_256"
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    B, H, W, seed = [int(v) for v in g[tag + ".shape"]]
    mk, cfg = _gan_case(tag)
    state = O.det_state(cfg, with_disc=True)
    x = O.det_input(B, H, W, seed).to(DEV)

    def encode(mode, codebook=None, xin=x):
        prev = K.set_conv_mode(mode)
        try:
            model = VQGANFCM(**mk, device=DEV)
            model.load_state_dict(state, strict=True)
            model = model.to(DEV).train()
            if codebook is not None:
                model.quantizer._codebook.embed.data.copy_(codebook)
            with torch.no_grad():                    # ONE codebook lookup (train mode: the EMA update runs behind it, as in the golden)
                h = model.encoder(K.to_cl(xin), inference=model.inference)[0]
                zq, ind, loss_q = model.quantizer(h)
                x_recon = model.decode(zq)[0]
            torch.cuda.synchronize()
            return dict(x_recon=x_recon, loss_q=loss_q.reshape(-1), ind=ind.cpu().numpy(), h=h,
                        loss_l1=(xin - x_recon).abs().mean().reshape(-1))
        finally:
            K.set_conv_mode(prev)

    ref_ind = g[tag + ".indices"]
    # b1 = bf16 conv operands AND (round 6) bf16 activation storage on the ResnetBlock chain; b1f = the same operands with fp32 storage
    # (rounds 1-5, FAVAE_BF16_STORAGE=0)
    prev_st = K.set_bf16_storage(False)
    try:
        a, b = encode("h3"), encode("b1")
        K.set_bf16_storage(True)
        bs = encode("b1")
    finally:
        K.set_bf16_storage(prev_st)
    n = ref_ind.size
    flips_h3, flips_b1 = int((a["ind"] != ref_ind).sum()), int((b["ind"] != ref_ind).sum())
    d = {k: abs(float(b[k][0]) - float(g[tag + "." + ("loss_q" if k == "loss_q" else k)][0])) / abs(float(g[tag + "." + k][0]))
         for k in ("loss_l1", "loss_q")}
    print("\ncfg5_256 closed-form codebook: index flips vs reference h3 %d / %d, b1 %d / %d (%.1f %%); b1 loss deltas vs reference %s"
          % (flips_h3, n, flips_b1, n, 100.0 * flips_b1 / n, {k: "%.2e" % v for k, v in d.items()}))
    assert flips_h3 == 0
    # The reference-derived bar: the REFERENCE under torch.autocast("cpu", bfloat16) -- what its accelerate
    # mixed-precision mode does to encode / decode (favae_scripts/train_favae.py:239-240), captured by oracle/gen_golden.py cfg5 --
    # flips ref_flips of the 512 indices against its own fp32 run and moves loss_l1 / loss_q by the stored fractions.  The HIP bf16 mode
    # (bf16 conv operands, fp32 activations and accumulation) must not deviate from the fp32 reference by more than that.
    ref_flips = int(g[tag + ".bf16ref.flips"])
    ref_dl1, ref_dq = float(g[tag + ".bf16ref.loss_l1_delta"]), float(g[tag + ".bf16ref.loss_q_delta"])
    xr32 = a["x_recon"].float()
    rms_b1 = float(((b["x_recon"] - xr32).pow(2).mean() / xr32.pow(2).mean()).sqrt())
    print("reference under bf16 autocast: %d / %d flips, loss_l1 %.2e, loss_q %.2e, x_recon rms-rel %.2e | HIP b1: %d flips, loss_l1 %.2e, "
          "loss_q %.2e, x_recon rms-rel %.2e" % (ref_flips, n, ref_dl1, ref_dq, float(g[tag + ".bf16ref.x_recon_rms_rel"]), flips_b1,
                                                d["loss_l1"], d["loss_q"], rms_b1))
    assert flips_b1 <= ref_flips, "b1 flips more codebook indices than the reference's own bf16 autocast run"
    assert d["loss_l1"] <= max(ref_dl1, 1e-3) and d["loss_q"] <= max(2 * ref_dq, 1e-3)
    assert rms_b1 <= float(g[tag + ".bf16ref.x_recon_rms_rel"])
    # With bf16 STORAGE the build rounds what the reference's autocast rounds (conv outputs kept in bf16, train_favae.py:239-240): it is no
    # longer MORE precise than the reference's bf16 run but a second draw from the same precision class -- one scalar per quantity, so
    # the bar is the class, not the draw: no more index flips than the reference's run, loss deltas within 2 x its deltas, x_recon rms
    # within 1.25 x.  (Statistics from the fp32 accumulators and a single rounding of conv + residual keep it at or below the reference's
    # rounding count per layer.)
    flips_bs = int((bs["ind"] != ref_ind).sum())
    ds = {k: abs(float(bs[k][0]) - float(g[tag + "." + k][0])) / abs(float(g[tag + "." + k][0])) for k in ("loss_l1", "loss_q")}
    rms_bs = float(((bs["x_recon"].float() - xr32).pow(2).mean() / xr32.pow(2).mean()).sqrt())
    print("HIP b1 + bf16 activation storage: %d flips, loss_l1 %.2e, loss_q %.2e, x_recon rms-rel %.2e" % (flips_bs, ds["loss_l1"], ds["loss_q"], rms_bs))
    margins.record("b1+storage flips", flips_bs, ref_flips + 0.5)
    margins.record("b1+storage loss_l1 delta", ds["loss_l1"], 2 * ref_dl1)
    margins.record("b1+storage x_recon rms-rel", rms_bs, 1.25 * float(g[tag + ".bf16ref.x_recon_rms_rel"]))
    assert flips_bs <= ref_flips
    assert ds["loss_l1"] <= 2 * max(ref_dl1, 1e-3) and ds["loss_q"] <= max(4 * ref_dq, 2e-3)
    assert rms_bs <= 1.25 * float(g[tag + ".bf16ref.x_recon_rms_rel"])
    # trained-like codebook: l2-normalised encoder outputs (fp32-grade) of other images, one per code
    codes = []
    with torch.no_grad():
        i = 0
        while sum(c.shape[0] for c in codes) < mk["codebook_size"]:
            hh = encode("h3", xin=O.det_input(2, H, W, 9000 + i).to(DEV))["h"]
            codes.append(torch.nn.functional.normalize(hh.permute(0, 2, 3, 1).reshape(-1, hh.shape[1]), dim=-1))
            i += 1
    cb = torch.cat(codes)[:mk["codebook_size"]].reshape(1, mk["codebook_size"], -1)
    a2, b2 = encode("h3", cb), encode("b1", cb)
    f2 = int((a2["ind"] != b2["ind"]).sum())
    l2 = {k: abs(float(b2[k][0]) - float(a2[k][0])) / abs(float(a2[k][0])) for k in ("loss_l1", "loss_q")}
    rms = float(((b2["x_recon"] - a2["x_recon"]).pow(2).mean() / a2["x_recon"].pow(2).mean()).sqrt())
    print("cfg5_256 trained-like codebook: b1 index flips vs h3 %d / %d (%.1f %%); loss deltas %s; x_recon rms-rel %.2e"
          % (f2, n, 100.0 * f2 / n, {k: "%.2e" % v for k, v in l2.items()}, rms))
    # measured on MI355X (round 4): closed-form codebook 6 / 512 flips (1.2 %), trained-like 7 / 512 (1.4 %), loss_l1 deltas 8e-4 / 4e-3
    assert flips_b1 <= 0.05 * n and f2 <= 0.05 * n and l2["loss_l1"] < 2e-2 and d["loss_l1"] < 2e-2


def test_flat_buffer_write_drops_weight_caches():
    """after a write THROUGH the flat parameter buffer (checkpoint restore, broadcast) the forward and the
    data-gradient convs must not run on the previous step's max|w| / Winograd records while the weight gradients read the new
    weights.  ts.pflat.mul_() moves only the flat buffer's version counter; the forward behind it must equal, bit for bit, the
    forward of a TrainStep that was built on the scaled weights."""
    from favae_hip import ops as K
    from favae_step import TrainStep
    x = O.det_input(1, 64, 64, 11).to(DEV)
    m1, _, _ = build("cfg1_k3")
    ts1 = TrainStep(m1, lr=1e-4)
    ts1.step(x)
    torch.cuda.synchronize()
    state = {k: v.detach().clone() for k, v in m1.state_dict().items()}
    w = m1.decoder.final[2].weight
    assert K._weight_amax(w) is not None
    ts1.pflat.mul_(1.25)                                # every weight changes; no parameter version counter moves
    assert K._weight_amax(w) is None and K._wino_cached(w, 0) is None, "stale cache entries survive a flat-buffer write"
    m1.train()
    with torch.no_grad():
        o1 = ts1.losses(x)
    m2, _, _ = build("cfg1_k3")
    m2.load_state_dict(state)
    with torch.no_grad():
        for p in list(m2.encoder.parameters()) + list(m2.decoder.parameters()) + list(m2.quantizer.parameters()):
            p.mul_(1.25)
    ts2 = TrainStep(m2, lr=1e-4)
    m2.train()
    with torch.no_grad():
        o2 = ts2.losses(x)
    torch.cuda.synchronize()
    assert torch.equal(o1["x_recon"], o2["x_recon"])
    for k in ("loss_l1", "loss_quant", "loss_ffl", "loss_dsl"):
        assert float(o1[k].reshape(-1)[0]) == float(o2[k].reshape(-1)[0]), k


def test_gan_trainstep_runs_two_iterations():
    """TrainStep.step() with train_disc (the path bench/production uses): finite losses, discriminator parameters move."""
    from models.vqgan_fcm import VQGANFCM
    from favae_step import TrainStep
    mk = dict(codebook_size=256, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True,
              use_l2_quantizer=True, kernel_size=9, dsl_init_sigma=3.0, use_same_conv_gauss=True, num_groups=32)
    cfg = O.OracleConfig(codebook_size=256, variant="same_conv_gauss", kernel_size=9, num_groups=32)
    model = VQGANFCM(**mk, device=DEV)
    model.load_state_dict(O.det_state(cfg, with_disc=True), strict=True)
    model = model.to(DEV)
    ts = TrainStep(model, lr=1e-4, train_disc=True)
    w0 = model.discriminator.head.weight.detach().clone()
    for s in range(2):
        out = ts.step(O.det_input(2, 128, 128, 50 + s).to(DEV))
        for k in ("loss_g", "loss_disc", "loss_d", "weight_d"):
            assert torch.isfinite(torch.as_tensor(out[k])).all(), k
    assert float((model.discriminator.head.weight - w0).abs().max()) > 0


def test_train_step_is_deterministic():
    """Two runs of the same two training steps (fresh models, same inputs) give bit-identical parameters, codebook and losses:
    every reduction is ordered (split-K slabs, fp64 block partials, counting-sort segment sums, order-independent maxima) and
    the second HIP stream only changes WHEN the weight gradients are computed, not what they are."""
    from models.vqgan_fcm import VQGANFCM
    from favae_step import TrainStep
    mk, ok = MODEL_KW["cfg1"]
    cfg = O.OracleConfig(**ok)
    state = O.det_state(cfg, with_disc=True)

    def run():
        model = VQGANFCM(**mk, device=DEV)
        model.load_state_dict(state, strict=True)
        model = model.to(DEV)
        ts = TrainStep(model, lr=1e-4)
        losses = []
        for s in range(2):
            out = ts.step(O.det_input(2, 128, 128, 70 + s).to(DEV))
            losses.append(float(out["loss_g"].reshape(-1)[0]))
        torch.cuda.synchronize()
        return ts.pflat.clone(), model.quantizer._codebook.embed.clone(), losses

    p1, e1, l1 = run()
    p2, e2, l2 = run()
    assert l1 == l2
    assert torch.equal(p1, p2), f"{int((p1 != p2).sum())} parameters differ"
    assert torch.equal(e1, e2)


@pytest.mark.parametrize("mode,tol_x,tol_l", [("h1", 3e-2, 2e-2), ("b1", 5e-1, 5e-2)])
def test_mixed_precision_step_tracks_fp32_grade_step(mode, tol_x, tol_l):
    """16-bit mixed-precision modes: ops.set_conv_mode("h1") = conv operands in one scaled fp16 plane, "b1" = in one bf16 plane
    (BASELINE configs[4] "bf16"), fp32 accumulation either way.  Not parity modes: the bar is that one full training step (forward,
    every loss, backward, Adam) stays within 16-bit-operand tolerance of the default fp32-grade step on identical state and input
    (bf16 has three significand bits less than the fp16 plane: on this closed-form state, whose codebook similarities are nearly
    degenerate, many tokens then pick another code, so the reconstruction is only held to a loose rms bar -- measured 0.28 -- while
    every loss stays within 2 %), and that it is not bit-identical to it (the one-plane kernels ran)."""
    from models.vqgan_fcm import VQGANFCM
    from favae_step import TrainStep
    from favae_hip import ops as K
    mk, ok = MODEL_KW["cfg1"]
    state = O.det_state(O.OracleConfig(**ok), with_disc=True)

    def run(mode):
        prev = K.set_conv_mode(mode)
        try:
            model = VQGANFCM(**mk, device=DEV)
            model.load_state_dict(state, strict=True)
            ts = TrainStep(model.to(DEV), lr=1e-4)
            out = ts.step(O.det_input(2, 128, 128, 91).to(DEV))
            torch.cuda.synchronize()
            return {k: out[k].detach().clone() for k in ("x_recon", "loss_l1", "loss_ffl", "loss_dsl", "loss_quant", "loss_g")}
        finally:
            K.set_conv_mode(prev)

    a, b = run("h3"), run(mode)
    assert K.get_conv_mode() == "h3"
    assert not torch.equal(a["x_recon"], b["x_recon"])
    print("\n%s vs h3: x_recon %.3e" % (mode, float((b["x_recon"] - a["x_recon"]).abs().max() / a["x_recon"].abs().max())),
          {k: "%.3e" % float((b[k] - a[k]).abs().max() / a[k].abs().max()) for k in ("loss_l1", "loss_ffl", "loss_dsl", "loss_g")})
    if mode == "h1":
        close(b["x_recon"], a["x_recon"], tol_x, "x_recon %s vs h3" % mode)
    else:
        rms = float(((b["x_recon"] - a["x_recon"]).pow(2).mean() / a["x_recon"].pow(2).mean()).sqrt())
        print("x_recon rms-rel %.3e" % rms)
        assert rms < tol_x, rms
    for k in ("loss_l1", "loss_ffl", "loss_dsl", "loss_g"):
        close(b[k], a[k], tol_l, k + " %s vs h3" % mode)
    assert torch.isfinite(b["loss_quant"]).all()


def test_zero_and_nonfinite_operands():
    """fp16 split scheme edge cases: an all-zero operand (max = 0 -> scale 1) gives exact zeros; a NaN input propagates."""
    from favae_hip import ops as K
    x = torch.zeros(1, 128, 16, 16, device=DEV, requires_grad=True)
    w = (torch.randn(128, 128, 3, 3, device=DEV) * 0.05).requires_grad_(True)
    y = K.fused_conv(x, w, None, None, None, None, K.ConvCfg(3, 3, 1, 1))
    assert float(y.abs().max()) == 0.0
    gx, gw = torch.autograd.grad(y, (x, w), torch.zeros_like(y))
    assert float(gx.abs().max()) == 0.0 and float(gw.abs().max()) == 0.0
    xn = torch.randn(1, 128, 16, 16, device=DEV)
    xn[0, 5, 3, 3] = float("nan")
    yn = K.fused_conv(xn, w.detach(), None, None, None, None, K.ConvCfg(3, 3, 1, 1))
    assert torch.isnan(yn).any()


# --------------------------------------------------------------------------------------------------------------
# (d) LPIPS perceptual term (losses/lpips.py).  VGG16 topology unpinned (torchvision absent), head pinned by
#     tests/golden/lpips_head.npz (test_gpu_ops.py); here: the whole module against the CPU oracle on deterministic weights.
# --------------------------------------------------------------------------------------------------------------
def _lpips_module():
    from losses.lpips import LPIPS
    lp = LPIPS(pretrained=False)
    lp.load_state_dict(O.lpips_det_state(), strict=True)
    return lp.to(DEV).eval()


@pytest.mark.parametrize("hw", [(64, 64), (32, 96)])
def test_lpips_vs_oracle(hw):
    """lpips(x, x_recon): per-image values and the gradient w.r.t. the second argument (train_favae.py:77)."""
    LP = O.lpips_det_state()
    lp = _lpips_module()
    x = O.det_input(2, hw[0], hw[1], 41)
    y = (O.det_input(2, hw[0], hw[1], 42) * 0.8).requires_grad_(True)
    ref = O.lpips_forward(LP, x, y)
    (gref,) = torch.autograd.grad(ref.mean(), y)
    yd = y.detach().to(DEV).requires_grad_(True)
    val = lp(x.to(DEV), yd)
    assert val.shape == (2,)
    close(val, ref, 1e-4, "lpips value")
    (gd,) = torch.autograd.grad(val.mean(), yd)
    close(gd, gref, 1e-3, "d lpips / d x_recon")
    with torch.no_grad():
        assert float(lp(x.to(DEV), x.to(DEV)).abs().max()) == 0.0
    assert all(p.grad is None for p in lp.parameters())


def test_lpips_features_vs_oracle():
    """the five VGG16 taps (pre-activation here, post-ReLU in the reference: relu(ours) == theirs)"""
    LP = O.lpips_det_state()
    lp = _lpips_module()
    x = O.det_input(1, 64, 64, 43)
    ref = O.lpips_vgg_features(LP, O.lpips_scaling(LP, x))
    with torch.no_grad():
        got = lp.net(lp.scaling_layer(x.to(DEV)))
    for k in range(5):
        assert got[k].shape == ref[k].shape
        close(torch.relu(got[k]), ref[k], 1e-4, "relu%d" % (k + 1))


def test_train_step_with_lpips_vs_oracle():
    """One stage-0 step with the perceptual term (loss_recon = L1 + pw * LPIPS, train_favae.py:77-79) against the oracle:
    losses and the gradients that reach the generator."""
    from favae_step import TrainStep
    model, cfg, state = build("cfg1_k3")
    LP = O.lpips_det_state()
    ts = TrainStep(model, lr=1e-4, dsl_weight=0.01, lpips=_lpips_module(), perceptual_weight=1.0)
    orc = O.OracleTrainer(cfg, O.StepConfig(lr=1e-4, dsl_weight=0.01, with_disc_forward=True, perceptual_weight=1.0), state,
                          lpips_state=LP)
    x = O.det_input(2, 64, 64, 100)
    ro = orc.step(x)
    out = ts.step(x.to(DEV))
    for k in ("loss_l1", "loss_perceptual", "loss_recon", "loss_quant", "loss_ffl", "loss_dsl", "loss_g"):
        close(out[k].reshape(-1), ro[k].reshape(-1), 1e-4, k)
    named = dict(model.named_parameters())
    worst = 0.0
    for k in orc.keys:
        go = ro["grads"].get(k)
        if go is None or float(go.abs().max()) <= 1e-6:
            continue
        worst = max(worst, rel(named[k].grad, go))
    assert worst < 2e-3, f"gradients with the perceptual term: worst per-tensor max-rel {worst:.3e}"


def test_gan_lpips_trainstep_runs():
    """BASELINE config 5 wiring: discriminator training + LPIPS + FFL + DSL in one iteration; finite and deterministic."""
    from favae_step import TrainStep
    outs = []
    for _ in range(2):
        model, cfg, state = build("cfg1_k3")
        ts = TrainStep(model, lr=1e-4, train_disc=True, lpips=_lpips_module())
        o = ts.step(O.det_input(2, 64, 64, 7).to(DEV))
        outs.append((o["loss_g"].detach().cpu(), o["loss_d"].detach().cpu(), ts.pflat.detach().cpu().clone()))
        assert torch.isfinite(outs[-1][0]).all() and torch.isfinite(outs[-1][2]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])


def test_train_step_with_fixed_sigma_spectrum_loss():
    """--SL_weight on the plain-FCM model (features un-blurred by the model, blurred with a fixed sigma by the loss,
    losses/vqgan_losses.py:34-50).  DSL is off: with both terms on the reference reverses dec_feats twice and pairs features of
    different shapes (its FFL then raises), so the two are mutually exclusive there as well."""
    from favae_step import TrainStep
    model, cfg, state = build("ffl_with_fcm")
    ts = TrainStep(model, lr=1e-4, dsl_weight=0.0, sl_weight=0.02, gaussian_kernel=5, gaussian_sigma=1.3)
    orc = O.OracleTrainer(cfg, O.StepConfig(lr=1e-4, dsl_weight=0.0, with_disc_forward=True, sl_weight=0.02, gaussian_kernel=5,
                                            gaussian_sigma=1.3), state)
    x = O.det_input(2, 64, 64, 100)
    ro = orc.step(x)
    out = ts.step(x.to(DEV))
    for k in ("loss_l1", "loss_quant", "loss_ffl", "loss_sl", "loss_g"):
        close(out[k].reshape(-1), ro[k].reshape(-1), 1e-4, k)
    for a, b in zip(out["loss_sl_levels"], ro["loss_sl_levels"]):
        close(a.reshape(-1), b.reshape(-1), 1e-4, "SL level")
    named = dict(model.named_parameters())
    worst = 0.0
    for k in orc.keys:
        go = ro["grads"].get(k)
        if go is None or float(go.abs().max()) <= 1e-6:
            continue
        worst = max(worst, rel(named[k].grad, go))
    assert worst < 2e-3, f"gradients with the SL term: worst per-tensor max-rel {worst:.3e}"


def test_backward_across_an_arena_reset_fails_loudly():
    """a graph built before TrainStep.step() / zero_arena_reset() and back-propagated after it would read recycled
    operand-range slots (max|x| = 0 or another tensor's) without any error: it must raise instead."""
    from favae_hip import ops as K
    from favae_step import TrainStep
    model, _, _ = build("cfg1_k3")
    ts = TrainStep(model, lr=1e-4)
    x = O.det_input(1, 64, 64, 3).to(DEV)
    ts.step(x)                                            # arms the arena
    model.train()
    out = ts.losses(x)                                    # graph built in epoch e
    K.zero_arena_reset(DEV)                               # what the next step() would do first
    # two guards: autograd's own version check (the saved slots are views of the arena buffer, and the reset is an in-place zero_() of
    # that buffer) fires in whichever node unpacks its saved tensors first; ops._check_arena_epoch names the cause where a range reaches a
    # backward without having been saved through autograd
    with pytest.raises(RuntimeError, match="zero_arena_reset|modified by an inplace operation"):
        out["loss_g"].sum().backward()
    torch.cuda.synchronize()
    K.reset_side_state()
    ts.step(x)                                            # and the object still trains afterwards
    torch.cuda.synchronize()


@pytest.mark.parametrize("mtag,hw", [("cfg1_k3", 64), ("f4_same_conv", 32)])
def test_late_weight_gradients_of_a_foreign_loop_are_race_free(mtag, hw):
    """A loop that is not TrainStep gets ordinary gradient tensors.  Their weight gradients still run on the second stream: the dense
    conv weights pass through identity nodes created at the start of VQGANFCM.forward, which the engine runs last, so the tensors reach
    AccumulateGrad only after the main stream has waited for the side stream (ops._LateGradFn).  With an artificially slow side stream
    every gradient must be bit-identical to the run that keeps them on the main stream."""
    from favae_hip import ops as K
    x = O.det_input(2, hw, hw, 31).to(DEV)

    def grads(late, delay):
        model, _, _ = build(mtag)
        model.train()
        prev = K._LATE["on"], K._SIDE["delay"]
        K._LATE["on"], K._SIDE["delay"] = late, delay
        try:
            x_recon, loss_q, _, _, enc_feats, dec_feats = model(x, stage=0)
            n_alias = len(K._LATE["map"])
            loss = (x - x_recon).abs().mean() + loss_q.sum() + sum(f.float().square().mean() for f in list(enc_feats) + list(dec_feats))
            loss.backward()
            torch.cuda.synchronize()
        finally:
            K._LATE["on"], K._SIDE["delay"] = prev
        assert not K._SIDE["jobs"]
        named = [(n, p) for n, p in model.named_parameters() if n.startswith(("encoder.", "decoder.", "quantizer.")) and p.requires_grad]
        assert all(p.grad is not None for n, p in named if p.dim() == 4), "every conv weight got its gradient"
        return n_alias, {n: p.grad.detach().clone() for n, p in named if p.grad is not None}
    n0, ref = grads(False, 0)
    n1, got = grads(True, 400000)
    assert n0 == 0 and n1 > 20, (n0, n1)
    assert ref.keys() == got.keys()
    for n in ref:
        assert torch.equal(ref[n], got[n]), "%s: %g" % (n, float((ref[n] - got[n]).abs().max()))


def test_autograd_grad_in_a_foreign_loop_forms_only_the_weight_gradient_it_asks_for():
    """torch.autograd.grad() w.r.t. ONE conv weight in the middle of a user's loop (the adaptive weight of the GAN stage,
    favae_scripts/train_favae.py:32-39, asks for the last layer's; here an early encoder weight, so that the whole conv chain lies on
    the way).  Every conv node on the way sees needs_input_grad = True for its weight (it requires grad); the conv nodes ask the
    engine whether their weight's identity node will run at all and skip the weight gradient otherwise."""
    import favae_hip as H
    from bench import Prof
    from favae_hip import ops as K
    x = O.det_input(2, 64, 64, 31).to(DEV)
    prof = Prof(H)

    def run(late):
        model, _, _ = build("cfg1_k3")               # a training-mode forward moves the EMA codebook: a fresh model per arm
        model.train()
        last = [p for n, p in model.encoder.named_parameters() if p.dim() == 4 and tuple(p.shape[1:]) == (p.shape[0], 3, 3)][0]
        prev = K._LATE["on"]
        K._LATE["on"] = late
        try:
            x_recon = model(x, stage=0)[0]
            loss = (x - x_recon).abs().mean()
            torch.cuda.synchronize()
            prof.start(2)
            g = torch.autograd.grad(loss, last, retain_graph=False)[0]
            torch.cuda.synchronize()
            t = prof.stop()
        finally:
            K._LATE["on"] = prev
        assert all(p.grad is None for p in model.parameters()), "autograd.grad() leaves .grad alone"
        return g, sum(v["launches"] for k, v in t.items() if "wgrad" in k)
    g0, n0 = run(False)
    g1, n1 = run(True)
    assert (g0 - g1).abs().max() <= 1e-6 * g0.abs().max()
    assert n0 > 60 and n1 <= n0 // 3, (n0, n1)     # what remains: the convs outside FusedConvFn (Up / Downsample, attention, conv_in / out)


def test_gan_iteration_of_the_reference_loop_matches_trainstep():
    """the GAN-stage iteration as the reference writes it (favae_scripts/train_favae.py:75-105: hinge generator term, adaptive weight from
    two torch.autograd.grad() calls w.r.t. decoder.final[2].weight with retain_graph, one loss_g.backward()) on the drop-in modules --
    ordinary gradient tensors, weight gradients on the second stream and delivered late -- against TrainStep(train_disc=True), which
    restates the same step with the x_recon gradients reused: same adaptive weight, same generator gradients."""
    from favae_hip import ops as K
    from favae_step import TrainStep
    from focal_frequency_loss import FocalFrequencyLoss as FFL
    from losses.hinge import hinge_g_loss
    from losses.vqgan_losses import recon_ffl_features_loss, recon_ffl_loss
    x = O.det_input(2, 64, 64, 31).to(DEV)
    keep = ("encoder.", "decoder.", "quantizer.")
    # A: TrainStep
    mA, _, _ = build("cfg1_k3")
    ts = TrainStep(mA, lr=1e-4, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01, train_disc=True, disc_weight=0.75)
    mA.train()
    ts.gflat.zero_()
    out = ts.losses(x)
    ts.backward(out)
    K.sync_side_stream()
    torch.cuda.synchronize()
    w_a = float(out["weight_d"])
    g_a = {n: p.grad.detach().clone() for n, p in mA.named_parameters() if n.startswith(keep)}
    # B: the reference's lines, on a second copy of the model
    mB, _, _ = build("cfg1_k3")
    mB.train()
    ffl_func, dsl_func = FFL(loss_weight=1.0, alpha=1.0), FFL(loss_weight=0.01, alpha=1.0)
    x_recon, loss_quant, logits_fake, _, enc_feats, dec_feats = mB(x, stage=0)
    assert len(K._LATE["map"]) > 20, "ordinary parameters: the dense conv weights went through their identity nodes"
    loss_recon = (x - x_recon).abs().mean()
    loss_g = loss_recon + 1.0 * loss_quant
    loss_disc = hinge_g_loss(logits_fake)
    last = mB.decoder.final[2].weight
    grad_disc = torch.autograd.grad(loss_disc, last, retain_graph=True)[0]
    grad_recon = torch.autograd.grad(loss_recon, last, retain_graph=True)[0]
    w_b = torch.clamp(torch.norm(grad_recon) / (torch.norm(grad_disc) + 1e-4), 0.0, 1e4).item()
    assert all(p.grad is None for p in mB.parameters())
    loss_g = loss_g + w_b * 0.75 * loss_disc + recon_ffl_loss(ffl_func, x, x_recon)
    loss_g = loss_g + recon_ffl_features_loss(dsl_func, enc_feats, dec_feats, torch.device(DEV))[0]
    loss_g.sum().backward()
    torch.cuda.synchronize()
    assert abs(w_a - w_b) <= 1e-4 * abs(w_b), (w_a, w_b)
    margins.record("reference GAN loop vs TrainStep: weight_d", abs(w_a - w_b) / abs(w_b), 1e-4)
    worst = 0.0
    for n, p in mB.named_parameters():
        if not n.startswith(keep) or not p.requires_grad:
            continue
        assert p.grad is not None, n
        e = float((p.grad - g_a[n]).abs().max()) / (float(g_a[n].abs().max()) + 1e-30)
        worst = max(worst, e)
        assert e <= 1e-4, (n, e)
    margins.record("reference GAN loop vs TrainStep: generator gradients (worst tensor)", worst, 1e-4)
