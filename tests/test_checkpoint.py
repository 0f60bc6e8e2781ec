"""Checkpoint wire format of the reference (favae_scripts/train_favae.py:366-379, utils.py:108-119): the flat-buffer
TrainStep exports / imports torch.optim.Adam state dicts, so checkpoints move between the two implementations.  Host logic
only (no kernels run): CPU."""
import os
import sys

import torch

import favae_oracle as O

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fa-vae_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

MK = dict(codebook_size=64, n_embed=32, ch=32, ch_mult=(1, 2), attn_resolutions=[], use_cosine_sim=True, use_l2_quantizer=True,
          kernel_size=3, dsl_init_sigma=3.0)


def _model(**extra):
    from models.vqgan_fcm import VQGANFCM
    torch.manual_seed(0)
    try:
        return VQGANFCM(**MK, device=torch.device("cpu"), **extra)
    except TypeError:                                   # `ch` is not a VQGANFCM kwarg in the reference either
        kw = dict(MK)
        kw.pop("ch")
        return VQGANFCM(**kw, device=torch.device("cpu"), **extra)


def _fill(ts, seed):
    g = torch.Generator().manual_seed(seed)
    ts.mflat.copy_(torch.randn(ts.mflat.shape, generator=g))
    ts.vflat.copy_(torch.rand(ts.vflat.shape, generator=g))
    if ts.train_disc:
        ts.dmflat.copy_(torch.randn(ts.dmflat.shape, generator=g))
        ts.dvflat.copy_(torch.rand(ts.dvflat.shape, generator=g))


def test_adam_state_dict_loads_into_torch_optim(tmp_path):
    from favae_step import TrainStep
    from utils import save_model
    for extra, train_disc in ((dict(use_gauss_resblock=True), True), (dict(use_non_pair_conv=True), False)):
        model = _model(**extra)
        ts = TrainStep(model, lr=1e-4, train_disc=train_disc)
        ts.t, ts.t_d = 3, 2                     # opt_d keeps its own step count (it starts later: disc_start_epochs)
        _fill(ts, 1)
        # the optimizers exactly as the reference builds them (train_favae.py:292-305)
        g_params = list(model.encoder.parameters()) + list(model.decoder.parameters()) + list(model.quantizer.parameters())
        if hasattr(model, "sigmas"):
            opt_g = torch.optim.Adam([{"params": g_params}, {"params": model.sigmas, "lr": 2.0e-7}], lr=1e-4, betas=(0.5, 0.9))
        else:
            opt_g = torch.optim.Adam(g_params, lr=1e-4, betas=(0.5, 0.9))
        opt_d = torch.optim.Adam(model.discriminator.parameters(), lr=1e-4, betas=(0.5, 0.9))
        ck = ts.checkpoint(epoch=5, step=0, loss_recon=0.25)
        assert set(ck) == {"model", "opt_g", "opt_d", "epoch", "step", "loss_recon"}
        path = os.path.join(tmp_path, "latest.pt")
        save_model(ck, path)
        save_model(ck, path)                                 # second save goes through the rename-old path
        assert sorted(os.listdir(tmp_path)) == ["latest.pt"]
        ck = torch.load(path, map_location="cpu", weights_only=False)
        opt_g.load_state_dict(ck["opt_g"])
        opt_d.load_state_dict(ck["opt_d"])
        assert len(opt_g.param_groups) == (2 if hasattr(model, "sigmas") else 1)
        off = 0
        for p in ts.params:
            st = opt_g.state[p]
            n = p.numel()
            assert float(st["step"]) == 3.0
            assert torch.equal(st["exp_avg"].reshape(-1), ts.mflat[off:off + n].as_strided(p.shape, p.stride()).reshape(-1))
            assert torch.equal(st["exp_avg_sq"].reshape(-1), ts.vflat[off:off + n].as_strided(p.shape, p.stride()).reshape(-1))
            off += n
        if train_disc:
            assert all(float(opt_d.state[p]["step"]) == 2.0 for p in model.discriminator.parameters())
        else:
            assert len(opt_d.state) == 0
        # and back: a fresh TrainStep resumes from torch's own state dicts
        model2 = _model(**extra)
        model2.load_state_dict(ck["model"], strict=True)
        ts2 = TrainStep(model2, lr=1e-4, train_disc=train_disc)
        ts2.load_opt_state_dicts(opt_g.state_dict(), opt_d.state_dict())
        assert ts2.t == 3 and ts2.t_d == (2 if train_disc else 0)
        assert torch.equal(ts2.mflat, ts.mflat) and torch.equal(ts2.vflat, ts.vflat)
        if train_disc:
            assert torch.equal(ts2.dmflat, ts.dmflat) and torch.equal(ts2.dvflat, ts.dvflat)
        assert torch.equal(ts2.pflat, ts.pflat)


def test_fresh_trainstep_exports_empty_adam_state():
    from favae_step import TrainStep
    model = _model(use_gauss_resblock=True)
    ts = TrainStep(model, lr=1e-4)
    sd = ts.opt_g_state_dict()
    assert sd["state"] == {} and len(sd["param_groups"][0]["params"]) == len(ts.params)
    torch.optim.Adam(ts.params, lr=1e-4, betas=(0.5, 0.9)).load_state_dict(sd)


def test_synthetic_batch_is_the_oracle_input_family():
    """bench.py's workload generator (utils.synthetic_batch, product package) == the oracle's det_input, bit for bit: the GPU leg of
    the bench does not import anything from oracle/."""
    import favae_oracle as O
    from utils import synthetic_batch
    for (B, H, W, seed) in ((2, 16, 16, 1234), (1, 8, 24, 7), (3, 32, 32, 1251)):
        assert torch.equal(synthetic_batch(B, H, W, seed), O.det_input(B, H, W, seed))
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    main_src = src[src.index("def main():"):]
    assert "favae_oracle" not in main_src, "only cpu_baseline() may touch oracle/"
