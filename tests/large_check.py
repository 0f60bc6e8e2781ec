"""Comparison of a large tensor with what tests/golden/blocks_large.npz keeps of the reference's (oracle/gen_golden.py large_summary):
values at O.sample_positions, per-channel sums and sums of squares over every element, the absolute maximum."""
import numpy as np
import torch

import favae_oracle as O


def check_large(g, key, t, tol, name=""):
    t = t.detach().cpu()
    ref_at = torch.from_numpy(g[key + ".at"])
    amax = float(g[key + ".absmax"])
    flat = t.reshape(-1)
    got_at = flat[O.sample_positions(flat.numel(), ref_at.numel())]
    e = float((got_at.double() - ref_at.double()).abs().max())
    assert e <= tol * amax, f"{name or key}: sampled values differ by {e:.3e} > {tol:g} x max {amax:.3e}"
    d = t.double().transpose(0, 1).reshape(t.shape[1], -1)
    per = d.shape[1]
    es = float((d.sum(1) - torch.from_numpy(g[key + ".csum"])).abs().max()) / per        # error of the per-channel MEAN: a wrong tile moves it
    assert es <= tol * amax, f"{name or key}: per-channel mean differs by {es:.3e} > {tol:g} x max {amax:.3e}"
    sq = torch.from_numpy(g[key + ".csq"])
    eq = float(((d.pow(2).sum(1) - sq).abs() / sq.clamp_min(1e-300)).max())
    assert eq <= max(10 * tol, 1e-4), f"{name or key}: per-channel sum of squares differs by {eq:.3e} (relative)"
    assert abs(float(t.abs().max()) - amax) <= 10 * tol * amax, f"{name or key}: absolute maximum"
