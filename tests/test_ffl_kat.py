"""Known-answer tests for the focal-frequency-loss restatement: these need no reference and are
the only pin of that third-party boundary (focal-frequency-loss==0.3.0 is not installable here -> parity unpinned)."""
import math

import torch

import favae_oracle as O


def test_identical_inputs_give_zero_not_nan():
    x = O.det_input(2, 16, 16, 3)
    assert float(O.focal_frequency_loss(x, x.clone(), 1.0)) == 0.0          # NaN -> 0 branch (0/0 weights)


def test_single_pixel_delta():
    H, W, a = 16, 32, 0.75
    t = torch.zeros(1, 1, H, W)
    p = t.clone()
    p[0, 0, 3, 5] = a
    # |F|=a/sqrt(HW) everywhere -> w=1 -> mean(d) = a^2/(HW)
    assert abs(float(O.focal_frequency_loss(p, t, 1.0)) - a * a / (H * W)) < 1e-9


def test_cosine_perturbation():
    H, W, A, k = 32, 32, 0.3, 5
    xx = torch.arange(W, dtype=torch.float64)
    t = torch.zeros(1, 1, H, W, dtype=torch.float64)
    p = t + A * torch.cos(2 * math.pi * k * xx / W).view(1, 1, 1, W)
    # two spectral lines of |F| = A*sqrt(HW)/2, w = 1 on both -> loss = 2*(A^2 HW/4)/(HW) = A^2/2
    assert abs(float(O.focal_frequency_loss(p, t, 1.0)) - A * A / 2) < 1e-12


def test_parseval_bound_and_weight():
    p = O.det_input(2, 32, 32, 1, torch.float64)
    t = O.det_input(2, 32, 32, 2, torch.float64)
    l = float(O.focal_frequency_loss(p, t, 1.0))
    assert 0 < l <= float(((p - t) ** 2).mean()) + 1e-12
    assert abs(float(O.focal_frequency_loss(p, t, 0.01)) - 0.01 * l) < 1e-15


def test_one_fft_analytic_gradient_matches_autograd():
    p = O.det_input(2, 16, 16, 5, torch.float64).requires_grad_(True)
    t = O.det_input(2, 16, 16, 6, torch.float64).requires_grad_(True)
    l = O.focal_frequency_loss(p, t, 0.37)
    l.backward()
    l2, g = O.focal_frequency_loss_grad(p.detach(), t.detach(), 0.37)
    assert abs(float(l) - float(l2)) < 1e-14
    assert float((g - p.grad).abs().max()) < 1e-12
    assert float((g + t.grad).abs().max()) < 1e-12                            # gradient flows to BOTH (App. A.6)
