"""Hinge GAN losses (reference losses/hinge.py:5-34) -- tiny reductions, kept as torch ops on the GPU."""
import torch
import torch.nn.functional as F


def hinge_g_loss(logits_fake):
    return -torch.mean(logits_fake)


def hinge_d_loss(logits_real, logits_fake):
    loss_real = torch.mean(F.relu(1.0 - logits_real))
    loss_fake = torch.mean(F.relu(1.0 + logits_fake))
    return 0.5 * (loss_real + loss_fake)
