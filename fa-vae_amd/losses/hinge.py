"""Hinge GAN losses (reference losses/hinge.py:5-34) on the HIP reduction kernels (favae_hinge_mean*)."""
from favae_hip import ops as K


def hinge_g_loss(logits_fake):
    return K.HingeMeanFn.apply(logits_fake, 0)                     # -mean(logits_fake)


def hinge_d_loss(logits_real, logits_fake):
    loss_real = K.HingeMeanFn.apply(logits_real, 1)                # mean(relu(1 - real))
    loss_fake = K.HingeMeanFn.apply(logits_fake, 2)                # mean(relu(1 + fake))
    return 0.5 * (loss_real + loss_fake)
