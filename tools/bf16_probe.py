#!/usr/bin/env python3
"""bf16 activation storage against fp32 storage in the b1 conv mode: a ResnetBlock-like chain (GroupNorm+SiLU+conv3x3 twice, residual)
forward + backward at a few shapes -- outputs / gradients relative to the fp32-storage run, and the time of both.
usage: python tools/bf16_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H
from favae_hip import ops as K

dev = torch.device("cuda:0")
K.set_conv_mode("b1")


def run(storage, N, C, HW, reps=3):
    K.set_bf16_storage(storage)
    torch.manual_seed(0)
    x0 = torch.randn(N, C, HW, HW, device=dev).contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn(C, C, 3, 3, device=dev) * (1.0 / (3 * C ** 0.5))).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(4)]
    bs = [(0.1 * torch.randn(C, device=dev)).requires_grad_(True) for _ in range(4)]
    gs = [(1 + 0.2 * torch.randn(C, device=dev)).requires_grad_(True) for _ in range(4)]
    gb = [(0.2 * torch.randn(C, device=dev)).requires_grad_(True) for _ in range(4)]
    gy = torch.randn(N, C, HW, HW, device=dev).contiguous(memory_format=torch.channels_last)
    cfg = K.ConvCfg(3, 3, 1, 1, groups=32)

    def fwd_bwd():
        x = x0.clone().requires_grad_(True)
        h = x
        for blk in range(2):
            a, skip = K.fused_conv(h, ws[2 * blk], bs[2 * blk], gs[2 * blk], gb[2 * blk], None, cfg, pass_input=True)
            h = K.fused_conv(a, ws[2 * blk + 1], bs[2 * blk + 1], gs[2 * blk + 1], gb[2 * blk + 1], skip, cfg)
        out = h.float() if h.dtype != torch.float32 else h
        grads = torch.autograd.grad(out, [x] + ws + gs, gy)
        K.sync_side_stream()
        return out.detach(), [g.float() for g in grads], h.dtype
    out, grads, dt = fwd_bwd()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fwd_bwd()
    e.record(); torch.cuda.synchronize()
    return out, grads, s.elapsed_time(e) / reps, dt


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def rms(a, b):
    return float(((a.double() - b.double()).pow(2).mean() / (b.double().pow(2).mean() + 1e-30)).sqrt())


for N, C, HW in ((2, 128, 64), (8, 128, 256), (8, 256, 64), (8, 512, 16)):
    o0, g0, t0, d0 = run(False, N, C, HW)
    o1, g1, t1, d1 = run(True, N, C, HW)
    print("N %d C %d @%d^2: fp32 storage %.2f ms (%s), bf16 storage %.2f ms (%s) = x%.2f | out rms %.2e max %.2e | dx rms %.2e | dw0 rms %.2e dw3 rms %.2e | dgamma0 rms %.2e"
          % (N, C, HW, t0, d0, t1, d1, t0 / t1, rms(o1, o0), rel(o1, o0), rms(g1[0], g0[0]), rms(g1[1], g0[1]), rms(g1[4], g0[4]), rms(g1[5], g0[5])))
