#!/usr/bin/env python3
"""The conv kernels of the b1 mode (one bf16 plane, bf16 activation storage) on a ResnetBlock-like chain, per shape: forward + backward
of two blocks (GroupNorm+SiLU+conv3x3 twice, residual) under the library's launch profiler -> average time of every conv kernel.
Arms are process-wide environment switches (read once by the library): run once per arm, e.g.
    python tools/halo_bench.py ; FAVAE_WINO1=0 python tools/halo_bench.py          (data gradient on the direct kernel too)
(FAVAE_HALO_TALL: the tall-tile experiment of commit 0230533, profiles/r06_halo_tall.txt; the shipped library ignores it)
usage: python tools/halo_bench.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H
from favae_hip import ops as K
from bench import Prof

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MODE = os.environ.get("HALO_BENCH_MODE", "b1")
K.set_conv_mode(MODE)
K.set_bf16_storage(MODE == "b1" and os.environ.get("FAVAE_BF16_STORAGE", "1") != "0")
prof = Prof(H)
print("# mode %s, bf16 storage %s, FAVAE_HALO_TALL=%s FAVAE_WINO1=%s, batch %d" %
      (MODE, K.bf16_storage(), os.environ.get("FAVAE_HALO_TALL", "auto"), os.environ.get("FAVAE_WINO1", "1"), B))


def chain(N, C, HW):
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
        return out.detach(), [g.float() for g in grads]
    out, grads = fwd_bwd()
    # fp32 reference of the same chain (torch): rms of the build against it (the precision class, whatever kernel ran)
    with torch.no_grad():
        h = x0
        for blk in range(2):
            a = torch.nn.functional.conv2d(torch.nn.functional.silu(torch.nn.functional.group_norm(h, 32, gs[2 * blk], gb[2 * blk], 1e-5)), ws[2 * blk], bs[2 * blk], padding=1)
            h = h + torch.nn.functional.conv2d(torch.nn.functional.silu(torch.nn.functional.group_norm(a, 32, gs[2 * blk + 1], gb[2 * blk + 1], 1e-5)), ws[2 * blk + 1], bs[2 * blk + 1], padding=1)
        err = float(((out.double() - h.double()).pow(2).mean() / h.double().pow(2).mean()).sqrt())
    fwd_bwd()
    torch.cuda.synchronize()
    prof.start(2)
    for _ in range(3):
        fwd_bwd()
    torch.cuda.synchronize()
    t = prof.stop()
    rows = [(k, v) for k, v in t.items() if k.startswith(("conv3x3_", "conv_wgrad", "gn_bwd_apply"))]
    rows.sort(key=lambda kv: -kv[1]["total_us"])
    print("%4d ch @%3d^2 x %d: out rms vs torch fp32 %.2e" % (C, HW, N, err))
    for k, v in rows:
        tf = v["flops"] / max(v["total_us"], 1e-9) * 1e-6
        print("      %-86s n %3d  avg %8.1f us  %6.0f TFLOP/s" % (k[:86], v["launches"], v["total_us"] / v["launches"], tf))


for C, HW in ((128, 256), (128, 128), (256, 128), (256, 64), (256, 32), (512, 32), (512, 16)):
    chain(B, C, HW)
    torch.cuda.empty_cache()
