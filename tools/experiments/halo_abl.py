#!/usr/bin/env python3
"""Timing ablations of conv3x3_halo_sp_kernel in the b1 mode (library built with -DFAVAE_HALO_ABL): forward conv with GroupNorm+SiLU on
load and the statistics epilogue, pieces of the kernel switched off at run time (results are wrong then).  One process per FAVAE_HALO_TALL arm.
The instrumented kernel and the FAVAE_HALO_TALL switch exist at commit 0230533 only (profiles/REJECTED.md, r06_halo_tall.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H
from favae_hip import ops as K
from bench import Prof
dev = torch.device("cuda:0")
K.set_conv_mode("b1")
K.set_bf16_storage(os.environ.get("FAVAE_BF16_STORAGE", "1") != "0")
prof = Prof(H)
NAMES = {0: "full", 1: "no MFMA", 2: "no y stores", 4: "no epilogue", 8: "no halo staging", 16: "no weight staging", 32: "no fragment reads",
         5: "no MFMA, no epilogue", 12: "no epilogue, no halo", 28: "no epilogue/halo/weights", 60: "only MFMA (+barriers)", 61: "only barriers", 64: "return at entry", 128: "return after the prologue", 136: "return after address setup + first loads"}
for C, HW in ((128, 256), (256, 64)):
    torch.manual_seed(0)
    N = 32
    x = torch.randn(N, C, HW, HW, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device=dev) * (1.0 / (3 * C ** 0.5))).contiguous(memory_format=torch.channels_last)
    b = 0.1 * torch.randn(C, device=dev)
    g, be = 1 + 0.2 * torch.randn(C, device=dev), 0.2 * torch.randn(C, device=dev)
    cfg = K.ConvCfg(3, 3, 1, 1, groups=32)
    print("# %d ch @%d^2 x %d, FAVAE_HALO_TALL=%s, bf16 storage %s" % (C, HW, N, os.environ.get("FAVAE_HALO_TALL", "auto"), K.bf16_storage()))
    for abl in [int(v) for v in os.environ.get('ABLS', '0,1,2,4,8,16,32,5,12,28,60,61').split(',')]:
        os.environ["FAVAE_HALO_ABL"] = str(abl)
        with torch.no_grad():
            for it in range(2):
                if it == 1:
                    torch.cuda.synchronize()
                    prof.start(2)
                for _ in range(3):
                    y, skip = K.fused_conv(x, w, b, g, be, None, cfg, pass_input=True)
            torch.cuda.synchronize()
        t = prof.stop()
        for k, v in t.items():
            if k.startswith("conv3x3_halo"):
                print("  abl %2d %-28s %-70s avg %8.1f us" % (abl, NAMES.get(abl, ""), k[:70], v["total_us"] / v["launches"]))
    os.environ["FAVAE_HALO_ABL"] = "0"
