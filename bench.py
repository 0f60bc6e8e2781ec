#!/usr/bin/env python3
"""FA-VAE training-step benchmark on MI355X (driver contract: `python bench.py --gpus N --steps K --warmup W`).

Workload (BASELINE.json configs[1]): FA-VAE f=16 CelebA-HQ config -- codebook 16384, embed_dim 256, residual FCM +
non pair-wise DSL (gaussian_kernel 9, sigma0 3), FFL 1.0 + DSL 0.01, LPIPS / discriminator training off -- on
synthetic 256x256x3 batches of 32 images per GPU, fp32, random-init weights.  A "step" is one full stage-0 training
step (forward incl. the always-executed discriminator forward, all losses, backward, [RCCL gradient all-reduce],
fused Adam) with the batch already resident in HBM.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement"):
  value      whole-job images/s  (sum over ranks / max-over-ranks time, barrier + synchronize on both sides)
  roofline   dominant kernel = the conv3x3_halo_sp_kernel<xform, planes, kernel size> instantiation with the largest time share (3x3
             forward / data-gradient conv on the split-precision matrix path): algorithmic FLOPs (2*M*Cout*KH*KW*Cin per
             launch) / launch durations measured with HIP events on the launch stream inside the timed region;
             peak = 2500 TFLOP/s dense 16-bit MFMA / products per fp32 multiply-add (3 with two fp16 planes, 6 with three bf16)
  cpu_baseline  the CPU oracle (kind "port": pure-PyTorch restatement of the reference step, oracle/) timed on this
             host's cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_F32_MFMA_TFLOPS = 157.3          # v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md)
PEAK_16BIT_MFMA_TFLOPS = 2500.0       # dense bf16 / fp16 MFMA (MI355X_MICROARCH.md)
SPLIT_PRODUCTS = {1: 1, 2: 3, 3: 6}        # 16-bit MFMA products issued per fp32 multiply-add: planes -> products (conv_split.h)


# name -> (description, codebook, n_embed, model kwargs, oracle config kwargs, default batch per GPU)
CONFIGS = {
    "celeba_f16": ("BASELINE configs[1]: FA-VAE f=16 CelebA-HQ config, codebook %d, embed_dim 256, residual FCM + non-pairwise DSL "
                   "(k=9, sigma0=3)", 16384, 256,
                   dict(ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_gauss_resblock=True),
                   dict(variant="gauss_resblock"), 32),
    "imagenet_f4": ("BASELINE configs[3]: FA-VAE f=4 ImageNet config, codebook %d, embed_dim 3 projected to codebook_dim 256, "
                    "ch_mult (1,2,4), conv FCM with one sigma per pair (use_same_conv_gauss, num_groups 3, k=9, sigma0=3)", 8192, 3,
                    dict(ch_mult=(1, 2, 4), attn_resolutions=[], codebook_dim=256, use_same_conv_gauss=True, num_groups=3),
                    dict(n_embed=3, ch_mult=(1, 2, 4), attn_resolutions=(), codebook_dim=256, variant="same_conv_gauss", num_groups=3), 16),
    "ffhq_f16": ("model of BASELINE configs[4]: FA-VAE f=16 FFHQ config, codebook %d, embed_dim 256, conv FCM with one sigma per pair "
                 "(use_same_conv_gauss, num_groups 32, k=9, sigma0=3), fp32", 2048, 256,
                 dict(ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_same_conv_gauss=True, num_groups=32),
                 dict(variant="same_conv_gauss", num_groups=32), 32),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the config's, 32 for celeba_f16)")
    ap.add_argument("--codebook", type=int, default=None, help="codebook size (default: the config's)")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gan", action="store_true",
                    help="also train the discriminator (BASELINE config-5 wiring: hinge terms, adaptive weight, stage 1; perceptual "
                         "term off) -- not the headline workload, reported under config.workload")
    ap.add_argument("--config", default="celeba_f16", choices=sorted(CONFIGS),
                    help="celeba_f16 = BASELINE configs[1] (the headline metric, default); imagenet_f4 = configs[3]; ffhq_f16 = "
                         "the model of configs[4] (add --gan --lpips for its loss path; fp32 here, not bf16)")
    ap.add_argument("--lpips", action="store_true",
                    help="add the perceptual term lpips(x, x_recon) (train_favae.py:77-79) on random-init VGG16/lin "
                         "weights (vgg16_lpips.pt is not available offline: timing only) -- not the headline workload")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"],
                    help="fp32 (default, the headline: fp32-grade convs by operand splitting, parity 1e-4) or fp16 = the 16-bit "
                         "mixed-precision mode asked for by BASELINE configs[4] (conv operands rounded to ONE scaled fp16 plane, fp32 "
                         "accumulation, everything else fp32; like the reference under accelerate mixed precision) -- not the "
                         "headline workload, reported as dtype f16 and under config.workload")
    ap.add_argument("--cpu-batch", type=int, default=8, help="images in the bounded CPU-baseline sample")
    args = ap.parse_args()
    desc, cb, n_embed, mk, ok, batch = CONFIGS[args.config]
    args.codebook = args.codebook or cb
    args.batch = args.batch or batch
    return args


class ConvEventHook:
    """Brackets every launch of the implicit-GEMM forward/data-gradient kernel family with HIP events (torch.cuda.Event
    records on the current stream, which is the stream favae_hip launches on).  Launches are keyed by the template
    instantiation they dispatch to, so each key corresponds to exactly one kernel name in a rocprofv3 trace."""

    def __init__(self, torch):
        self.torch = torch
        self.recs = {}          # kernel name -> list of (start, end, flops)
        self.enabled = False

    @staticmethod
    def kernel_name(name, d, has_affine, planes):
        """kernel instantiation a favae_conv_fwd_split call dispatches to (mirrors conv_fwd_impl in csrc/conv.hip)"""
        if name != "favae_conv_fwd_split" or d.Cin % 16 or d.Cout <= 64:
            return None                                    # narrow tiles / fp32-MFMA fallbacks: not the dominant family
        xf = 0 if not has_affine else {0: 1, 1: 2, 2: 3, 3: 3}[d.act]
        halo = (d.KH == 3 and d.KW == 3 and d.stride == 1 and d.pad == 1 and d.gather == 0 and d.Hout == d.Hin and
                d.Wout == d.Win and d.Hin % 8 == 0 and d.Win % 16 == 0)
        if halo and d.lat_step != 2 and d.pad_dw == 0:
            return "conv3x3_halo_sp_kernel<%d, %d, 3>" % (xf, planes)
        halo2 = (d.KH == 2 and d.KW == 2 and d.stride == 1 and d.gather == 0 and d.lat_step == 2 and d.Hout == d.Hin and
                 d.Wout == d.Win and d.Hin % 8 == 0 and d.Win % 16 == 0 and xf == 0 and os.environ.get("FAVAE_CONV_HALO2", "1") != "0")
        if halo2:
            return "conv3x3_halo_sp_kernel<0, %d, 2>" % planes      # 2x2 phase convs (Upsample, Downsample data gradient)
        return "conv_fwd_sp_kernel<%d, %d, true, 8, %d>" % (d.gather, xf, planes)

    def __call__(self, name, args, launch):
        if not self.enabled or name != "favae_conv_fwd_split":
            return launch()
        d = args[0]._obj                                   # (desc, x, wsplit, planes, x_absmax, bias, resid, scale, shift, y)
        kn = self.kernel_name(name, d, args[7] is not None, args[3])
        if kn is None:
            return launch()
        flops = 2.0 * d.N * d.Hout * d.Wout * d.Cout * d.KH * d.KW * d.Cin
        s = self.torch.cuda.Event(enable_timing=True)
        e = self.torch.cuda.Event(enable_timing=True)
        s.record()
        launch()
        e.record()
        self.recs.setdefault(kn, []).append((s, e, flops))

    def summary(self):
        out = {}
        for kn, recs in self.recs.items():
            ms = [s.elapsed_time(e) for s, e, _ in recs]
            fl = [f for _, _, f in recs]
            tot_ms, tot_fl = sum(ms), sum(fl)
            out[kn] = {"launches": len(ms), "avg_us": 1e3 * tot_ms / len(ms), "avg_gflop": 1e-9 * tot_fl / len(ms),
                       "tflops": 1e-12 * tot_fl / (1e-3 * tot_ms), "total_ms": tot_ms}
        return out


def usable_cores(torch):
    """CPU cores this process may really use: min(affinity, cgroup cpu.max quota, torch's default pool size)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, torch.get_num_threads()))


def cpu_baseline(args, torch):
    import favae_oracle as O
    ncores = usable_cores(torch)
    torch.set_num_threads(ncores)
    cfg = O.OracleConfig(codebook_size=args.codebook, kernel_size=9, **CONFIGS[args.config][4])
    sc = O.StepConfig(lr=4.5e-6 * args.batch, with_disc_forward=True)
    tr = O.OracleTrainer(cfg, sc)
    B = args.cpu_batch
    tr.step(O.det_input(1, args.res, args.res, 1234))      # untimed batch-1 step: thread-pool / allocator warm-up
    t0 = time.perf_counter()
    tr.step(O.det_input(B, args.res, args.res, 1235))
    dt = time.perf_counter() - t0
    return {"value": B / dt, "unit": "images/s", "cores": ncores, "kind": "port",
            "sample": f"1 timed training step (after 1 warm-up step) of the same {args.config} config at batch {B}, "
                      f"oracle/favae_oracle.py on torch {torch.__version__} CPU, {dt:.1f} s"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # FAVAE_FORCE_DIST=1 exercises the multi-GPU code path (RCCL process group, codebook + gradient all-reduce) at any
    # world size, including 1 (used by tests/test_gpu_dist.py on the single-GPU box)
    use_dist = world > 1 or os.environ.get("FAVAE_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl")           # RCCL on ROCm
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import favae_hip
    from favae_step import TrainStep
    from utils import synthetic_batch              # the GPU leg never touches oracle/ (only cpu_baseline() below does)
    from models.vqgan_fcm import VQGANFCM

    favae_hip.load()
    if args.precision == "fp16":
        from favae_hip import ops as _Kp
        _Kp.set_conv_mode("h1")
    torch.manual_seed(0)                           # favae_scripts/train_favae.py:235
    desc, _, n_embed, mk, _, _ = CONFIGS[args.config]
    model = VQGANFCM(args.codebook, n_embed, use_cosine_sim=True, use_l2_quantizer=True, sync_codebook=use_dist,
                     commitment_weight=1.0, kernel_size=9, dsl_init_sigma=3.0, device=dev, **mk).to(dev)
    lr = 4.5e-6 * args.batch * world               # train_favae.py:250-251
    lpips = None
    if args.lpips:
        from losses.lpips import LPIPS
        lpips = LPIPS(pretrained=False)            # random-init VGG16 / lin weights (vgg16_lpips.pt is not available offline)
        lpips = lpips.to(dev).eval()               # train_favae.py:308
    ts = TrainStep(model, lr=lr, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01, distributed=use_dist, train_disc=args.gan,
                   lpips=lpips, perceptual_weight=1.0)
    xs = [synthetic_batch(args.batch, args.res, args.res, 1234 + 17 * rank + i).to(dev) for i in range(2)]

    hook = ConvEventHook(torch)
    favae_hip.set_call_hook(hook)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        ts.step(xs[i % 2])
    sync()
    hook.enabled = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = ts.step(xs[i % 2])
    sync()
    dt = time.perf_counter() - t0
    hook.enabled = False
    # Untimed extra: the same launches with the second HIP stream off.  In the timed region the weight-gradient kernels run
    # concurrently with the data-gradient kernels of the next layer, so the per-launch durations of the latter include time
    # in which they share the CUs; the stand-alone rate of the kernel is reported next to the in-step figure.
    from favae_hip import ops as _K                        # (every rank: the steps contain the gradient all-reduce)
    side_on = _K._SIDE["on"]
    _K._SIDE["on"] = False
    timed_recs, hook.recs = hook.recs, {}
    hook.enabled = True
    for i in range(2):
        ts.step(xs[i % 2])
    sync()
    hook.enabled = False
    excl = hook.summary()
    hook.recs = timed_recs
    _K._SIDE["on"] = side_on
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out["loss_g"].reshape(-1)[0])

    if rank == 0:
        conv = hook.summary()
        res = {
            "metric": "images/sec (256x256, f=16 FA-VAE train step)" if args.config == "celeba_f16" else "images/sec (%dx%d FA-VAE train step, config %s)" % (args.res, args.res, args.config),
            "value": args.batch * world * args.steps / dt,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f16",
            "data": "synthetic",
            "config": {"workload": (desc % args.codebook) + ", FFL 1.0 + DSL 0.01, %dx%d, batch %d per GPU, stage-0 step "
                                   "(LPIPS/disc training off, disc forward on)" % (args.res, args.res, args.batch)
                                   + (" + discriminator training (hinge, adaptive weight, stage 1)" if args.gan else "")
                                   + (" + LPIPS perceptual term (random-init weights: timing only)" if args.lpips else "")
                                   + (" -- MIXED PRECISION: conv operands in one scaled fp16 plane (FAVAE_CONV_MODE=h1), fp32 "
                                      "accumulation; not fp32-grade, not the headline" if args.precision == "fp16" else ""),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "loss_g_last": loss},
        }
        if conv:
            def entry(kn, c):
                planes = int(kn.rstrip(">").split(",")[1 if kn.startswith("conv3x3_halo") else -1])
                peak = PEAK_16BIT_MFMA_TFLOPS / SPLIT_PRODUCTS[planes]
                e = excl.get(kn)
                return {"bound": "mfma", "achieved": c["tflops"], "peak": peak, "unit": "TFLOP/s",
                        "frac": c["tflops"] / peak, "vs_fp32_mfma_peak": c["tflops"] / PEAK_F32_MFMA_TFLOPS,
                        "achieved_single_stream": e["tflops"] if e else None,
                        "frac_single_stream": e["tflops"] / peak if e else None,
                        "traffic": (traffic.get(kn) or {}).get("hbm_bytes_per_launch_corrected"), "kernel": kn,
                        "launches": c["launches"], "avg_launch_us": c["avg_us"],
                        "avg_algorithmic_gflop_per_launch": c["avg_gflop"], "share_of_step_time": c["total_ms"] / (1e3 * dt)}
            traffic = {}
            try:                                            # per-launch HBM bytes from the committed PMC passes (profiles/)
                traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
            except Exception:
                pass
            ranked = sorted(conv.items(), key=lambda kv: -kv[1]["total_ms"])
            res["roofline"] = entry(*ranked[0])            # dominant instantiation (largest share of the timed region)
            res["roofline"]["note"] = ("fp32 3x3 conv on the 16-bit matrix pipe by operand splitting (conv_split.h): planes=2 -> two "
                                       "scaled fp16 planes, 3 v_mfma_f32_32x32x16_f16 products per fp32 multiply-add, peak = 2500/3 "
                                       "TFLOP/s; planes=3 -> three bf16 planes, 6 products, peak = 2500/6; achieved = algorithmic "
                                       "fp32 FLOPs / launch time; template args = <fused input transform (0 plain: data gradients "
                                       "and un-normalised convs, 2 GroupNorm+SiLU), planes, kernel size (3: the 3x3 convs, 2: 2x2 phase convs)>")
            res["roofline"]["stream_note"] = ("achieved/frac: launch durations inside the timed region, where weight-gradient kernels "
                                              "run concurrently on a second HIP stream (data-gradient launches share the CUs with "
                                              "them); *_single_stream: the same launches in 2 untimed steps with that stream off")
            res["roofline"]["traffic_note"] = ("bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 averaged over the launches of one "
                                               "step, separate rocprofv3 --pmc passes (gfx950 FETCH_SIZE x2 correction of "
                                               "MI355X_MICROARCH.md; fabric-side counter, includes Infinity-Cache hits)")
            res["roofline_others"] = [entry(kn, c) for kn, c in ranked[1:]]
        if world == 1 and not args.no_cpu_baseline:
            favae_hip.set_call_hook(None)
            print("[bench] GPU part done: %.2f images/s; timing the CPU baseline sample..." % res["value"], file=sys.stderr, flush=True)
            res["cpu_baseline"] = cpu_baseline(args, torch)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
