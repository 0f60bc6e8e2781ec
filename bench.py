#!/usr/bin/env python3
"""FA-VAE training-step benchmark on MI355X (driver contract: `python bench.py --gpus N --steps K --warmup W`).

Workload (BASELINE.json configs[1]): FA-VAE f=16 CelebA-HQ config -- codebook 16384, embed_dim 256, residual FCM +
non pair-wise DSL (gaussian_kernel 9, sigma0 3), FFL 1.0 + DSL 0.01, LPIPS / discriminator training off -- on
synthetic 256x256x3 batches of 32 images per GPU, fp32, random-init weights.  A "step" is one full stage-0 training
step (forward incl. the always-executed discriminator forward, all losses, backward, [RCCL gradient all-reduce],
fused Adam) with the batch already resident in HBM.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU; it starts N ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`, one per GPU, RCCL over xGMI), relays rank 0's JSON line
and exits with the children's status.  Under torchrun (WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement"):
  value      whole-job images/s  (sum over ranks / max-over-ranks time, barrier + synchronize on both sides)
  (the line stays < 4 KB: the kernel table, the other matrix-bound kernels and all prose notes go to the side file
   gpurun_out/bench_detail.json, or $FAVAE_BENCH_DETAIL)
  roofline   the kernel with the largest time share of the timed region among the matrix-bound kernels (conv forward, data
             gradient AND weight gradient families): algorithmic FLOPs (2*M*Cout*KH*KW*Cin per launch) / launch durations, both
             recorded by the library's launch profiler (csrc/prof.hip): two HIP events around every such launch, on the stream the
             kernel is launched on (the weight gradients run on a second stream), inside the timed region;
             peak = 2500 TFLOP/s dense 16-bit MFMA / products per fp32 multiply-add (3 with two fp16 planes, 6 with three bf16)
  kernel_table  (side file) every kernel of two extra (untimed) steps, same profiler at "all launches": per-kernel time, achieved
             GB/s or TFLOP/s
  cpu_baseline  the CPU oracle (kind "port": pure-PyTorch restatement of the reference step, oracle/) timed on this
             host's cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "fa-vae_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)                # the product package only; oracle/ is added by cpu_baseline() alone

PEAK_F32_MFMA_TFLOPS = 157.3          # v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md)
PEAK_16BIT_MFMA_TFLOPS = 2500.0       # dense bf16 / fp16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_TBS = 8.0                    # HBM3E (MI355X_MICROARCH.md)
MFMA_WALL_RANDOM_TFLOPS = 1400.0      # measured: register-resident v_mfma_f32_32x32x16_f16 loop on random operands, 1.34-1.44 PFLOP/s
                                      # over 3 ms .. 1.5 s runs (tools/experiments/mfma_power.hip; 1.84-1.92 on zeros): what the matrix
                                      # pipe sustains on real data
SPLIT_PRODUCTS = {0: None, 1: 1, 2: 3, 3: 6, 4: 1}   # 16-bit MFMA products per multiply-add by scheme id (conv_split.h): h1, h3, b6, b1
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_hbm_traffic.json")
MAX_LINE_BYTES = 4096                 # the driver keeps an 8 KB tail of stdout: the JSON line must stay far below it
NOTES = {
    "roofline": "dominant kernel of the timed region by total time among all matrix-bound launches (conv forward, data-gradient and "
                "weight-gradient kernels; the latter run on a second HIP stream). achieved = algorithmic fp32 FLOPs (2*M*Cout*KH*KW*Cin "
                "per launch) / launch durations from two HIP events around every launch on its own stream (csrc/prof.hip), first "
                "`profiled_steps` steps of the timed region. peak = the hardware peak of the pipe the kernel's matrix instruction runs "
                "on (MI355X_MICROARCH.md: 2500 TFLOP/s dense for v_mfma_f32_32x32x16_f16/bf16, 157.3 for the fp32 MFMA) and frac = "
                "achieved / peak. The split-precision kernels spend `products_per_fma` 16-bit MFMA products per fp32 multiply-add "
                "(planes=2: two scaled fp16 planes, 3 products; planes=3: 6; one plane: 1) and conv3x3_wino_sp_kernel (Winograd "
                "F(2x2,3x3), csrc/conv_wino.h) executes 4/9 of the direct conv's multiplies, conv3x3_wino4_sp_kernel (F(4x4,3x3), "
                "csrc/conv_wino4.h) 1/4: executed_gflop_per_launch = algorithmic x products x (4/9 | 1/4) is the MFMA work really issued, frac_of_pipe_peak = executed / time / 2500. "
                "peak_fp32_equivalent = 2500 / products (833 for h3) and frac_fp32_equivalent = achieved / that: the fraction of what "
                "a direct 3-product kernel could reach at best (a Winograd kernel may exceed its own share of it).",
    "single_stream": "*_single_stream: the same launches in 2 untimed steps with the weight-gradient stream off (exclusive durations)",
    "traffic": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per launch averaged over the launches of one step, separate rocprofv3 "
               "--pmc passes (gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md; fabric-side counter, includes Infinity-Cache "
               "hits); null when the traffic file was collected on a different build of csrc/ (build stamp mismatch)",
    "kernel_table": "every kernel launch of 2 untimed steps (two-stream), sorted by total time; per-launch duration from HIP events on "
                    "the launch stream",
    "with_lpips": "same workload + perceptual_weight * lpips(x, x_recon).mean() every step (train_favae.py:77-79); VGG16 / lin weights "
                  "random-init (vgg16_lpips.pt is not available offline): timing only",
    "roofline_step": "whole step per GPU against SURVEY 8(d)'s per-image work (conv + attention + VQ FLOPs; fused-ideal fp32 bytes)",
}


class stdout_to_stderr:
    """RCCL prints a five-line version banner on fd 1 when a communicator is created; stdout of this program carries ONE JSON line.
    Route fd 1 to fd 2 while a process group is being initialised (OS level: the banner comes from C++)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def rnd(v, digits=5):
    """5 significant digits: keeps the JSON line short without touching `value` / `ms_per_step`"""
    if v is None or not isinstance(v, float) or v != v or v in (float("inf"), float("-inf")):
        return v
    return float("%.*g" % (digits, v))


def build_stamp():
    """sha256 over the kernel sources and the C header: identifies the build a PMC traffic file belongs to"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.h")) +
                    [os.path.join(PKG, "csrc", "Makefile"), os.path.join(ROOT, "include", "favae_hip.h")]):   # Makefile: per-file flags
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def load_traffic():
    """per-launch HBM bytes from the committed PMC passes (profiles/), only when they were collected on THIS build"""
    try:
        t = json.load(open(TRAFFIC_FILE))
    except Exception:
        return {}, {"file": os.path.relpath(TRAFFIC_FILE, ROOT), "status": "missing"}
    stamp, want = t.get("_build"), build_stamp()
    if stamp != want:
        return {}, {"file": os.path.relpath(TRAFFIC_FILE, ROOT), "status": "stale", "file_build": stamp, "this_build": want}
    return t, {"file": os.path.relpath(TRAFFIC_FILE, ROOT), "status": "ok", "build": want}


def compact_roofline(e):
    keep = ("kernel", "launches", "avg_launch_us", "share_of_step_time", "bound", "achieved", "peak", "unit", "frac",
            "executed_gflop_per_launch", "frac_of_pipe_peak", "peak_fp32_equivalent", "frac_fp32_equivalent",
            "algorithmic_bytes_per_launch", "avg_algorithmic_gflop_per_launch", "traffic",
            "traffic_over_algorithmic", "achieved_single_stream", "frac_single_stream", "avg_launch_us_single_stream", "profiled_steps")
    return {k: rnd(e[k]) for k in keep if k in e}


def write_detail(detail):
    """kernel table, the other matrix-bound kernels and the prose notes: a side file, never the stdout line"""
    path = os.environ.get("FAVAE_BENCH_DETAIL")
    if not path:
        d = os.path.join(os.getcwd(), "gpurun_out")
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_detail.json")
        except OSError:
            path = os.path.join(os.getcwd(), "bench_detail.json")
    try:
        with open(path, "w") as f:
            json.dump(detail, f, indent=1)
        return path
    except OSError as e:
        print("[bench] could not write %s: %s" % (path, e), file=sys.stderr)
        return None

def fit_line(res):
    """the stdout JSON line, kept below MAX_LINE_BYTES by dropping optional keys (they stay in the side file) -- never by failing"""
    line = json.dumps(res)
    for k in ("with_lpips", "per_rank_images_per_s", "roofline_step", "ms_per_step_profiler_off", "reference_loop"):
        if len(line) < MAX_LINE_BYTES:
            break
        res = {kk: v for kk, v in res.items() if kk != k}
        print("[bench] line too long: dropped optional key %r (kept in the side file)" % k, file=sys.stderr)
        line = json.dumps(res)
    if len(line) >= MAX_LINE_BYTES and isinstance(res.get("config"), dict):
        res = dict(res, config=dict(res["config"], workload=res["config"]["workload"][:400]))
        line = json.dumps(res)
    return line


# name -> (description, codebook, n_embed, model kwargs, oracle config kwargs, default batch per GPU,
#          algorithmic TFLOP and GB per image of the stage-0 step (None where the survey gives none))
CONFIGS = {
    "celeba_f16": ("BASELINE configs[1]: FA-VAE f=16 CelebA-HQ, codebook %d, embed_dim 256, FCM Res + non-pairwise DSL (k=9)", 16384, 256,
                   dict(ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_gauss_resblock=True),
                   dict(variant="gauss_resblock"), 32, (1.316, 8.1)),
    "imagenet_f4": ("BASELINE configs[3]: FA-VAE f=4 ImageNet, codebook %d, embed_dim 3 -> codebook_dim 256, ch_mult (1,2,4), "
                    "use_same_conv_gauss (num_groups 3, k=9)", 8192, 3,
                    dict(ch_mult=(1, 2, 4), attn_resolutions=[], codebook_dim=256, use_same_conv_gauss=True, num_groups=3),
                    dict(n_embed=3, ch_mult=(1, 2, 4), attn_resolutions=(), codebook_dim=256, variant="same_conv_gauss", num_groups=3), 16,
                    (3.436, 11.3)),
    "ffhq_f16": ("BASELINE configs[4] model: FA-VAE f=16 FFHQ, codebook %d, embed_dim 256, use_same_conv_gauss (num_groups 32, k=9)",
                 2048, 256,
                 dict(ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_same_conv_gauss=True, num_groups=32),
                 dict(variant="same_conv_gauss", num_groups=32), 32, None),
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
    ap.add_argument("--no-comm-diag", action="store_true",
                    help="skip the untimed communication diagnostics of a distributed run (both gradient-exchange arms, 4 extra steps each)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra passes (single-stream rates, all-kernel table, with-LPIPS figure)")
    ap.add_argument("--gan", action="store_true",
                    help="also train the discriminator (BASELINE config-5 wiring: hinge terms, adaptive weight, stage 1; perceptual "
                         "term off) -- not the headline workload, reported under config.workload")
    ap.add_argument("--config", default="celeba_f16", choices=sorted(CONFIGS),
                    help="celeba_f16 = BASELINE configs[1] (the headline metric, default); imagenet_f4 = configs[3]; ffhq_f16 = "
                         "the model of configs[4] (add --gan --lpips for its loss path; fp32 here, not bf16)")
    ap.add_argument("--lpips", action="store_true",
                    help="add the perceptual term lpips(x, x_recon) (train_favae.py:77-79) on random-init VGG16/lin "
                         "weights (vgg16_lpips.pt is not available offline: timing only) -- not the headline workload")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "bf16"],
                    help="fp32 (default, the headline: fp32-grade convs by operand splitting, parity 1e-4) or fp16 = the 16-bit "
                         "mixed-precision mode asked for by BASELINE configs[4] (conv operands rounded to ONE scaled fp16 plane, fp32 "
                         "accumulation, everything else fp32; like the reference under accelerate mixed precision) -- not the "
                         "headline workload, reported as dtype f16 and under config.workload")
    ap.add_argument("--cpu-batch", type=int, default=8, help="images per step of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed steps of the bounded CPU-baseline sample (median reported)")
    args = ap.parse_args()
    desc, cb, n_embed, mk, ok, batch, _ = CONFIGS[args.config]
    args.codebook = args.codebook or cb
    args.batch = args.batch or batch
    return args


# ------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks (this process never initialises a GPU)
# ------------------------------------------------------------------------------------------------------------------
def launch_ranks(n):
    import torch                                    # device_count() does not initialise the GPU on this image
    have = torch.cuda.device_count()
    if have < n:
        print("[bench] --gpus %d but only %d GPU(s) are visible on this node" % (n, have), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:
        if ln.startswith("{"):
            line = ln.rstrip("\n")
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if rc != 0 or line is None:
        print("[bench] the %d-rank launch failed (exit code %s)" % (n, rc), file=sys.stderr)
        return rc or 1
    if len(line) >= MAX_LINE_BYTES:
        line = fit_line(json.loads(line))
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------------------------------------------------
# launch profiler of the library (csrc/prof.hip)
# ------------------------------------------------------------------------------------------------------------------
class Prof:
    """level 1: launches that carry >= 1 GFLOP of algorithmic work (the matrix-bound conv kernels), 2: every launch."""

    def __init__(self, favae_hip):
        self.lib = favae_hip.load()                  # argtypes / restypes come from favae_hip.SIGNATURES

    def start(self, level):
        self.lib.favae_prof_reset()
        self.lib.favae_prof_enable(level)

    def stop(self):
        """-> {kernel name: dict(launches, total_us, min_us, max_us, flops, bytes)} of everything recorded since start()"""
        self.lib.favae_prof_enable(0)
        n = self.lib.favae_prof_report(None, 0)
        buf = ctypes.create_string_buffer(int(n) + 16)
        self.lib.favae_prof_report(buf, len(buf))
        self.lib.favae_prof_reset()
        out = {}
        for ln in buf.value.decode().splitlines():
            name, cnt, tot, mn, mx, fl, by = ln.split("\t")
            out[name] = dict(launches=int(cnt), total_us=float(tot), min_us=float(mn), max_us=float(mx), flops=float(fl),
                             bytes=float(by))
        return out


def kernel_planes(name):
    """operand planes of a split-precision conv kernel from its instantiation name (None: not a split kernel)"""
    if name.startswith(("vq_dist_top2_h3_kernel", "bgemm_sp_kernel")):      # h3 (sp::Scheme<2>) whatever the build mode: codebook scores, SDPA
        return 2
    if not name.endswith(">"):
        return None
    args = [a.strip() for a in name[name.index("<") + 1:-1].split(",")]
    try:
        if name.startswith("conv3x3_halo_sp_kernel") or name.startswith("conv_wgrad_nine_sp_kernel"):
            return int(args[1])
        if name.startswith("conv_fwd_sp_kernel") or name.startswith("conv_wgrad_sp_kernel"):
            return int(args[-1])
        if name.startswith("conv3x3_wino_sp_kernel") and len(args) >= 5:
            return int(args[4])   # <XFORM, GB, SE, WIDE, PLN, storage type>: 2 = h3, 1 = h1, 4 = b1
        if name.startswith(("conv3x3_wino_sp_kernel", "conv3x3_winow_sp_kernel", "conv3x3_wino4_sp_kernel")):
            return 2          # priced on the direct conv's FLOPs (F(2x2) executes 4/9 of the products, F(4x4) 1/4)
    except ValueError:
        pass
    return None


def executed_factor(name, planes):
    """MFMA FLOPs issued per algorithmic FLOP: 16-bit products per fp32 multiply-add x 4/9 for the Winograd F(2x2,3x3) kernels, x 1/4 for F(4x4,3x3)"""
    f = float(SPLIT_PRODUCTS[planes]) if planes else 1.0
    if name.startswith(("conv3x3_wino_sp_kernel", "conv3x3_winow_sp_kernel")):
        f *= 4.0 / 9.0
    if name.startswith("conv3x3_wino4_sp_kernel"):       # F(4x4, 3x3): 36 products per 16 outputs instead of 144
        f *= 1.0 / 4.0
    return f


def roofline_entry(name, r, step_us, traffic, excl=None):
    planes = kernel_planes(name)
    avg_us = r["total_us"] / r["launches"]
    e = {"kernel": name, "launches": r["launches"], "avg_launch_us": avg_us, "share_of_step_time": r["total_us"] / step_us}
    if r["flops"] > 0 and not name.startswith("thin_"):          # thin_*: 3-channel ends, vector-ALU kernels priced against HBM
        ach = 1e-12 * r["flops"] / (1e-6 * r["total_us"])
        peak = PEAK_16BIT_MFMA_TFLOPS if planes else PEAK_F32_MFMA_TFLOPS      # the pipe the kernel's matrix instruction runs on
        ex = executed_factor(name, planes)
        e.update({"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                  "products_per_fma": SPLIT_PRODUCTS[planes] if planes else 1,
                  "executed_gflop_per_launch": 1e-9 * ex * r["flops"] / r["launches"],
                  "frac_of_pipe_peak": ex * ach / peak,
                  "vs_fp32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
                  "frac_of_measured_mfma_wall": (ex * ach / MFMA_WALL_RANDOM_TFLOPS) if planes else None,
                  "avg_algorithmic_gflop_per_launch": 1e-9 * r["flops"] / r["launches"]})
        if planes:
            e["peak_fp32_equivalent"] = PEAK_16BIT_MFMA_TFLOPS / SPLIT_PRODUCTS[planes]
            e["frac_fp32_equivalent"] = ach / e["peak_fp32_equivalent"]
    elif r["bytes"] > 0:
        ach = 1e-9 * r["bytes"] / (1e-6 * r["total_us"])
        e.update({"bound": "hbm", "achieved": ach, "peak": 1e3 * PEAK_HBM_TBS, "unit": "GB/s", "frac": ach / (1e3 * PEAK_HBM_TBS)})
    if r["bytes"] > 0:
        e["algorithmic_bytes_per_launch"] = r["bytes"] / r["launches"]
    t = (traffic.get(name) or traffic.get(name.split("(")[0]) or {}).get("hbm_bytes_per_launch_corrected")
    e["traffic"] = t
    if t and r["bytes"] > 0:
        e["traffic_over_algorithmic"] = t / (r["bytes"] / r["launches"])
    if excl and name in excl and "achieved" in e:
        x = excl[name]
        num = x["flops"] if e["bound"] == "mfma" else x["bytes"]
        e["achieved_single_stream"] = (1e-12 if e["bound"] == "mfma" else 1e-9) * num / (1e-6 * x["total_us"])
        e["frac_single_stream"] = e["achieved_single_stream"] / e["peak"]
        if "peak_fp32_equivalent" in e:
            e["frac_fp32_equivalent_single_stream"] = e["achieved_single_stream"] / e["peak_fp32_equivalent"]
        e["avg_launch_us_single_stream"] = x["total_us"] / x["launches"]
    return e


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
    """The oracle as the checker-side CPU port of the same step (the only place bench.py touches oracle/)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import favae_oracle as O
    ncores = usable_cores(torch)
    torch.set_num_threads(ncores)
    cfg = O.OracleConfig(codebook_size=args.codebook, kernel_size=9, **CONFIGS[args.config][4])
    sc = O.StepConfig(lr=4.5e-6 * args.batch, with_disc_forward=True)
    tr = O.OracleTrainer(cfg, sc)
    B = args.cpu_batch
    t0 = time.perf_counter()
    tr.step(O.det_input(B, args.res, args.res, 1234))      # untimed warm-up step AT THE TIMED BATCH SIZE: thread pool, allocator and the
    t_warm = time.perf_counter() - t0                      # oneDNN primitives of exactly the shapes the timed steps run
    dts = []
    for i in range(max(1, args.cpu_steps)):
        t0 = time.perf_counter()
        tr.step(O.det_input(B, args.res, args.res, 1235 + i))
        dts.append(time.perf_counter() - t0)
    med = sorted(dts)[len(dts) // 2] if len(dts) % 2 else 0.5 * (sorted(dts)[len(dts) // 2 - 1] + sorted(dts)[len(dts) // 2])
    return {"value": rnd(B / med), "unit": "images/s", "cores": ncores, "kind": "port",
            "step_s": [rnd(v, 4) for v in dts],
            "sample": "median of %d timed steps (after 1 warm-up step at the same batch, %.1f s) of the same %s step at batch %d, "
                      "oracle/favae_oracle.py, torch %s CPU, %.1f s timed"
                      % (len(dts), t_warm, args.config, B, torch.__version__, sum(dts))}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # FAVAE_FORCE_DIST=1 exercises the multi-GPU code path (RCCL process group, codebook + gradient all-reduce) at any
    # world size, including 1 (used by tests/test_gpu_dist.py on the single-GPU box)
    use_dist = world > 1 or os.environ.get("FAVAE_FORCE_DIST") == "1"
    # before the first HIP call of the process: 8 hardware queues instead of 4 (streams sharing a queue serialise; with the RCCL process
    # group's streams in the process the weight-gradient stream otherwise lands on the compute stream's queue: +8.6 % step time at every
    # N >= 2, measured at world 1 -- fa-vae_amd/favae_hip/ops.py _side_stream, profiles/r05_dist_overhead.txt)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        with stdout_to_stderr():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))       # RCCL on ROCm
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import favae_hip
    from favae_hip import ops as K
    from favae_step import TrainStep
    from utils import synthetic_batch              # the GPU leg never touches oracle/ (only cpu_baseline() does)
    from models.vqgan_fcm import VQGANFCM

    favae_hip.load()
    if args.precision != "fp32":
        K.set_conv_mode({"fp16": "h1", "bf16": "b1"}[args.precision])
    torch.manual_seed(0)                           # favae_scripts/train_favae.py:235 (ranks are synchronised by TrainStep's broadcast)
    desc, _, n_embed, mk, _, _, work = CONFIGS[args.config]

    # FAVAE_DIST_DEBUG (experiments only): "pg_only" = process group initialised, model / TrainStep as without one; "no_codebook" =
    # distributed TrainStep without the quantizer's codebook all-reduces
    dbg = os.environ.get("FAVAE_DIST_DEBUG", "")

    def build(with_lpips):
        model = VQGANFCM(args.codebook, n_embed, use_cosine_sim=True, use_l2_quantizer=True,
                         sync_codebook=use_dist and dbg not in ("pg_only", "no_codebook"),
                         commitment_weight=1.0, kernel_size=9, dsl_init_sigma=3.0, device=dev, **mk).to(dev)
        lpips = None
        if with_lpips:
            from losses.lpips import LPIPS
            lpips = LPIPS(pretrained=False).to(dev).eval()        # random-init VGG16 / lin weights; .eval(): train_favae.py:308
        lr = 4.5e-6 * args.batch * world               # train_favae.py:250-251
        return TrainStep(model, lr=lr, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01, distributed=use_dist and dbg != "pg_only",
                         train_disc=args.gan, lpips=lpips, perceptual_weight=1.0)
    ts = build(args.lpips)
    exchange_desc = (("RCCL all-reduce, %d segments " % len(ts.exchange.segments)) +
                     ("queued behind backward (FAVAE_COMM_DEFER=0: overlapped with it)" if ts.exchange.defer else "overlapped with backward")
                     if ts.exchange is not None else ("one RCCL all-reduce after backward" if use_dist else "none (1 GPU)"))
    # x = randn(B, 3, 256, 256, generator seed 1234 + rank).clamp(-1, 1), resident on the device; two batches drawn one after
    # the other from that generator alternate over the steps (FAVAE_BENCH_INPUT=hash: the hash-noise batch of rounds 1-5, utils.synthetic_batch)
    if os.environ.get("FAVAE_BENCH_INPUT") == "hash":
        xs = [synthetic_batch(args.batch, args.res, args.res, 1234 + 17 * rank + i).to(dev) for i in range(2)]
    else:
        gen = torch.Generator().manual_seed(1234 + rank)
        xs = [torch.randn(args.batch, 3, args.res, args.res, generator=gen).clamp_(-1.0, 1.0).to(dev) for i in range(2)]
    prof = Prof(favae_hip)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_obj, steps, level, prof_steps=None):
        """`steps` steps between two barrier + synchronize pairs; the launch profiler records the first `prof_steps` of them
        (all by default) -- in the headline region only PROF_STEPS, which keeps its 0.6 % event overhead out of most of `value`"""
        sync()
        prof.start(level)
        t0 = time.perf_counter()
        for i in range(steps):
            if prof_steps is not None and i == prof_steps:
                prof.lib.favae_prof_enable(0)                  # host-side switch: nothing is synchronised
            o = step_obj.step(xs[i % 2])
        sync()
        t = time.perf_counter() - t0
        return t, prof.stop(), o

    # distributed runs: the gradient-exchange arm is MEASURED on the job's first steps (favae_step.CommArmProbe: 9 steps, untimed here,
    # in front of the W warm-up steps), not assumed; the line says which arm the timed region ran and on what numbers
    n_probe = 0
    while ts.comm_probe is not None and ts.comm_probe.active and n_probe < 16:
        ts.step(xs[n_probe % 2])
        n_probe += 1
    if ts.exchange is not None:
        exchange_desc = ("RCCL all-reduce, %d segments, %s (%s)" % (
            len(ts.exchange.segments), "queued behind backward" if ts.exchange.defer else "overlapped with backward",
            (ts.comm_choice or {}).get("how", "")))
    for i in range(args.warmup):
        ts.step(xs[i % 2])
    PROF_STEPS = min(args.steps, 4)
    dt_local, timed_recs, out = timed(ts, args.steps, 1, PROF_STEPS)
    dt = dt_local
    per_rank = [args.batch * args.steps / dt_local]
    if use_dist:
        t = torch.tensor([dt_local], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(allt, t)
        per_rank = [args.batch * args.steps / float(v.item()) for v in allt]
        dt = max(float(v.item()) for v in allt)
    loss = float(out["loss_g"].reshape(-1)[0])
    peak_mem = torch.cuda.max_memory_allocated(dev)

    # ---- communication diagnostics (untimed; every rank; world > 1 or FAVAE_FORCE_DIST=1): BOTH arms of the gradient exchange on
    # this object -- collectives queued behind backward (FAVAE_COMM_DEFER=1) and started where backward finishes each segment (=0) --
    # with the per-segment overlap table of TrainStep.comm_report() and the two codebook all-reduces of the quantizer forward
    # (models/l2_quantize.py:419,427; favae_scripts/train_favae.py:344-347).  Nobody on the builder's side sees the first N > 1 run:
    # its line has to say where the communication time went and which arm is faster.
    comm = None
    if use_dist and ts.exchange is not None and not args.no_comm_diag:
        cb = ts.model.quantizer._codebook
        was_defer, was_timing = ts.exchange.defer, ts.exchange.timing
        comm = {"default_arm": "defer" if was_defer else "eager", "chosen": ts.comm_choice, "steps_per_arm": 3, "arms": {}}
        for arm, defer in (("defer", True), ("eager", False)):
            ts.exchange.defer = defer
            # (i) the arm's step time WITHOUT any instrumentation: 1 warm-up + 3 steps between barrier + synchronize pairs, max over ranks
            ts.exchange.set_timing(False)
            cb.comm_timing = None
            ts.step(xs[0])
            sync()
            t0 = time.perf_counter()
            for i in range(3):
                ts.step(xs[i % 2])
            sync()
            t_arm = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(t_arm, op=dist.ReduceOp.MAX)
            # (ii) two more steps with the events on: the table of the last one (rank 0)
            ts.exchange.set_timing(True)
            for i in range(2):
                cb.comm_timing = []
                ts.step(xs[i % 2])
            sync()
            rep = ts.comm_report() or {"backward_ms": None, "segments": []}
            comm["arms"][arm] = {
                "ms_per_step": rnd(1e3 * float(t_arm.item()) / 3),
                "backward_ms": rnd(rep["backward_ms"]) if rep["backward_ms"] is not None else None,
                # per segment: [MB, start_ms, end_ms, overlapped_ms, exposed_ms] relative to the start of backward (rank 0, last step)
                "segments": [[rnd(r["MB"], 4), rnd(r["start_ms"], 4), rnd(r["end_ms"], 4), rnd(r["overlapped_ms"], 4), rnd(r["exposed_ms"], 4)]
                             for r in rep["segments"]],
                "exposed_ms": rnd(sum(r["exposed_ms"] for r in rep["segments"]), 4),
                # the two SUM all-reduces inside the quantizer forward: [MB, ms] each (cluster-size bins, embedding sums)
                "codebook_allreduce": [[rnd(nb / 1e6, 4), rnd(e0.elapsed_time(e1), 4)] for nb, e0, e1 in (cb.comm_timing or [])[:2]],
            }
        cb.comm_timing = None
        ts.exchange.defer = was_defer
        ts.exchange.set_timing(was_timing)
        a = comm["arms"]
        comm["faster_arm"] = min(a, key=lambda k: a[k]["ms_per_step"])

    # ---- untimed extras (every rank: the steps contain collectives) ------------------------------------------------
    extras = {}
    if not args.no_extras:
        # (a) profiler overhead: the same steps with the profiler off
        t_off, _, _ = timed(ts, 2, 0)
        extras["ms_per_step_profiler_off"] = 1e3 * t_off / 2
        # (b) every launch of two steps (two-stream, as in the timed region)
        t_all, all_recs, _ = timed(ts, 2, 2)
        # (c) the same with the weight-gradient stream off: exclusive kernel durations
        side_on = K._SIDE["on"]
        K._SIDE["on"] = False
        ts.step(xs[0])
        t_excl, excl_recs, _ = timed(ts, 2, 2)
        K._SIDE["on"] = side_on
        extras.update({"all": all_recs, "t_all": t_all, "excl": excl_recs, "t_excl": t_excl})
        # (d) the reference's unconditional perceptual term (train_favae.py:77-79) on top of the headline workload
        if not args.lpips and args.config == "celeba_f16" and not args.gan:
            del ts
            torch.cuda.empty_cache()
            ts_lp = build(True)
            ts_lp.step(xs[0])
            n_lp = min(args.steps, 4)
            t_lp, _, _ = timed(ts_lp, n_lp, 0)
            if use_dist:
                t = torch.tensor([t_lp], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                t_lp = float(t.item())
            extras["with_lpips"] = {"value": rnd(args.batch * world * n_lp / t_lp), "unit": "images/s",
                                    "ms_per_step": rnd(1e3 * t_lp / n_lp), "steps": n_lp}
            del ts_lp
        # (e) the loop a user of the drop-in boundary runs: train_favae.py:68-119 restated on the drop-in modules -- torch DDP
        # (find_unused_parameters=True), torch.optim.Adam(betas=(0.5, 0.9)), the ten-scalar .item() read-back (tools/ref_loop_bench.py).
        # Last of the extras: it needs a process group (RCCL, world 1 when bench.py runs without one), created and destroyed here.
        # World 1 only: at N > 1 nothing here has ever run over RCCL (single-GPU boxes), and an untimed extra must not be able to hang or
        # kill the run that carries the first scaling numbers.
        if args.config == "celeba_f16" and not args.gan and not args.lpips and args.precision == "fp32" and world == 1:
          try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from ref_loop_bench import build_model, reference_loop
            ts = None
            torch.cuda.empty_cache()
            own_pg = not use_dist
            if own_pg:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29549")
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                with stdout_to_stderr():
                    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            torch.manual_seed(0)
            ref_model = build_model(dev, args.codebook, n_embed, sync_codebook=use_dist, **mk)
            n_ref = min(args.steps, 4)
            with stdout_to_stderr():                  # (the first collective may print as well)
                r = reference_loop(ref_model, xs, n_ref, warmup=2, lr=4.5e-6 * args.batch * world, sync_fn=sync)
            t = torch.tensor([r["ms_per_step"]], device=dev, dtype=torch.float64)
            if use_dist:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms_ref = float(t.item())
            extras["reference_loop"] = {"images_per_s": rnd(args.batch * world * 1e3 / ms_ref), "ms_per_step": rnd(ms_ref), "steps": n_ref,
                                        "what": "train_favae.py:68-119 on the drop-in modules: DDP(find_unused_parameters), "
                                                "torch.optim.Adam, 10 x .item() per step (tools/ref_loop_bench.py)"}
            # the same loop in one process without the DDP wrapper (what accelerate builds on one GPU) ...
            r1 = reference_loop(ref_model, xs, n_ref, warmup=1, lr=4.5e-6 * args.batch * world, ddp=False, sync_fn=sync)
            extras["reference_loop"]["one_process"] = {"images_per_s": rnd(args.batch * 1e3 / r1["ms_per_step"]),
                                                       "ms_per_step": rnd(r1["ms_per_step"]), "what": "no DDP wrapper, torch.optim.Adam"}
            # ... and with favae_step.FlatAdam for torch.optim.Adam: the one-line swap
            r2 = reference_loop(ref_model, xs, n_ref, warmup=2, lr=4.5e-6 * args.batch * world, ddp=False, optimizer="flat_direct", sync_fn=sync)
            extras["reference_loop"]["with_flat_adam"] = {"images_per_s": rnd(args.batch * 1e3 / r2["ms_per_step"]),
                                                          "ms_per_step": rnd(r2["ms_per_step"]),
                                                          "what": "no DDP, favae_step.FlatAdam in place of torch.optim.Adam"}
            del ref_model
            if own_pg:
                dist.destroy_process_group()
          except Exception as e:                      # an untimed extra: report, never lose the line
            print("[bench] reference_loop extra failed: %r" % (e,), file=sys.stderr)
            extras["reference_loop"] = {"error": repr(e)[:200]}

    if rank == 0:
        step_us = 1e6 * dt * PROF_STEPS / args.steps           # wall time of the profiled steps (for share_of_step_time)
        res = {
            "metric": "images/sec (256x256, f=16 FA-VAE train step)" if args.config == "celeba_f16" else "images/sec (%dx%d FA-VAE train step, config %s)" % (args.res, args.res, args.config),
            "value": args.batch * world * args.steps / dt,
            "unit": "images/s",
            "n_gpus": world,
            "rccl_world_size": dist.get_world_size() if use_dist else 1,
            "per_rank_images_per_s": [rnd(v) for v in per_rank],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "peak_mem_gib": rnd(peak_mem / 2.0 ** 30, 4),      # torch.cuda.max_memory_allocated over warm-up + timed region, this rank
            "dtype": {"fp32": "f32", "fp16": "f16", "bf16": "bf16"}[args.precision],
            "data": "synthetic",
            "config": {"workload": (desc % args.codebook) + ", FFL 1.0 + DSL 0.01, %dx%d, batch %d/GPU, stage-0 step (discriminator "
                                   "forward on" % (args.res, args.res, args.batch)
                                   + (", discriminator training" if args.gan else "")
                                   + (", LPIPS term on random-init weights" if args.lpips else ", no LPIPS term") + ")"
                                   + {"fp32": "", "fp16": "; MIXED PRECISION h1: conv operands in one scaled fp16 plane, fp32 accumulate",
                                      "bf16": "; MIXED PRECISION b1: conv operands in one bf16 plane, fp32 accumulate, activations of the "
                                              "ResnetBlock chain STORED as bf16 (%s)" % ("on" if K.bf16_storage() else "off: FAVAE_BF16_STORAGE=0")}[args.precision],
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "loss_g_last": rnd(loss, 7),
                       "gradient_exchange": exchange_desc},
        }
        traffic, traffic_meta = load_traffic()
        excl = extras.get("excl")
        ranked = sorted(timed_recs.items(), key=lambda kv: -kv[1]["total_us"])
        detail = {"line": None, "traffic_file": traffic_meta, "notes": NOTES}
        if ranked:
            full = roofline_entry(ranked[0][0], ranked[0][1], step_us, traffic, excl)
            full["profiled_steps"] = PROF_STEPS
            res["roofline"] = compact_roofline(full)
            # where `traffic` comes from: NOT measured in this run (PMC passes cannot share a run with the timing) but read from the
            # committed passes of the same build -- or null when that file belongs to another build of csrc/ 
            res["roofline"]["traffic_source"] = ("%s (own --pmc passes, build %s)" % (traffic_meta["file"], traffic_meta["build"])
                                                 if traffic_meta["status"] == "ok" else "none: %s is %s" % (traffic_meta["file"], traffic_meta["status"]))
            detail["roofline"] = full
            detail["roofline_others"] = [roofline_entry(kn, c, step_us, traffic, excl) for kn, c in ranked[1:]]
        if work is not None and not args.gan and not args.lpips:
            tf, gb = work
            a_f = tf * args.batch * args.steps / dt                                  # per GPU
            a_b = gb * 1e-3 * args.batch * args.steps / dt
            planes = {"fp32": 2, "fp16": 1, "bf16": 4}[args.precision]
            peak_eq = PEAK_16BIT_MFMA_TFLOPS / SPLIT_PRODUCTS[planes]
            res["roofline_step"] = {"tflop_per_image": tf, "gb_per_image": gb,
                                    "mfma": {"achieved": rnd(a_f), "peak": PEAK_16BIT_MFMA_TFLOPS, "unit": "TFLOP/s",
                                             "frac": rnd(a_f / PEAK_16BIT_MFMA_TFLOPS), "peak_fp32_equivalent": rnd(peak_eq),
                                             "frac_fp32_equivalent": rnd(a_f / peak_eq)},
                                    "hbm": {"achieved": rnd(1e3 * a_b), "peak": 1e3 * PEAK_HBM_TBS, "unit": "GB/s",
                                            "frac": rnd(a_b / PEAK_HBM_TBS)}}
        if "all" in extras:
            res["ms_per_step_profiler_off"] = rnd(extras["ms_per_step_profiler_off"])
            su = 1e6 * extras["t_all"]
            tab = sorted(extras["all"].items(), key=lambda kv: -kv[1]["total_us"])[:64]
            detail["kernel_table"] = {"ms_per_step": 1e3 * extras["t_all"] / 2,
                                      "ms_per_step_single_stream": 1e3 * extras["t_excl"] / 2,
                                      "kernels": [roofline_entry(kn, c, su, traffic, excl) for kn, c in tab]}
            res["ms_per_step_single_stream"] = rnd(1e3 * extras["t_excl"] / 2)
        if "with_lpips" in extras:
            res["with_lpips"] = extras["with_lpips"]
        if "reference_loop" in extras:
            res["reference_loop"] = extras["reference_loop"]
            if "images_per_s" in res["reference_loop"]:
                res["reference_loop"]["vs_trainstep"] = rnd(res["reference_loop"]["images_per_s"] / res["value"], 4)
                for k in ("one_process", "with_flat_adam"):
                    if k in res["reference_loop"]:
                        res["reference_loop"][k]["vs_trainstep"] = rnd(res["reference_loop"][k]["images_per_s"] / res["value"], 4)
        if comm is not None:
            res["comm"] = comm
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            print("[bench] GPU part done: %.2f images/s; timing the CPU baseline sample..." % res["value"], file=sys.stderr, flush=True)
            res["cpu_baseline"] = cpu_baseline(args, torch)
        detail["line"] = dict(res)
        path = write_detail(detail)                  # the side file first: nothing of a finished run is lost to a formatting problem
        if path:
            print("[bench] kernel table / notes: %s" % path, file=sys.stderr, flush=True)
        print(fit_line(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
