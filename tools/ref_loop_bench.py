#!/usr/bin/env python3
"""What a user of the drop-in boundary gets: the reference's own training loop, restated, on the drop-in modules.

`bench.py`'s headline number is `favae_step.TrainStep` (flat parameter / gradient / Adam buffers, one fused Adam launch, one grouped
refresh of the per-weight caches, no host synchronisation inside a step).  `north_star` says `train_favae.py` drops in unchanged; the
loop that script would run is `favae_scripts/train_favae.py:68-119`:

    opt_g.zero_grad(); model(x, stage=0) under DDP(find_unused_parameters=True); loss_l1 = (x - x_recon).abs().mean();
    recon_ffl_loss / recon_ffl_features_loss; accelerator.backward(loss_g); torch.optim.Adam(betas=(0.5, 0.9)).step();
    ten scalars -> torch.tensor([...]) -> .item() each                                           (train_favae.py:72-119, 292-305)

This file restates exactly that on `models.vqgan_fcm.VQGANFCM`, `losses.vqgan_losses`, `focal_frequency_loss.FocalFrequencyLoss` of
this package (the same imports the script makes) and times it.  The perceptual term is left out as in the headline workload
(`vgg16_lpips.pt` cannot be shipped; `bench.py` reports it separately as `with_lpips`), the discriminator is not trained
(`disc_start_epochs` > epoch: train_favae.py:82-84) -- its forward still runs inside `model(x, stage=0)`.

`reference_loop(...)` is called by `bench.py` in its untimed extras (`reference_loop` in the JSON line); run as a script it prints the
itemisation: the loop as the reference writes it, then with one piece at a time replaced by what TrainStep does instead.
usage: python tools/ref_loop_bench.py [--batch 32] [--steps 6] [--no-ddp]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fa-vae_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)


def build_model(dev, codebook=16384, n_embed=256, sync_codebook=False, **mk):
    from models.vqgan_fcm import VQGANFCM
    if not mk:
        mk = dict(ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_gauss_resblock=True)
    return VQGANFCM(codebook, n_embed, use_cosine_sim=True, use_l2_quantizer=True, sync_codebook=sync_codebook,
                    commitment_weight=1.0, kernel_size=9, dsl_init_sigma=3.0, device=dev, **mk).to(dev)


def reference_loop(model, xs, steps, warmup=2, lr=4.5e-6 * 32, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01, ddp=True,
                   item_sync=True, optimizer="adam", sync_fn=None):
    """Times `steps` iterations of train_favae.py:68-119 on `model` (a VQGANFCM of this package) over the device batches `xs`.
    ddp: wrap in torch DDP(find_unused_parameters=True) as accelerate.prepare does (train_favae.py:28,239-240,344); needs an
    initialised process group.  item_sync: the ten-scalar host read-back of train_favae.py:118-119.  optimizer: "adam" =
    torch.optim.Adam as the script builds it (:297-305); "flat" = favae_step.FlatAdam, the one-line swap (flat buffers, ONE Adam launch
    per group; safe under DDP); "flat_direct" = FlatAdam(direct_grads=True): the backward kernels accumulate straight into the flat
    gradient buffer as in TrainStep (not under DDP).
    Returns {"ms_per_step", "images_per_s", "loss_g"}."""
    import torch
    from focal_frequency_loss import FocalFrequencyLoss as FFL
    from losses.vqgan_losses import recon_ffl_features_loss, recon_ffl_loss

    dev = xs[0].device
    net = model
    if ddp:
        from torch.nn.parallel import DistributedDataParallel as DDP
        net = DDP(model, device_ids=[dev.index], find_unused_parameters=True)          # train_favae.py:28, accelerate's DDP kwargs
    inner = net.module if ddp else net
    g_params = list(inner.encoder.parameters()) + list(inner.decoder.parameters()) + list(inner.quantizer.parameters())
    if optimizer == "adam":
        make = torch.optim.Adam
    else:
        if ddp and optimizer == "flat_direct":
            raise ValueError("direct accumulation bypasses the AccumulateGrad hooks DDP's reducer is driven by")
        from favae_step import FlatAdam
        make = (lambda g, **kw: FlatAdam(g, direct_grads=optimizer == "flat_direct", **kw))
    if hasattr(inner, "sigmas"):
        opt_g = make([{"params": g_params}, {"params": inner.sigmas, "lr": 2.0e-7}], lr=lr, betas=(0.5, 0.9))
    else:
        opt_g = make(g_params, lr=lr, betas=(0.5, 0.9))                                # train_favae.py:297-302
    ffl_func = FFL(loss_weight=ffl_weight, alpha=1.0)                                  # train_favae.py:313
    dsl_feature_func = FFL(loss_weight=dsl_weight, alpha=1.0)                          # train_favae.py:318
    net.train()
    zero = torch.zeros(1, device=dev)
    out = {}

    def iteration(x):
        opt_g.zero_grad()
        x_recon, loss_quant, logits_fake, _, enc_feats, dec_feats = net(x, stage=0)    # :75
        loss_l1 = (x - x_recon).abs().mean()                                           # :76
        loss_perceptual = zero                                                         # :77 left out (see the module docstring)
        loss_recon = loss_l1 + 0.0 * loss_perceptual
        loss_g = loss_recon + codebook_weight * loss_quant                             # :80
        loss_disc = torch.tensor(0.).to(dev)                                           # :83 (epoch < disc_start_epochs)
        loss_ffl = recon_ffl_loss(ffl_func, x, x_recon)                                # :96
        loss_g = loss_g + loss_ffl
        loss_dsl_features, _ = recon_ffl_features_loss(dsl_feature_func, enc_feats, dec_feats, dev)   # :99
        loss_g = loss_g + loss_dsl_features
        loss_g.sum().backward()                                                        # :105 accelerator.backward(loss_g)
        opt_g.step()                                                                   # :106
        loss_d = torch.tensor(0.).to(dev)                                              # :110
        if item_sync:                                                                  # :118-119
            losses = torch.tensor([loss_g, loss_recon, loss_l1, loss_perceptual, loss_ffl, loss_dsl_features, zero, loss_quant,
                                   loss_disc, loss_d])
            out["loss_g"] = [v.item() for v in losses][0]
        else:
            out["loss_g"] = loss_g
    for i in range(warmup):
        iteration(xs[i % len(xs)])
    (sync_fn or torch.cuda.synchronize)()
    t0 = time.perf_counter()
    for i in range(steps):
        iteration(xs[i % len(xs)])
    (sync_fn or torch.cuda.synchronize)()
    dt = time.perf_counter() - t0
    B = xs[0].shape[0]
    return {"ms_per_step": 1e3 * dt / steps, "images_per_s": B * steps / dt, "loss_g": float(out["loss_g"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--codebook", type=int, default=16384)
    ap.add_argument("--no-ddp", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    import favae_hip
    from favae_step import TrainStep
    from utils import synthetic_batch
    favae_hip.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    use_ddp = not args.no_ddp
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29547")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    xs = [synthetic_batch(args.batch, args.res, args.res, 1234 + i).to(dev) for i in range(2)]
    lr = 4.5e-6 * args.batch
    rows = {}

    def fresh():
        torch.manual_seed(0)
        return build_model(dev, args.codebook)
    # TrainStep on the same model / inputs: the headline path
    ts = TrainStep(fresh(), lr=lr, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01)
    for i in range(2):
        ts.step(xs[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        o = ts.step(xs[i % 2])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows["TrainStep (bench.py headline path)"] = {"ms_per_step": 1e3 * dt / args.steps, "images_per_s": args.batch * args.steps / dt,
                                                  "loss_g": float(o["loss_g"].reshape(-1)[0])}
    del ts
    torch.cuda.empty_cache()
    arms = [("reference loop (DDP, torch.optim.Adam, .item() x 10)" if use_ddp else "reference loop without DDP", dict(ddp=use_ddp)),
            ("  ... without the ten .item() read-backs", dict(ddp=use_ddp, item_sync=False)),
            ("  ... with favae_step.FlatAdam for torch.optim.Adam (one-line swap, DDP-safe)", dict(ddp=use_ddp, optimizer="flat")),
            ("reference loop without DDP (one process: accelerate wraps nothing)", dict(ddp=False)),
            ("  ... with favae_step.FlatAdam", dict(ddp=False, optimizer="flat")),
            ("  ... with favae_step.FlatAdam(direct_grads=True)", dict(ddp=False, optimizer="flat_direct"))]
    for name, kw in arms:
        rows[name] = reference_loop(fresh(), xs, args.steps, lr=lr, **kw)
        torch.cuda.empty_cache()
    base = rows["TrainStep (bench.py headline path)"]["ms_per_step"]
    print("# the reference's training loop (train_favae.py:68-119) on the drop-in modules, batch %d, %dx%d, codebook %d, %d timed steps"
          % (args.batch, args.res, args.res, args.codebook, args.steps))
    for k, v in rows.items():
        print("%-82s %8.2f ms/step  %7.1f images/s  (%+.1f %% vs TrainStep)  loss_g %.6f" %
              (k, v["ms_per_step"], v["images_per_s"], 100 * (v["ms_per_step"] / base - 1), v["loss_g"]))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)
    if use_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
