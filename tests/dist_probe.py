"""Child of tests/test_gpu_dist.py (run under torch.distributed.run, one rank per GPU, backend nccl = RCCL): the PRODUCT
TrainStep(distributed=True) -- initial broadcast, codebook all-reduces inside the quantizer forward, gradient marks + overlapped
bucketed all-reduce -- on each rank's slice of a global batch, against ONE non-distributed TrainStep on the whole batch
(favae_scripts/train_favae.py:239-259,344-347: what DDP + sync_codebook give the reference).  Checked: identical codebooks on every
rank and equal to the global-batch EMA; all-reduced gradients / world == global-batch gradients; after a full step() the
parameters of all ranks are identical.  world_size 1 (the single-GPU box) must be bit-identical to the non-distributed step.

FAVAE_PROBE_BACKEND=gloo: the same probe with gloo collectives on device tensors and every rank on cuda:0 -- a world-2 run of the
real model + HIP kernels + GradExchange marks + codebook all-reduces on a box with ONE GPU (gloo stages device tensors through the
host; the collectives' results are the same sums RCCL produces, the product code path is unchanged)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

import favae_oracle as O
from favae_hip import ops as K
from favae_step import TrainStep
from models.vqgan_fcm import VQGANFCM

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
BACKEND = os.environ.get("FAVAE_PROBE_BACKEND", "nccl")
if BACKEND == "gloo":
    local = local % max(1, torch.cuda.device_count())       # all ranks share the GPU(s) of the box
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if BACKEND == "gloo":
    dist.init_process_group("gloo")
else:
    dist.init_process_group("nccl", device_id=dev)
VARIANT = os.environ.get("FAVAE_PROBE_VARIANT", "gauss_resblock")
flag = {"gauss_resblock": "use_gauss_resblock", "same_conv_gauss": "use_same_conv_gauss"}[VARIANT]
mk = dict(codebook_size=256, n_embed=256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True, use_l2_quantizer=True,
          kernel_size=3, dsl_init_sigma=3.0, **{flag: True})
cfg = O.OracleConfig(codebook_size=256, variant=VARIANT, kernel_size=3)
state = O.det_state(cfg, with_disc=True)
PER = 2
xg = O.det_input(PER * world, 64, 64, 5)


def make(distributed, perturb):
    model = VQGANFCM(**mk, sync_codebook=distributed, device=dev)
    sd = {k: v.clone() for k, v in state.items()}
    if perturb:                                 # ranks > 0 start from different weights: the initial broadcast must repair it
        for k in sd:
            if sd[k].is_floating_point():
                sd[k] = sd[k] + 0.01
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    return model, TrainStep(model, lr=1e-4, distributed=distributed)


def grads(ts, x):
    ts.model.train()
    ts.gflat.zero_()
    out = ts.losses(x)
    ts.backward(out)
    K.sync_side_stream()
    if ts.exchange is not None:
        ts.exchange.finish()
    elif ts.distributed:
        dist.all_reduce(ts.gflat)
    torch.cuda.synchronize()
    return out


model, ts = make(True, perturb=rank > 0)
assert ts.distributed and ts.world == world
snaps = {}
if ts.exchange is not None and world == 1:       # the all-reduce is the identity: what fire(i) hands over must already be final
    _fire = ts.exchange.fire

    def _fire_snap(i):
        if not ts.exchange.fired[i]:
            K.sync_side_stream()                  # weight gradients of the segment are queued on the side stream
            snaps[i] = [ts.gflat[a:b].clone() for a, b in ts.exchange.segments[i]]
        return _fire(i)
    ts.exchange.fire = _fire_snap
expect_overlap = os.environ.get("FAVAE_OVERLAP_COMM", "1") != "0"
assert (ts.exchange is not None) == expect_overlap
x = xg[PER * rank:PER * (rank + 1)].to(dev)
out = grads(ts, x)
if ts.exchange is not None:
    assert all(ts.exchange.fired), ts.exchange.fired
    for i, parts in snaps.items():
        for (a, b), t in zip(ts.exchange.segments[i], parts):
            assert torch.equal(t, ts.gflat[a:b]), "segment %d [%d, %d) was exchanged before its gradient was complete" % (i, a, b)
    if world == 1:
        assert len(snaps) == len(ts.exchange.segments)
        ts.exchange.fire = _fire
g_dist = ts.gflat.clone() / world
embed = model.quantizer._codebook.embed.clone()
cluster = model.quantizer._codebook.cluster_size.clone()

# every rank holds the same exchanged gradients and codebook
for t, name in ((g_dist, "gradients"), (embed, "codebook"), (cluster, "cluster sizes")):
    ref = t.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(ref, t), "rank %d: %s differ from rank 0" % (rank, name)

# the same exchange without any overlap: a non-distributed step on this rank's slice, ONE all-reduce of the whole buffer after backward.
# Two ranks: a + b in either order, so the overlapped, segmented exchange must reproduce it bit for bit (diagnostic for the comparison
# below: tells an exchange problem from a gradient problem)
model2, ts2 = make(False, perturb=False)
grads(ts2, x)
g_simple = ts2.gflat.clone()
dist.all_reduce(g_simple)
torch.cuda.synchronize()
g_simple /= world
del model2, ts2

# one non-distributed step on the whole batch (every rank computes it: no collectives inside)
model1, ts1 = make(False, perturb=False)
grads(ts1, xg.to(dev))
scale = float(ts1.gflat.abs().max())
err = float((g_dist - ts1.gflat).abs().max()) / scale
e_err = float((embed - model1.quantizer._codebook.embed).abs().max())
if world > 1 and not err < 2e-5:                 # say where and which side before failing
    g_ref1 = ts1.gflat.clone()
    grads(ts1, xg.to(dev))
    print("[rank %d] DIAG err %.3e; overlapped exchange == plain all-reduce of local gradients: %s (max diff %.3e); global-batch "
          "reference reproducible: %s (max diff %.3e); plain exchange vs reference %.3e"
          % (rank, err, torch.equal(g_dist, g_simple), float((g_dist - g_simple).abs().max()) / scale, torch.equal(g_ref1, ts1.gflat),
             float((g_ref1 - ts1.gflat).abs().max()) / scale, float((g_simple - g_ref1).abs().max()) / scale), file=sys.stderr, flush=True)
    off, rows = 0, []
    names = {id(p): n for n, p in model.named_parameters()}
    for p in ts.params:
        n = p.numel()
        rows.append((float((g_dist[off:off + n] - g_ref1[off:off + n]).abs().max()) / scale,
                     float((g_simple[off:off + n] - g_ref1[off:off + n]).abs().max()) / scale, names.get(id(p), "?"), off, n))
        off += n
    for e1, e2, nm, o, n in sorted(rows, reverse=True)[:10]:
        print("[rank %d] DIAG   %-50s [%9d, +%8d)  overlapped %.3e  plain %.3e" % (rank, nm, o, n, e1, e2), file=sys.stderr, flush=True)
    ts1.gflat.copy_(g_ref1)
if world == 1:
    assert torch.equal(g_dist, ts1.gflat), "world 1: distributed step is not bit-identical (%g)" % err
    assert torch.equal(embed, model1.quantizer._codebook.embed)
else:
    # per-sample-mean losses: sum over ranks / world == global-batch gradient up to fp32 summation order
    assert err < 2e-5, "all-reduced gradients differ from the global-batch gradients: %g of the max" % err
    assert e_err < 1e-6, "codebook differs from the global-batch EMA update: %g" % e_err
    assert torch.equal(cluster, model1.quantizer._codebook.cluster_size), "cluster sizes differ from the global-batch EMA update"

# a full step() (exchange + Adam) keeps the ranks identical
ts.step(x)
torch.cuda.synchronize()
rep = ts.comm_report()                       # FAVAE_COMM_TIMING=1: when did each segment's collective run, relative to backward?
if rep is not None and rank == 0:
    print("COMM TABLE world=%d defer=%s backward %.2f ms" % (world, ts.exchange.defer, rep["backward_ms"]))
    for r in rep["segments"]:
        print("  segment %d  %7.2f MB  start %8.3f ms  end %8.3f ms  overlapped %7.3f ms  exposed %7.3f ms"
              % (r["segment"], r["MB"], r["start_ms"], r["end_ms"], r["overlapped_ms"], r["exposed_ms"]))
p0 = ts.pflat.clone()
dist.broadcast(p0, 0)
assert torch.equal(p0, ts.pflat), "rank %d: parameters diverged after step()" % rank
dist.barrier()
if rank == 0:
    print("DIST PROBE OK world=%d backend=%s variant=%s overlap=%s grad_err=%.2e embed_err=%.2e"
          % (world, BACKEND, VARIANT, expect_overlap, err, e_err))
dist.destroy_process_group()
