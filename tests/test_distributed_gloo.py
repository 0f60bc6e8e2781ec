"""CPU, world_size 2, gloo: the data-parallel scheme of favae_step.TrainStep is correct by construction.

TrainStep exchanges exactly three things per step: SUM all-reduce of the flat gradient buffer
(scaled by 1/world inside the Adam kernel) and the two SUM all-reduces of the codebook statistics inside the quantizer
forward (reference models/l2_quantize.py:419,427).  Here the same exchanges are driven through the CPU oracle on 2 gloo
ranks and compared with ONE process stepping on the concatenated global batch: identical codebooks on all ranks, equal to
the global-batch EMA, and rank-averaged gradients equal to the global-batch gradients (every loss term is a per-sample
mean).  The HIP kernels themselves cannot run here (no GPU, no fallback); the host-side flat-buffer plumbing is
checked in test_trainstep_flat_views."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import favae_oracle as O

TINY = dict(codebook_size=64, n_embed=32, ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(16,), resolution=32,
            kernel_size=3, variant="gauss_resblock")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = O.OracleConfig(**TINY)
    sc = O.StepConfig(lr=1e-3, dsl_weight=0.01)
    tr = O.OracleTrainer(cfg, sc, dtype=torch.float64)
    xg = O.det_input(4, 32, 32, 5, torch.float64)
    x = xg[2 * rank:2 * rank + 2]

    def grad_avg(g):
        dist.all_reduce(g)
        g.div_(world)

    for _ in range(2):
        res = tr.step(x, all_reduce=dist.all_reduce, grad_all_reduce=grad_avg)
    # numpy arrays are pickled BY VALUE; torch tensors would travel as shared-memory file descriptors that die with this process
    # (ConnectionResetError in q.get when the producer exits first: 2 of 11 runs in round 2)
    out = {"embed": tr.P["quantizer._codebook.embed"], "cluster": tr.P["quantizer._codebook.cluster_size"],
           "w": tr.P["encoder.conv_in.weight"], "sig": tr.P["decoder.sigmas"], "g": res["grads"]["decoder.final.2.weight"]}
    q.put((rank, {k: v.detach().cpu().numpy().copy() for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_global_batch_step():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, d in got.items()}
    # single process, global batch
    cfg = O.OracleConfig(**TINY)
    tr = O.OracleTrainer(cfg, O.StepConfig(lr=1e-3, dsl_weight=0.01), dtype=torch.float64)
    xg = O.det_input(4, 32, 32, 5, torch.float64)
    for _ in range(2):
        res = tr.step(xg)
    ref = {"embed": tr.P["quantizer._codebook.embed"], "cluster": tr.P["quantizer._codebook.cluster_size"],
           "w": tr.P["encoder.conv_in.weight"].detach(), "sig": tr.P["decoder.sigmas"].detach(),
           "g": res["grads"]["decoder.final.2.weight"]}
    for k in ref:
        assert torch.equal(got[0][k], got[1][k]) or float((got[0][k] - got[1][k]).abs().max()) < 1e-12, f"ranks diverged: {k}"
        err = float((got[0][k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30))
        assert err < 1e-9, f"{k}: 2-rank result differs from the global-batch step by {err:.2e}"


def test_trainstep_flat_views():
    """Host plumbing of TrainStep: parameters/gradients are views of the flat buffers (one Adam launch, one all-reduce),
    channels-last conv weights keep their OHWI memory, pair-wise sigmas sit in their own tail segment."""
    from favae_step import TrainStep
    from models.vqgan_fcm import VQGANFCM
    m = VQGANFCM(64, 3, ch_mult=(1, 2, 4), attn_resolutions=[], use_cosine_sim=True, codebook_dim=8, use_l2_quantizer=True,
                 kernel_size=3, dsl_init_sigma=3.0, use_same_conv_gauss=True, num_groups=3, device="cpu")
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    ts = TrainStep(m, lr=1e-4)
    n_train = sum(p.numel() for k, p in m.named_parameters() if not k.startswith("discriminator."))
    assert ts.pflat.numel() == n_train and ts.pflat.numel() - ts.n_main == 4
    lo, hi = ts.pflat.data_ptr(), ts.pflat.data_ptr() + 4 * ts.pflat.numel()
    for k, p in m.named_parameters():
        if k.startswith("discriminator."):
            continue
        assert lo <= p.data_ptr() < hi and torch.equal(p.detach(), before[k]), k
        assert p.grad is not None and p.grad.stride() == p.stride()
        if p.dim() == 4 and p.shape[2] > 1:
            assert p.is_contiguous(memory_format=torch.channels_last), k
    assert m.sigmas.data_ptr() == ts.pflat.data_ptr() + 4 * ts.n_main
    ts.gflat.fill_(1.0)
    assert float(m.encoder.conv_in.weight.grad.sum()) == m.encoder.conv_in.weight.numel()


# ---------------------------------------------------------------------------------------------------------------------------
# Host logic of the PRODUCT TrainStep at world_size 2 (gloo, CPU): initial broadcast, gradient marks, bucketed overlapped
# all-reduce (favae_step.GradExchange).  The HIP kernels cannot run here, so the model is a stand-in with the same module
# skeleton (encoder.{down,mid,final}, decoder.{head,up,final}, quantizer) made of nn.Linear layers; everything that is exercised --
# flat buffers, segment tiling, hook placement, firing order, collectives -- is the product code.
# ---------------------------------------------------------------------------------------------------------------------------
class _ToyEnc(torch.nn.Module):
    """EncoderGauss layout (models/codec.py:193-314): conv_in/down, mid, final and its OWN `sigmas`, whose taps sit before AND after
    `mid` -- the taps before `mid` run their backward after the gradient mark on mid's input has fired."""

    def __init__(self):
        super().__init__()
        self.down = torch.nn.Linear(6, 8)
        self.mid = torch.nn.Linear(8, 8)
        self.final = torch.nn.Linear(8, 4)
        self.sigmas = torch.nn.Parameter(torch.tensor([3.0, 2.0, 1.5, 1.0]))

    def forward(self, x):
        feats = []
        h = torch.tanh(self.down(x))
        feats.append(h * self.sigmas[0])
        feats.append(torch.sin(h) * self.sigmas[1])
        h = torch.tanh(self.mid(h))
        feats.append(h * self.sigmas[2])
        h = self.final(h)
        feats.append(h * self.sigmas[3])
        return h, feats


class _ToyDec(torch.nn.Module):
    """decoder with its own sigmas registered FIRST (models/codec.py:882-1004) and taps before and after `up`"""

    def __init__(self):
        super().__init__()
        self.sigmas = torch.nn.Parameter(torch.tensor([1.0, 2.0]))
        self.head = torch.nn.Linear(4, 8)
        self.up = torch.nn.Linear(8, 8)
        self.final = torch.nn.Linear(8, 6)

    def forward(self, z):
        h = torch.tanh(self.head(z))
        t0 = h * self.sigmas[0]
        h = torch.tanh(self.up(h))
        return self.final(h), [t0, h * self.sigmas[1]]


class _ToyModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder, self.decoder = _ToyEnc(), _ToyDec()
        self.quantizer = torch.nn.Linear(4, 4)
        self.register_buffer("codebook", torch.randn(5, 4))

    def forward(self, x):
        h, ef = self.encoder(x)
        y, df = self.decoder(self.quantizer(h))
        self.taps = sum((f ** 2).mean() for f in ef + df)         # the DSL terms: every tap reaches the loss
        return y


def _exchange_worker(rank, world, port, q, overlap):
    # overlap: True = segments fired by the marks, "defer" = the same segments, every collective queued by finish() (FAVAE_COMM_DEFER=1),
    # False = one all-reduce after backward
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      FAVAE_OVERLAP_COMM="1" if overlap else "0", FAVAE_COMM_DEFER="1" if overlap == "defer" else "0")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from favae_step import TrainStep
    torch.manual_seed(100 + rank)                            # ranks start DIFFERENT: the initial broadcast must fix that
    model = _ToyModel()
    ts = TrainStep(model, lr=1e-3, distributed=True, ffl_weight=0.0, dsl_weight=0.0)
    state0 = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}      # numpy: pickled by value through the queue
    order = []
    if overlap:
        assert ts.exchange is not None and ts.exchange.defer == (overlap == "defer")
        fire0 = ts.exchange.fire
        ts.exchange.fire = lambda i: (order.append(i) if not ts.exchange.fired[i] else None, fire0(i))[1]
    xs = torch.randn(4, 6, generator=torch.Generator().manual_seed(7))
    per = 4 // world
    x = xs[per * rank:per * rank + per]
    ts.gflat.zero_()
    snaps = {}
    if overlap:                                              # the invariant of the scheme: a segment is FINAL when it is fired
        fire1 = ts.exchange.fire

        def fire_snap(i):
            if not ts.exchange.fired[i] and world == 1:      # world 1: the all-reduce is the identity, the buffer stays local
                snaps[i] = [ts.gflat[a:b].clone() for a, b in ts.exchange.segments[i]]
            return fire1(i)
        ts.exchange.fire = fire_snap
    loss = ((model(x) - x) ** 2).mean() + 0.1 * model.taps
    ts.backward({"loss_g": loss})
    if overlap == "defer":
        assert not ts.exchange.works, "deferred mode queued a collective inside backward"
    for i, parts in snaps.items():
        for (a, b), t in zip(ts.exchange.segments[i], parts):
            assert torch.equal(t, ts.gflat[a:b]), "segment %d [%d, %d) was exchanged before its gradient was complete" % (i, a, b)
    local = ts.gflat.clone() if not overlap else None
    if ts.exchange is not None:
        ts.exchange.finish()
    else:
        dist.all_reduce(ts.gflat)
    q.put((rank, {"state0": state0, "g": ts.gflat.clone().numpy(), "order": order, "segments": ts.exchange.segments if overlap else None,
                  "snapshots": len(snaps)}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, "defer", False])
def test_trainstep_host_logic_two_ranks(overlap):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # (1) the initial broadcast made every rank start from rank 0's parameters and buffers
    torch.manual_seed(100)
    ref_model = _ToyModel()
    for k, v in ref_model.state_dict().items():
        assert np.array_equal(got[0]["state0"][k], v.numpy()) and np.array_equal(got[1]["state0"][k], v.numpy()), k
    # (2) the exchanged flat gradient = sum over ranks of the local gradients = world x the global-batch gradient
    xs = torch.randn(4, 6, generator=torch.Generator().manual_seed(7))
    loss = ((ref_model(xs) - xs) ** 2).mean() + 0.1 * ref_model.taps
    loss.backward()
    main = list(ref_model.encoder.parameters()) + list(ref_model.decoder.parameters()) + list(ref_model.quantizer.parameters())
    gref = torch.cat([p.grad.reshape(-1) for p in main]) * world           # mean over 4 samples vs sum of two means over 2
    for r in range(world):
        assert float((torch.from_numpy(got[r]["g"]) - gref).abs().max()) < 1e-5 * float(gref.abs().max())
    if overlap:
        # (3) segments tile the buffer, and the marks fired them in backward order: decoder tail, decoder head + quantizer,
        # encoder mid/final, then finish() fires the encoder's down path
        assert got[0]["order"] == [0, 1, 2, 3]
        segs = got[0]["segments"]
        n = gref.numel()
        assert sorted(r for s in segs for r in s)[0][0] == 0 and sorted(r for s in segs for r in s)[-1][1] == n
        # (4) the sigmas of encoder and decoder (taps on both sides of the marks) are exchanged with the LAST segment only
        sig_ranges, pos = [], 0
        for mod in (ref_model.encoder, ref_model.decoder, ref_model.quantizer):        # the flat layout of TrainStep
            for name, p in mod.named_parameters():
                if name == "sigmas":
                    sig_ranges.append((pos, pos + p.numel()))
                pos += p.numel()
        assert len(sig_ranges) == 2
        for a, b in sig_ranges:
            owners = [i for i, sg in enumerate(segs) for (lo, hi) in sg if lo < b and a < hi]
            assert owners == [3], (a, b, owners)


def _probe_worker(rank, world, port, q, slow_eager):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), FAVAE_OVERLAP_COMM="1")
    os.environ.pop("FAVAE_COMM_DEFER", None)
    os.environ.pop("FAVAE_COMM_AUTO", None)
    import time
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from favae_step import TrainStep
    torch.manual_seed(100)
    model = _ToyModel()
    ts = TrainStep(model, lr=1e-3, distributed=True, ffl_weight=0.0, dsl_weight=0.0)
    assert ts.comm_probe is not None and ts.comm_probe.active and ts.exchange.defer and ts.comm_choice["how"].startswith("being measured")
    x = torch.randn(2, 6, generator=torch.Generator().manual_seed(7 + rank))
    arms = []
    for step in range(12):                                  # the host side of TrainStep.step(): backward, finish, probe tick
        arms.append("defer" if ts.exchange.defer else "eager")
        ts.gflat.zero_()
        loss = ((model(x) - x) ** 2).mean() + 0.1 * model.taps
        ts.backward({"loss_g": loss})
        if not ts.exchange.defer and slow_eager and rank == 1:
            time.sleep(0.05)                                # ONE rank is slow in the eager arm: the MAX over ranks must decide for all
        if ts.exchange.defer and not slow_eager and rank == 0:
            time.sleep(0.05)
        ts.exchange.finish()
        if ts.comm_probe.active:
            ts.comm_probe.step_done()
            if not ts.comm_probe.active:
                ts.comm_choice = ts.comm_probe.choice
    q.put((rank, {"arms": arms, "choice": ts.comm_choice}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("slow_eager", [True, False])
def test_comm_arm_is_measured_at_start_and_agreed_across_ranks(slow_eager):
    """favae_step.CommArmProbe over gloo, world 2: 2 warm-up + 3 timed steps deferred, 1 + 3 eager, then every rank keeps the SAME arm
    -- the one whose slowest rank was faster -- for the rest of the job."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_probe_worker, args=(r, world, port, q, slow_eager)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = "defer" if slow_eager else "eager"
    for r in range(world):
        assert got[r]["arms"][:9] == ["defer"] * 5 + ["eager"] * 4, got[r]["arms"]
        assert got[r]["arms"][9:] == [want] * 3, got[r]["arms"]
        ch = got[r]["choice"]
        assert ch["arm"] == want and ch["how"].startswith("measured at start") and set(ch["ms_per_step"]) == {"defer", "eager"}
    assert got[0]["choice"]["ms_per_step"] == got[1]["choice"]["ms_per_step"], "ranks decided on different numbers"


def test_comm_arm_pinned_by_the_environment():
    from favae_step import CommArmProbe  # noqa: F401
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fa-vae_amd", "favae_step.py")).read()
    assert 'if "FAVAE_COMM_DEFER" in os.environ:' in src and '"pinned by FAVAE_COMM_DEFER"' in src


def test_segments_are_final_when_fired():
    """World 1 (the all-reduce is the identity): what fire(i) hands to the collective equals the gradient at the end of backward,
    for every segment.  A layout that puts the encoder's sigmas into the encoder.mid segment fails here: the taps in front of `mid`
    add their dsigma after mark 2 has fired -- TrainStep therefore assigns segments per owning module and sends
    every parameter registered directly on encoder / decoder / model with the last segment."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_exchange_worker, args=(0, 1, _free_port(), q, True))
    p.start()
    rank, got = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert got["order"] == [0, 1, 2, 3] and got["snapshots"] == 4


def test_grad_exchange_rejects_bad_tilings():
    from favae_step import GradExchange
    g = torch.zeros(10)
    with pytest.raises(ValueError):
        GradExchange(g, [[(0, 4)], [(5, 10)]])
    with pytest.raises(ValueError):
        GradExchange(g, [[(0, 6)], [(4, 10)]])
    with pytest.raises(ValueError):
        GradExchange(g, [[(0, 9)]])
    GradExchange(g, [[(4, 10)], [(0, 4), (10, 10)]])


def _flat_adam_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from favae_step import FlatAdam
    p = [torch.nn.Parameter(torch.zeros(4))]
    try:
        FlatAdam(p, lr=1e-3, direct_grads=True)
        msg = "accepted"
    except RuntimeError as e:
        msg = str(e)
    try:                                   # the default falls back to ordinary gradients there; on the CPU it then stops at the device check
        FlatAdam(p, lr=1e-3)
        msg2 = "accepted"
    except RuntimeError as e:
        msg2 = str(e)
    q.put((rank, msg, msg2, hasattr(p[0], "_favae_flat")))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_adam_refuses_direct_accumulation_in_a_multi_rank_group():
    """gradients the kernels accumulate straight into the flat buffer never pass an AccumulateGrad hook, i.e. never reach DDP's reducer:
    with more than one rank FlatAdam(direct_grads=True) must not be constructible (TrainStep(distributed=True) owns that case)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_adam_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg, msg2, marked in got:
        assert "2-rank process group" in msg and "TrainStep(distributed=True)" in msg, msg
        assert "no CPU path" in msg2, msg2          # direct_grads=None resolved to False, then the (CPU) parameters were refused
        assert not marked, "a refused construction leaves the parameters untouched"
