"""One FA-VAE training iteration on the MI355X-native kernels -- restates the hot loop of the reference's train()
(favae_scripts/train_favae.py:68-116): stage 0 (encoder + decoder + quantizer: L1 + commit + FFL + 4-level DSL, optionally
the hinge generator term with the adaptive weight of :32-39) and, with train_disc, stage 1 (discriminator, hinge loss), with
the optimizers of :292-305 (Adam, betas (0.5, 0.9), lr = base_lr * batch * world).  The perceptual term
`perceptual_weight * lpips(x, x_recon).mean()` (:77-79) is on when an LPIPS module (losses/lpips.py) is passed.

MI355X-first choices:
  * all trainable parameters, their gradients and both Adam moments live in four flat fp32 buffers; parameters are
    views into them, so the optimizer is ONE fused HIP kernel launch over the whole model and the data-parallel
    gradient exchange is ONE RCCL all-reduce over the flat gradient buffer (342 MB -> a few ms over xGMI);
  * the input batch is converted to channels-last once; every activation stays NHWC in HBM;
  * no host synchronisation inside the step: losses stay on the device (the reference's ten .item() calls per step,
    train_favae.py:118-119, are left to the caller's logging cadence).

Epoch-dependent switches of the reference (disc_start_epochs, ffl_start_epochs, train_favae.py:82,93,108) are fixed per TrainStep
(`train_disc`, `ffl_weight` / `dsl_weight` at construction): at such an epoch boundary build a new TrainStep on the same model and
carry the optimizer state over with opt_g_state_dict() / opt_d_state_dict() -> load_opt_state_dicts(); opt_g and opt_d keep separate
step counts (`t`, `t_d`), as the reference's two Adam instances do.
"""
import os

import torch
import torch.distributed as dist

from favae_hip import ops as K
from focal_frequency_loss import FocalFrequencyLoss
from losses.hinge import hinge_d_loss, hinge_g_loss
from losses.vqgan_losses import recon_ffl_features_loss, recon_ffl_loss, recon_sl_gaussian_features_loss


def _flatten(params, dev):
    """parameters, gradients and Adam moments of `params` as views into four flat fp32 buffers"""
    total = sum(p.numel() for p in params)
    pflat = torch.empty(total, dtype=torch.float32, device=dev)
    gflat = torch.zeros(total, dtype=torch.float32, device=dev)
    mflat = torch.zeros(total, dtype=torch.float32, device=dev)
    vflat = torch.zeros(total, dtype=torch.float32, device=dev)
    off = 0
    for p in params:
        n = p.numel()
        view = pflat[off:off + n].as_strided(p.shape, p.stride())
        view.copy_(p.data)
        p.data = view
        p.grad = gflat[off:off + n].as_strided(p.shape, p.stride())
        p._favae_flat = True          # ops._direct_grad: reductions accumulate straight into this (zeroed) view
        off += n
    return pflat, gflat, mflat, vflat


class FlatAdam(torch.optim.Optimizer):
    """`torch.optim.Adam(params, lr, betas, eps)` (no weight decay, no amsgrad: what train_favae.py:297-305 builds) over flat buffers --
    the one-line swap for a training loop that is not `TrainStep` (the reference's own `train()`, train_favae.py:68-119): per parameter
    group, the parameters, their `.grad`s and both moments become views into four flat fp32 buffers; `step()` is ONE `favae_adam_step`
    launch per group instead of torch's multi-tensor passes over ~400 tensors; `zero_grad()` zeroes the flat gradient buffers and keeps
    the views (`set_to_none` is ignored: a `None` gradient would cut the view).  Parameter groups keep their own `lr` (the pair-wise
    `model.sigmas` group of train_favae.py:297-299; `group["lr"]` may be changed by a scheduler), `state_dict()` /
    `load_state_dict()` speak `torch.optim.Adam`'s format, so checkpoints move between the two (train_favae.py:366-374).

    direct_grads=True lets the conv / GroupNorm / blur backward kernels ACCUMULATE straight into these `.grad` views and hand autograd
    `None` -- what TrainStep does: no AccumulateGrad per parameter, the split-K slab reductions of the weight gradients grouped on the
    second stream.  Measured on the reference's loop at batch 32 (profiles/r06_ref_loop.txt): torch.optim.Adam 134.8 ms/step, FlatAdam
    with direct_grads=False 134.8 (the multi-tensor Adam was never the cost), with direct_grads=True 132.4 (TrainStep: 129.1).
    Direct accumulation bypasses the AccumulateGrad hooks that `torch.nn.parallel.DistributedDataParallel`'s reducer is driven by, so
    it is refused when a process group with more than one rank exists (use `TrainStep(distributed=True)` there, or pass False and
    keep DDP); the default (None) is True exactly when there is no such group -- one process, or `accelerate` on one GPU, wraps nothing.
    `torch.autograd.grad()` calls of the loop (the adaptive weight of train_favae.py:32-39) keep working: inside one the kernels return
    ordinary gradient tensors and leave `.grad` alone (ops._engine_accumulates).
    Every parameter must be a CUDA fp32 tensor; there is no CPU path."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, direct_grads=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if direct_grads and multi:
            raise RuntimeError("FlatAdam(direct_grads=True) with a %d-rank process group: gradients accumulated by the kernels never reach "
                               "DDP's reducer (no AccumulateGrad hook fires) -- use TrainStep(distributed=True), or direct_grads=False "
                               "under DDP" % dist.get_world_size())
        self.direct_grads = (not multi) if direct_grads is None else bool(direct_grads)
        if self.direct_grads:
            K._ENGINE_CHECK = True               # the loop may call torch.autograd.grad() itself (train_favae.py:32-39): ops._direct_grad
        self._flat = []                          # per group: [pflat, gflat, mflat, vflat, grad views]
        for group in self.param_groups:
            ps = group["params"]
            if not ps:
                raise ValueError("FlatAdam: empty parameter group")
            dev = ps[0].device
            for p in ps:
                if p.device != dev or p.device.type != "cuda" or p.dtype != torch.float32:
                    raise RuntimeError("FlatAdam: parameters of a group must be fp32 tensors on one CUDA device (no CPU path)")
            pf, gf, mf, vf = _flatten(ps, dev)
            for p in ps:
                p._favae_flat = self.direct_grads
            self._flat.append([pf, gf, mf, vf, [p.grad for p in ps]])
            group.setdefault("step", 0)

    def zero_grad(self, set_to_none=True):
        for group, fl in zip(self.param_groups, self._flat):
            fl[1].zero_()
            for p, view in zip(group["params"], fl[4]):
                if p.grad is not view:           # someone set it to None / replaced it: the view goes back in
                    p.grad = view

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group, fl in zip(self.param_groups, self._flat):
            for p, view in zip(group["params"], fl[4]):
                if p.grad is not view:           # autograd assigned a fresh tensor (the view had been dropped): take it over
                    if p.grad is not None:
                        view.copy_(p.grad)
                    p.grad = view
            group["step"] += 1
            K.adam_step(fl[0], fl[1], fl[2], fl[3], group["step"], group["lr"], group["betas"], group["eps"], 1.0)
        return loss

    def state_dict(self):
        state, pgroups, idx = {}, [], 0
        for group, fl in zip(self.param_groups, self._flat):
            ids, off = [], 0
            for p in group["params"]:
                n = p.numel()
                if group["step"] > 0:
                    state[idx] = {"step": torch.tensor(float(group["step"])),
                                  "exp_avg": fl[2][off:off + n].as_strided(p.shape, p.stride()).clone(),
                                  "exp_avg_sq": fl[3][off:off + n].as_strided(p.shape, p.stride()).clone()}
                ids.append(idx)
                idx += 1
                off += n
            pgroups.append({"lr": group["lr"], "betas": tuple(group["betas"]), "eps": group["eps"], "weight_decay": 0, "amsgrad": False,
                            "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                            "decoupled_weight_decay": False, "params": ids})
        return {"state": state, "param_groups": pgroups}

    def load_state_dict(self, sd):
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("FlatAdam.load_state_dict: %d parameter groups, the optimizer has %d" % (len(sd["param_groups"]), len(self.param_groups)))
        for group, fl, sg in zip(self.param_groups, self._flat, sd["param_groups"]):
            if len(sg["params"]) != len(group["params"]):
                raise ValueError("FlatAdam.load_state_dict: a parameter group of a different size")
            off, step = 0, 0
            for p, i in zip(group["params"], sg["params"]):
                n = p.numel()
                st = sd["state"].get(i)
                if st is not None:
                    fl[2][off:off + n].as_strided(p.shape, p.stride()).copy_(st["exp_avg"])
                    fl[3][off:off + n].as_strided(p.shape, p.stride()).copy_(st["exp_avg_sq"])
                    step = max(step, int(st["step"]))
                else:
                    fl[2][off:off + n].zero_()
                    fl[3][off:off + n].zero_()
                off += n
            group["step"] = step
            group["lr"], group["betas"], group["eps"] = sg["lr"], tuple(sg["betas"]), sg["eps"]


class _GradMark(torch.autograd.Function):
    """Identity whose backward runs a callback: placed on a module input, it fires once the gradient with respect to that input is
    being formed, i.e. after every autograd node downstream of it in the forward pass has run its backward -- the point at which
    all weight gradients of those modules have been launched."""

    @staticmethod
    def forward(ctx, x, cb):
        ctx.cb = cb
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.cb()
        return g, None


class GradExchange:
    """Bucketed gradient all-reduce overlapped with the backward pass (what DDP's reducer does for the reference,
    favae_scripts/train_favae.py:344-347), over contiguous slices of ONE flat gradient buffer.

    `segments` lists, in the order they become final during backward, the [a, b) ranges of the flat buffer each exchange step
    covers; together they must tile the buffer exactly.  fire(i) queues the SUM all-reduce of segment i (asynchronously, behind
    everything queued so far on the compute streams); finish() queues whatever has not been fired and makes the current stream
    wait for all of them.  RCCL ("nccl" backend) on the GPU; works unchanged on CPU tensors over gloo (tests)."""

    def __init__(self, gflat, segments, side_stream=None, defer=False, timing=False):
        cover = sorted(r for seg in segments for r in seg if r[1] > r[0])
        pos = 0
        for a, b in cover:
            if a != pos:
                raise ValueError("gradient segments must tile the flat buffer: gap or overlap at %d (next range starts at %d)" % (pos, a))
            pos = b
        if pos != gflat.numel():
            raise ValueError("gradient segments end at %d, the flat buffer has %d elements" % (pos, gflat.numel()))
        self.gflat, self.segments = gflat, [[r for r in seg if r[1] > r[0]] for seg in segments]
        self.cuda = gflat.is_cuda
        self.comm = torch.cuda.Stream() if self.cuda else None
        self.side_stream = side_stream          # callable -> the stream that carries the weight gradients (or None)
        # defer: fire(i) only marks segment i; every all-reduce is queued by finish(), behind backward (FAVAE_COMM_DEFER=1).  The
        # collectives then never compete with the whole-CU conv kernels for CUs -- the A/B arm for the first multi-GPU run: whether the
        # eagerly queued segments really overlap, or slow the conv chain by more than they hide, has never been observed (profiles/HISTORY.md: round-5 DESIGN section 6).
        self.defer = defer
        # timing: one event in front of every segment's collective and one behind it, both on the communication stream (the collective
        # is queued there: its end event is ordered behind it); report() turns them into an overlap table against the backward pass
        self.timing = timing and self.cuda
        self.fired = [False] * len(self.segments)
        self.works = []
        self.marks = []                         # (segment, start event, end event, bytes)

    def set_timing(self, on):
        """switch the per-segment events on / off after construction (bench.py's comm diagnostics run both arms on one object)"""
        self.timing = bool(on) and self.cuda

    def reset(self):
        self.fired = [False] * len(self.segments)
        self.works = []
        self.marks = []

    def _launch(self, i):
        if not self.segments[i]:
            return
        nbytes = 4 * sum(b - a for a, b in self.segments[i])
        if self.cuda:
            self.comm.wait_stream(torch.cuda.current_stream())
            side = self.side_stream() if self.side_stream is not None else None
            if side is not None:
                self.comm.wait_stream(side)
            with torch.cuda.stream(self.comm):
                ev0 = None
                if self.timing:
                    ev0 = torch.cuda.Event(enable_timing=True)
                    ev0.record(self.comm)
                ws = [dist.all_reduce(self.gflat[a:b], async_op=True) for a, b in self.segments[i]]
                if self.timing:
                    # end of the segment: the COMMUNICATION stream waits for the collective (it would only queue the next segment's
                    # collective behind it anyway) and takes the time stamp.  Round 4 waited on a watcher stream per segment: that
                    # block alone cost 30 ms per step (132 -> 165, bisected on the box: profiles/r05_dist_overhead.txt)
                    for w in ws:
                        w.wait()
                    ev1 = torch.cuda.Event(enable_timing=True)
                    ev1.record(self.comm)
                    self.marks.append((i, ev0, ev1, nbytes))
            self.works += ws
        else:
            for a, b in self.segments[i]:
                self.works.append(dist.all_reduce(self.gflat[a:b], async_op=True))

    def fire(self, i):
        if self.fired[i]:
            return
        self.fired[i] = True
        if not self.defer:
            self._launch(i)

    def finish(self):
        for i in range(len(self.segments)):
            self.fire(i)                          # whatever no mark has fired yet (the last segment always)
        if self.defer:
            for i in range(len(self.segments)):
                self._launch(i)
        for w in self.works:
            w.wait()                              # GPU: the current stream waits for the collective; CPU: blocks
        self.works = []

    def report(self, ev_bwd_start, ev_bwd_end):
        """after a synchronize: per segment (bytes, start / end of its collective in ms after the start of backward, the part of it that
        ran before backward ended = overlapped, the rest = exposed)"""
        rows = []
        t_end = ev_bwd_start.elapsed_time(ev_bwd_end)
        for i, e0, e1, nb in self.marks:
            t0, t1 = ev_bwd_start.elapsed_time(e0), ev_bwd_start.elapsed_time(e1)
            ov = max(0.0, min(t1, t_end) - min(t0, t_end))
            rows.append({"segment": i, "MB": nb / 1e6, "start_ms": t0, "end_ms": t1, "overlapped_ms": ov, "exposed_ms": (t1 - t0) - ov})
        return {"backward_ms": t_end, "segments": rows}


class CommArmProbe:
    """Which arm of the gradient exchange -- every segment's all-reduce queued behind backward ("defer") or started where backward
    finishes the segment ("eager") -- is faster on THIS job is measured on the job's own first steps, not assumed (VERDICT r05 item 8:
    no N > 1 run has ever been observed here).  No extra training steps: the caller's steps are timed in blocks

        2 warm-up (defer) | 3 timed (defer) | 1 warm-up (eager) | 3 timed (eager)

    with a barrier + device synchronisation at the block boundaries only (4 in the first 9 steps).  The two block times are MAX-reduced
    over the ranks, so every rank takes the same decision from the same numbers; eager must win by more than 0.5 % (noise keeps the
    collectives off the conv kernels' CUs).  FAVAE_COMM_DEFER=0|1 in the environment pins an arm and switches the probe off;
    FAVAE_COMM_AUTO=0 keeps the default (defer).  Pure host logic: exercised over gloo on CPU (tests/test_distributed_gloo.py)."""
    PLAN = (("warm", True, 2), ("defer", True, 3), ("warm", False, 1), ("eager", False, 3))

    def __init__(self, exchange, device):
        import time
        self._now = time.perf_counter
        self.ex, self.dev = exchange, device
        self.block, self.count, self.t0, self.ms = 0, 0, None, {}
        self.active, self.choice = True, None
        self._enter()

    def _sync(self):
        dist.barrier()
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)

    def _enter(self):
        name, defer, _ = self.PLAN[self.block]
        self.ex.defer = defer
        if name != "warm":
            self._sync()
            self.t0 = self._now()

    def step_done(self):
        """call once at the end of every training step while `active`"""
        if not self.active:
            return
        name, _, n = self.PLAN[self.block]
        self.count += 1
        if self.count < n:
            return
        if name != "warm":
            self._sync()
            self.ms[name] = 1e3 * (self._now() - self.t0) / n
        self.block, self.count = self.block + 1, 0
        if self.block < len(self.PLAN):
            self._enter()
            return
        t = torch.tensor([self.ms["defer"], self.ms["eager"]], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        d, e = float(t[0]), float(t[1])
        arm = "eager" if e < 0.995 * d else "defer"
        self.ex.defer = arm == "defer"
        self.choice = {"arm": arm, "how": "measured at start: 3 steps per arm, max over ranks", "ms_per_step": {"defer": d, "eager": e}}
        self.active = False


class TrainStep:
    def __init__(self, model, lr, betas=(0.5, 0.9), eps=1e-8, codebook_weight=1.0, ffl_weight=1.0, dsl_weight=0.01,
                 sigma_lr=2.0e-7, distributed=False, train_disc=False, disc_weight=0.75, lpips=None, perceptual_weight=1.0,
                 sl_weight=0.0, gaussian_kernel=None, gaussian_sigma=None):
        self.model = model
        self.lpips = lpips                                   # losses.lpips.LPIPS in eval mode (train_favae.py:308) or None
        self.pw = perceptual_weight if lpips is not None else 0.0
        self.lr, self.betas, self.eps, self.sigma_lr = lr, betas, eps, sigma_lr
        self.cw = codebook_weight
        self.train_disc, self.disc_weight = train_disc, disc_weight
        self.ffl = FocalFrequencyLoss(loss_weight=ffl_weight, alpha=1.0) if ffl_weight > 0 else None
        self.dsl = FocalFrequencyLoss(loss_weight=dsl_weight, alpha=1.0) if dsl_weight > 0 else None
        # --SL_weight / --gaussian_kernel / --gaussian_sigma: fixed-sigma Spectrum Loss (train_favae.py:101-103,324-326)
        self.sl = FocalFrequencyLoss(loss_weight=sl_weight, alpha=1.0) if sl_weight > 0 else None
        self.sl_kernel, self.sl_sigma = gaussian_kernel, gaussian_sigma
        self.distributed = distributed and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.distributed else 1
        import favae_hip
        if self.distributed and (favae_hip.HW_QUEUES_AT_INIT is None or favae_hip.HW_QUEUES_AT_INIT < 8):
            import warnings
            warnings.warn("GPU_MAX_HW_QUEUES as HIP read it: %s.  With a process group in the process the weight-gradient stream loses its "
                          "overlap with the compute stream below 8 hardware queues (+8.6 %% step time measured); export "
                          "GPU_MAX_HW_QUEUES=8 before HIP is initialised, or import favae_hip before the first torch.cuda call"
                          % ("unknown (HIP was initialised before favae_hip was imported, or the value is not a number)"
                             if favae_hip.HW_QUEUES_AT_INIT is None else favae_hip.HW_QUEUES_AT_INIT), RuntimeWarning)
        self.t = 0                     # opt_g steps taken
        self.t_d = 0                   # opt_d steps taken (its own count: the reference's opt_d only starts at disc_start_epochs)
        # opt_g parameter set: encoder + decoder + quantizer (+ pair-wise model.sigmas at its own lr)
        main = list(model.encoder.parameters()) + list(model.decoder.parameters()) + list(model.quantizer.parameters())
        extra = [model.sigmas] if hasattr(model, "sigmas") else []
        self.params = main + extra
        self.n_main = sum(p.numel() for p in main)
        total = self.n_main + sum(p.numel() for p in extra)
        dev = main[0].device
        self.pflat, self.gflat, self.mflat, self.vflat = _flatten(self.params, dev)
        if train_disc:                 # opt_d (train_favae.py:304-305)
            self.dparams = list(model.discriminator.parameters())
            self.dpflat, self.dgflat, self.dmflat, self.dvflat = _flatten(self.dparams, dev)
        self.exchange = None
        self.comm_probe = None                    # CommArmProbe while the gradient-exchange arm is being measured
        self.comm_choice = None                   # {"arm": "defer" | "eager", "how": ..., ["ms_per_step": {...}]} once distributed
        if self.distributed:
            self._broadcast_initial_state()
            self._setup_overlapped_exchange()
        # max |w| of every conv / linear weight, refreshed with the parameters (one launch per optimizer step)
        self.wmax = [K.WeightMaxima(self.pflat, self.params)] if dev.type == "cuda" else []
        if train_disc and dev.type == "cuda":
            self.wmax.append(K.WeightMaxima(self.dpflat, self.dparams))
        for wm in self.wmax:
            wm.refresh()
        # Winograd weight records of every dense 3x3 conv of the model, refreshed behind the maxima after each optimizer step
        self.wino = K.WinoRecords(self.params) if dev.type == "cuda" and os.environ.get("FAVAE_WINO_GROUPED", "1") != "0" else None
        if self.wino is not None:
            self.wino.refresh()

    # ------------------------------------------------------------------------------------------------------------
    # data parallelism: one process per GPU, each on its own slice of the global batch
    # ------------------------------------------------------------------------------------------------------------
    def _broadcast_initial_state(self):
        """Rank 0's parameters and buffers to every rank, once (what DDP does when accelerate.prepare wraps the model,
        train_favae.py:344): ranks then stay identical because every update is computed from all-reduced quantities."""
        dist.broadcast(self.pflat, 0)
        if self.train_disc:
            dist.broadcast(self.dpflat, 0)
        flat = set(id(p) for p in self.params) | (set(id(p) for p in self.dparams) if self.train_disc else set())
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            if id(t) in flat:
                continue
            if t.dtype == torch.bool:           # e.g. the codebook's `initted` flag: collectives have no bool, go through uint8
                u = t.data.to(torch.uint8)
                dist.broadcast(u, 0)
                t.data.copy_(u.to(torch.bool))
            else:                               # every other buffer, integer ones included (BatchNorm num_batches_tracked)
                dist.broadcast(t.data, 0)

    def _setup_overlapped_exchange(self):
        """Cut the flat gradient buffer where backward finishes whole module groups and start each group's all-reduce right there
        (FAVAE_OVERLAP_COMM=0: one all-reduce after backward).  Flat layout = registration order: encoder [conv_in, down, mid, final,
        (sigmas)], decoder [(sigmas), fcm_1, conv_in, fcm_2, mid, fcm_3 | up, fcm_4, final], quantizer, model.sigmas; backward
        finishes decoder.up..final first, then the rest of the decoder and the quantizer, then encoder.mid..final, then the encoder's
        down path.  A segment may be several ranges (the sigmas are cut out of their neighbours)."""
        import os
        total = self.gflat.numel()
        if os.environ.get("FAVAE_OVERLAP_COMM", "1") == "0":
            return
        enc, dec = self.model.encoder, self.model.decoder
        if not (hasattr(dec, "up") and hasattr(enc, "mid") and list(dec.up.parameters()) and list(enc.mid.parameters())):
            return
        # Segment of a parameter = the mark behind which EVERY autograd node that uses it has run.  That is a property of the
        # module the parameter lives in: decoder.{up, fcm_4, final} are only used downstream of decoder.up's input (mark 0), the
        # other decoder submodules and the quantizer downstream of the encoder output (mark 1), encoder.{mid, final} downstream of
        # encoder.mid's input (mark 2).  Everything else goes to the last segment, which is exchanged after backward has returned:
        # the encoder's down path, and every parameter registered directly on encoder / decoder / model -- the learnable blur
        # sigmas, whose taps sit on BOTH sides of the marks (EncoderGauss blurs the conv_in and down outputs before encoder.mid's
        # input exists: their dsigma arrives after mark 2 has fired).
        seg_of = {}
        for mods, i in (((getattr(dec, n, None) for n in ("up", "fcm_4", "final")), 0),
                        ((m for n, m in dec.named_children() if n not in ("up", "fcm_4", "final")), 1),
                        ((self.model.quantizer,), 1),
                        ((enc.mid, enc.final), 2)):
            for m in mods:
                if m is not None:
                    for p in m.parameters():
                        seg_of[id(p)] = i
        segs = [[], [], [], []]
        pos = 0
        for p in self.params:
            i, n = seg_of.get(id(p), 3), p.numel()
            if segs[i] and segs[i][-1][1] == pos:
                segs[i][-1] = (segs[i][-1][0], pos + n)       # contiguous with the previous range of this segment
            else:
                segs[i].append((pos, pos + n))
            pos += n
        assert pos == total
        self.exchange = GradExchange(self.gflat, segs, side_stream=K.side_stream_flushed,
                                     # default since the end of round 4: every collective behind backward.  RCCL's reduction kernels
                                     # would otherwise share SIMDs with this library's MFMA waves -- the situation in which two of its own
                                     # kernels turned out not to be bit-reproducible (profiles/HISTORY.md: round-5 DESIGN section 6) and which nobody has been able to run
                                     # here (single-GPU boxes).  331 MB over xGMI is ~2 ms of a 135 ms step; FAVAE_COMM_DEFER=0 = eager overlap
                                     defer=os.environ.get("FAVAE_COMM_DEFER", "1") != "0",
                                     timing=os.environ.get("FAVAE_COMM_TIMING", "0") == "1")
        self._armed = False
        self._bwd_events = None
        # the arm is measured on the job's first steps unless the environment pins it (CommArmProbe)
        if "FAVAE_COMM_DEFER" in os.environ:
            self.comm_choice = {"arm": "defer" if self.exchange.defer else "eager", "how": "pinned by FAVAE_COMM_DEFER"}
        elif os.environ.get("FAVAE_COMM_AUTO", "1") == "0":
            self.comm_choice = {"arm": "defer", "how": "default (FAVAE_COMM_AUTO=0)"}
        else:
            self.comm_probe = CommArmProbe(self.exchange, self.gflat.device)
            self.comm_choice = {"arm": "defer", "how": "being measured (first 9 steps)"}

        def mark(i):
            def cb():
                if self._armed:
                    self.exchange.fire(i)
            return cb

        def marked(x, i):
            y = _GradMark.apply(x, mark(i))
            for attr in ("_favae_gnstats", "_favae_amax"):
                st = getattr(x, attr, None)
                if st is not None:              # the view shares x's version counter: the by-products the producing conv left on x
                    setattr(y, attr, st)        # (tile statistics, max|x|) stay valid behind the mark, as without marks
            return y

        def pre_hook(i):
            def hook(mod, args):
                x = args[0]
                if torch.is_grad_enabled() and isinstance(x, torch.Tensor) and x.requires_grad:
                    return (marked(x, i),) + tuple(args[1:])
                return None
            return hook

        def enc_out_hook(mod, args, out):
            h = out[0]
            if torch.is_grad_enabled() and h.requires_grad:
                return (marked(h, 1),) + tuple(out[1:])
            return None
        dec.up.register_forward_pre_hook(pre_hook(0))
        enc.register_forward_hook(enc_out_hook)
        enc.mid.register_forward_pre_hook(pre_hook(2))

    # ------------------------------------------------------------------------------------------------------------
    # checkpoint wire format of the reference (train_favae.py:366-379): {"model", "opt_g", "opt_d", "epoch", "step", "loss_recon"}
    # with opt_* = torch.optim.Adam.state_dict().  The flat moment buffers are exported / imported in that layout, so a
    # checkpoint written here loads into the reference's optimizers (and vice versa).
    # ------------------------------------------------------------------------------------------------------------
    def _adam_state_dict(self, groups, mflat, vflat, step):
        state, pgroups, idx, off = {}, [], 0, 0
        for params, lr in groups:
            ids = []
            for p in params:
                n = p.numel()
                if step > 0:
                    state[idx] = {"step": torch.tensor(float(step)),
                                  "exp_avg": mflat[off:off + n].as_strided(p.shape, p.stride()).clone(),
                                  "exp_avg_sq": vflat[off:off + n].as_strided(p.shape, p.stride()).clone()}
                ids.append(idx)
                idx += 1
                off += n
            pgroups.append({"lr": lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                            "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                            "decoupled_weight_decay": False, "params": ids})
        return {"state": state, "param_groups": pgroups}

    def _load_adam_state(self, sd, params, mflat, vflat):
        off, step = 0, 0
        for i, p in enumerate(params):
            n = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                mflat[off:off + n].as_strided(p.shape, p.stride()).copy_(st["exp_avg"])
                vflat[off:off + n].as_strided(p.shape, p.stride()).copy_(st["exp_avg_sq"])
                step = max(step, int(st["step"]))
            else:
                mflat[off:off + n].zero_()
                vflat[off:off + n].zero_()
            off += n
        return step

    def _g_groups(self):
        n_main_params = len(self.params) - (1 if hasattr(self.model, "sigmas") else 0)
        groups = [(self.params[:n_main_params], self.lr)]
        if n_main_params < len(self.params):
            groups.append((self.params[n_main_params:], self.sigma_lr))       # train_favae.py:296-299
        return groups

    def opt_g_state_dict(self):
        return self._adam_state_dict(self._g_groups(), self.mflat, self.vflat, self.t)

    def opt_d_state_dict(self):
        if not self.train_disc:              # the reference creates (and saves) opt_d even when it never steps: empty state
            return self._adam_state_dict([(list(self.model.discriminator.parameters()), self.lr)], None, None, 0)
        return self._adam_state_dict([(self.dparams, self.lr)], self.dmflat, self.dvflat, self.t_d)

    def load_opt_state_dicts(self, opt_g=None, opt_d=None):
        """Resume the optimizer moments and step count from torch.optim.Adam state dicts (the reference saves them but only
        restores the model on --resume, train_favae.py:335-341; restoring them is optional here too)."""
        if opt_g is not None:
            self.t = self._load_adam_state(opt_g, self.params, self.mflat, self.vflat)
        if opt_d is not None and self.train_disc:
            self.t_d = self._load_adam_state(opt_d, self.dparams, self.dmflat, self.dvflat)

    def checkpoint(self, epoch, step, loss_recon=None):
        """the dict the reference hands to utils.save_model (train_favae.py:366-374)"""
        state = {"model": self.model.state_dict(), "opt_g": self.opt_g_state_dict(), "opt_d": self.opt_d_state_dict(),
                 "epoch": epoch, "step": step}
        if loss_recon is not None:
            state["loss_recon"] = loss_recon
        return state

    def losses(self, x):
        """Forward + loss assembly of train() :75-102."""
        m = self.model
        x = K.to_cl(x)
        x_recon, loss_q, _logits_fake, _z, enc_feats, dec_feats = m(x, stage=0)
        out = {"loss_l1": K.l1_loss(x, x_recon), "loss_quant": loss_q}
        loss_recon = out["loss_l1"]
        if self.pw > 0:                                      # train_favae.py:77-79
            out["loss_perceptual"] = self.lpips(x, x_recon).mean()
            loss_recon = loss_recon + self.pw * out["loss_perceptual"]
        out["loss_recon"] = loss_recon
        rest = self.cw * loss_q                              # every term of loss_g whose graph does not pass through loss_recon
        loss_g = loss_recon + rest
        if self.ffl is not None:
            out["loss_ffl"] = recon_ffl_loss(self.ffl, x, x_recon)
            loss_g, rest = loss_g + out["loss_ffl"], rest + out["loss_ffl"]
        if self.dsl is not None:
            out["loss_dsl"], out["loss_dsl_levels"] = recon_ffl_features_loss(self.dsl, enc_feats, dec_feats, x.device)
            loss_g, rest = loss_g + out["loss_dsl"], rest + out["loss_dsl"]
        if self.sl is not None:                              # reverses dec_feats in place once more, like the reference
            out["loss_sl"], out["loss_sl_levels"] = recon_sl_gaussian_features_loss(self.sl, self.sl_kernel, self.sl_sigma,
                                                                                    enc_feats, dec_feats, x.device)
            loss_g, rest = loss_g + out["loss_sl"], rest + out["loss_sl"]
        if self.train_disc:                                  # train_favae.py:82-88
            out["loss_disc"] = hinge_g_loss(_logits_fake)
            out["weight_d"], g_recon, g_disc = self.adaptive_weight(loss_recon, out["loss_disc"], x_recon)
            # loss_g.backward() would walk the LPIPS stack and the discriminator a second time to reach x_recon; their
            # gradients at x_recon are already known from the adaptive weight, so step() back-propagates
            #   rest = cw * loss_q + loss_ffl + loss_dsl [+ loss_sl]            (everything else, through the graph; assembled
            #                                                                    from those terms, NOT as loss_g - loss_recon:
            #                                                                    autograd would walk loss_recon's graph with zeros)
            #   x_recon <- g_recon + w_d * disc_weight * g_disc                 (handed in at x_recon)
            # -- the same total gradient for encoder / decoder / quantizer.  The discriminator's own stage-0 gradients, which
            # the reference computes here and discards at opt_d.zero_grad() (train_favae.py:109), are not computed at all.
            out["_bwd"] = (rest, x_recon, g_recon + (out["weight_d"] * self.disc_weight) * g_disc)
            loss_g = loss_g + out["weight_d"] * self.disc_weight * out["loss_disc"]
        out["logits_fake"] = _logits_fake
        out["loss_g"] = loss_g
        out["x_recon"] = x_recon
        return out

    def adaptive_weight(self, loss_recon, loss_disc, x_recon):
        """compute_adaptive_weight (train_favae.py:32-39): ratio of the gradient norms at decoder.final[2].weight, clamped to
        [0, 1e4] and detached.  Kept as a device scalar (the reference syncs with .item() here).
        d loss / d last = (d loss / d x_recon) pulled back through the final conv only, so each loss is back-propagated to
        x_recon ONCE (data gradients only: the LPIPS stack is frozen, the discriminator's stage-0 weight gradients are never
        used) and only the final conv's weight gradient is evaluated on top; the two x_recon gradients are returned for reuse
        by the main backward.  Returns (weight, d loss_recon / d x_recon, d loss_disc / d x_recon)."""
        last = self.model.decoder.final[2].weight
        with K.no_direct_grad():
            with K.grad_only("data"):
                g_recon = torch.autograd.grad(loss_recon, x_recon, retain_graph=True)[0]
                g_disc = torch.autograd.grad(loss_disc, x_recon, retain_graph=True)[0]
            with K.grad_only("param"):
                grad_recon = torch.autograd.grad(x_recon, last, g_recon, retain_graph=True)[0]
                grad_disc = torch.autograd.grad(x_recon, last, g_disc, retain_graph=True)[0]
        w = torch.norm(grad_recon) / (torch.norm(grad_disc) + 1e-4)
        return torch.clamp(w, 0.0, 1e4).detach(), g_recon.detach(), g_disc.detach()

    def backward(self, out):
        """loss_g.backward() of train_favae.py:105 on the dict losses() returned.  With discriminator training the gradients of
        loss_recon and loss_disc at x_recon were already computed for the adaptive weight and are handed in there (see losses())."""
        bwd = out.pop("_bwd", None)
        if self.exchange is not None:           # the gradient marks start their all-reduces only inside THIS backward pass
            self.exchange.reset()               # (not in the partial passes of adaptive_weight)
            self._armed = True
            if self.exchange.timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._bwd_events = (e0, e1)
        try:
            if bwd is None:
                out["loss_g"].sum().backward()
            else:
                rest, x_recon, g_x = bwd
                torch.autograd.backward([rest.sum(), x_recon], [None, g_x])
        finally:
            if self.exchange is not None:
                self._armed = False
                if self.exchange.timing and self._bwd_events is not None:
                    K.sync_side_stream()        # "end of backward" = the compute stream behind the last weight gradient
                    self._bwd_events[1].record()

    def comm_report(self):
        """FAVAE_COMM_TIMING=1: the overlap table of the last step's gradient exchange (call after torch.cuda.synchronize())"""
        if self.exchange is None or not self.exchange.timing or self._bwd_events is None:
            return None
        return self.exchange.report(*self._bwd_events)

    def disc_step(self, x):
        """Stage 1 (train_favae.py:108-116): discriminator update on (x, x_recon.detach()); model(x, stage=1) recomputes the
        reconstruction under no_grad in train mode (second EMA codebook update of the iteration, as in the reference)."""
        K.reset_side_state()
        self.dgflat.zero_()                                  # opt_d.zero_grad(): also drops what stage 0 left here
        logits_real, logits_fake = self.model(K.to_cl(x), stage=1)
        loss_d = hinge_d_loss(logits_real, logits_fake)
        loss_d.backward()
        K.sync_side_stream()
        if self.distributed:
            dist.all_reduce(self.dgflat)
        self.t_d += 1
        K.adam_step(self.dpflat, self.dgflat, self.dmflat, self.dvflat, self.t_d, self.lr, self.betas, self.eps, 1.0 / self.world)
        if len(self.wmax) > 1:
            self.wmax[1].refresh()
        return {"loss_d": loss_d, "logits_real": logits_real, "logits_fake_d": logits_fake}

    def step(self, x):
        self.model.train()
        K.reset_side_state()                                 # nothing of an aborted earlier pass reaches this step's gradients
        if self.gflat.is_cuda:
            K.zero_arena_reset(self.gflat.device)            # one memset for all of this step's max|x| targets (ops._ARENA)
        self.gflat.zero_()
        K.set_dropout_seed(self.t + 1)                       # dropout sites (attention FCM only): fresh masks every step
        out = self.losses(x)
        self.backward(out)
        K.sync_side_stream()                                 # weight gradients run on a second stream (ops._SIDE)
        if self.exchange is not None:
            self.exchange.finish()                           # segments not yet started + wait for all (RCCL over xGMI)
        elif self.distributed:
            dist.all_reduce(self.gflat)                      # one RCCL all-reduce; averaged inside the Adam kernel
        self.t += 1
        gs = 1.0 / self.world
        nm = self.n_main
        K.adam_step(self.pflat[:nm], self.gflat[:nm], self.mflat[:nm], self.vflat[:nm], self.t, self.lr, self.betas, self.eps, gs)
        if self.pflat.numel() > nm:
            K.adam_step(self.pflat[nm:], self.gflat[nm:], self.mflat[nm:], self.vflat[nm:], self.t, self.sigma_lr, self.betas,
                        self.eps, gs)
        if self.wmax:
            self.wmax[0].refresh()
        if self.wino is not None:
            self.wino.refresh()
        if self.train_disc:
            out.update(self.disc_step(x))
        if self.comm_probe is not None and self.comm_probe.active:
            self.comm_probe.step_done()
            if not self.comm_probe.active:
                self.comm_choice = self.comm_probe.choice
        return out
