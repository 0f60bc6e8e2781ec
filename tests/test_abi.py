"""CPU: the C-ABI library loads and exports exactly the entry points include/favae_hip.h declares; the Python binding
mirrors them one to one; the product path refuses to run without a GPU (no fallback)."""
import os
import re

import pytest
import torch

import favae_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "favae_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(favae_[a-z0-9_]+)\s*\(", src))


def test_header_binding_and_library_agree():
    syms = header_symbols()
    assert len(syms) >= 35
    assert syms == set(favae_hip.SIGNATURES), syms ^ set(favae_hip.SIGNATURES)
    lib = favae_hip.load()
    for s in syms:
        assert hasattr(lib, s), f"libfavae_hip.so does not export {s}"
    assert lib.favae_abi_version() == favae_hip.ABI_VERSION


def test_no_cpu_fallback():
    from favae_hip import ops
    x = torch.zeros(1, 4, 8, 8)
    with pytest.raises(RuntimeError):
        ops.to_cl(x)
    with pytest.raises(RuntimeError):
        ops.focal_frequency_loss(x, x, 1.0)


def test_dropin_surface_and_state_dict_keys():
    """models.vqgan_fcm.VQGANFCM keeps the reference's constructor and state_dict key set/shapes."""
    import favae_oracle as O
    from models.vqgan_fcm import VQGANFCM
    m = VQGANFCM(1024, 256, ch_mult=(1, 1, 2, 2, 4), attn_resolutions=[16], use_cosine_sim=True, use_l2_quantizer=True,
                 kernel_size=9, dsl_init_sigma=3.0, use_gauss_resblock=True, device="cpu")
    shapes = O.param_shapes(O.OracleConfig(codebook_size=1024, variant="gauss_resblock"))
    sd = m.state_dict()
    assert set(sd) == set(shapes)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert hasattr(m.decoder.final[2], "weight") and hasattr(m.encoder, "sigmas") and hasattr(m.decoder, "sigmas")
    m2 = VQGANFCM(512, 3, ch_mult=(1, 2, 4), attn_resolutions=[], use_cosine_sim=True, codebook_dim=32, use_l2_quantizer=True,
                  kernel_size=3, dsl_init_sigma=3.0, use_same_conv_gauss=True, num_groups=3, device="cpu")
    shapes2 = O.param_shapes(O.OracleConfig(codebook_size=512, n_embed=3, ch_mult=(1, 2, 4), attn_resolutions=(), codebook_dim=32,
                                            kernel_size=3, variant="same_conv_gauss", num_groups=3))
    assert {k: tuple(v.shape) for k, v in m2.state_dict().items()} == {k: tuple(v) for k, v in shapes2.items()}
    with pytest.raises(ValueError):
        m.forward(torch.zeros(1), stage=2)


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` with fewer than N GPUs visible exits 2 before any GPU call (here: no GPU at all); the N-rank
    launch itself is covered on the GPU box (tests/test_gpu_dist.py)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has GPUs: covered by tests/test_gpu_dist.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "--gpus 2 but only" in out.stderr, out.stderr[-500:]


def test_launch_profiler_reports_empty_without_launches():
    lib = favae_hip.load()
    assert lib.favae_prof_enable(3) != 0 and lib.favae_prof_enable(0) == 0
    import ctypes
    n = lib.favae_prof_report(None, 0)
    buf = ctypes.create_string_buffer(int(n) + 8)
    lib.favae_prof_report(buf, len(buf))
    assert buf.value == b""


def test_weight_cache_keys_follow_every_visible_write():
    """a cached max|w| / Winograd record of a parameter that is a view of a flat buffer is valid only while the
    parameter's version counter, the FLAT BUFFER's version counter and the storage's update count are the ones seen at refresh --
    pflat.copy_(checkpoint) / dist.broadcast(pflat) move only the flat buffer's counter.  Host logic only (no kernel runs here)."""
    from favae_hip import ops as K
    flat = torch.zeros(64 + 32)
    p1 = torch.nn.Parameter(torch.zeros(4, 4, 2, 2))
    p2 = torch.nn.Parameter(torch.zeros(8, 4))
    off = 0
    for p in (p1, p2):
        p.data = flat[off:off + p.numel()].view(p.shape)
        off += p.numel()
    wm = K.WeightMaxima(flat, [p1, p2])
    wm.versions = [K._weights_key(p) for p in wm.params]          # what refresh() records behind its kernel launch
    assert K._weight_amax(p1) is not None and K._weight_amax(p2) is not None
    flat.mul_(2.0)                                                # a write through the flat buffer
    assert K._weight_amax(p1) is None and K._weight_amax(p2) is None
    wm.versions = [K._weights_key(p) for p in wm.params]
    with torch.no_grad():
        p1.add_(1.0)                                              # a write through the parameter
    assert K._weight_amax(p1) is None and K._weight_amax(p2) is not None
    wm.versions = [K._weights_key(p) for p in wm.params]
    K.invalidate_weight_caches(flat)                              # a raw-pointer write announced by its author
    assert K._weight_amax(p1) is None and K._weight_amax(p2) is None


def test_cout64_wants_split_weights_exactly_when_the_winograd_kernel_takes_it():
    """a dense 3x3 conv with 64 output channels has ONE split-operand kernel, the Winograd one.
    `favae_conv_wants_split_weights` must say "h3 records" exactly when `favae_conv_wino_ok` does -- under every A/B switch that takes
    the Winograd kernel away (FAVAE_CONV_HALO=0, FAVAE_WINO=0), with a fused affine, and for an input-channel count the fused
    GroupNorm staging does not hold -- or Python would build plain h3 records that only the 128-channel tiles understand.
    Host logic only (child processes: the switches are read once per process)."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from ctypes import byref\n"
        "from favae_hip import query\n"
        "from favae_hip.ops import make_conv_desc, GATHER_PLAIN, ACT_NONE, ACT_SILU\n"
        "for cin, aff, act in ((64, 0, ACT_NONE), (64, 1, ACT_SILU), (1024, 1, ACT_SILU), (128, 0, ACT_NONE)):\n"
        "    d = make_conv_desc(2, 32, 32, cin, 32, 32, 64, 3, 3, 1, 1, GATHER_PLAIN, act, 1)\n"
        "    print(cin, aff, query('favae_conv_wants_split_weights', byref(d), aff), query('favae_conv_wino_ok', byref(d), aff))\n"
    ) % os.path.join(ROOT, "fa-vae_amd")
    seen_on = seen_off = 0
    for env_extra in ({}, {"FAVAE_CONV_HALO": "0"}, {"FAVAE_WINO": "0"}):
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        for line in out.stdout.strip().splitlines():
            cin, aff, wants, wino = (int(v) for v in line.split())
            assert (wants == 2) == (wino == 1) and wants in (0, 2), (env_extra, line)
            if env_extra:
                assert wants == 0, (env_extra, line)
                seen_off += 1
            elif cin <= 512:
                assert wants == 2, line
                seen_on += 1
            else:
                assert wants == 0, line                 # 1024 input channels with a fused affine: beyond the LDS staging of (scale, shift)
    assert seen_on == 3 and seen_off == 8


def test_partial_sum_tile_counts_follow_the_kernel_that_will_run():
    """The per-tile partial sums of the statistics / GroupNorm-backward epilogues are on the grid of the kernel that writes them: the
    F(2x2) Winograd kernel in its 16 x 8 x 128 tiling where Cout % 128 == 0, in its 16 x 16 x 64 tiling otherwise or under
    FAVAE_WINO_WIDE=0 / favae_set_wino_wide(0), the F(4x4) kernel (16 x 16 partials) when the call will pass F(4x4) records -- the
    queries take the planes word of that call (ABI 19).  A count for the wrong kernel under- or over-sizes the buffer the consumer
    then sums.  Host logic only."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from ctypes import byref\n"
        "from favae_hip import query\n"
        "from favae_hip.ops import make_conv_desc, GATHER_PLAIN, ACT_NONE, ACT_SILU, PLANES_WINO4\n"
        "H, W = 64, 96\n"
        "for cin, cout in ((128, 128), (64, 64), (128, 192), (256, 256)):\n"
        "    d = make_conv_desc(2, H, W, cin, H, W, cout, 3, 3, 1, 1, GATHER_PLAIN, ACT_SILU, 1)\n"
        "    d2 = make_conv_desc(2, H, W, cin, H, W, cout, 3, 3, 1, 1, GATHER_PLAIN, ACT_NONE, 1)\n"
        "    row = [query('favae_conv_stats_tiles', byref(d), 1, 0), query('favae_conv_gnbwd_tiles', byref(d2), 0),\n"
        "           query('favae_conv_stats_tiles', byref(d), 1, PLANES_WINO4), query('favae_conv_gnbwd_tiles', byref(d2), PLANES_WINO4),\n"
        "           query('favae_conv_wino4_ok', byref(d2), 0)]\n"
        "    prev = query('favae_set_wino_wide', 0)\n"
        "    row += [query('favae_conv_stats_tiles', byref(d), 1, 0), query('favae_conv_gnbwd_tiles', byref(d2), 0), prev]\n"
        "    query('favae_set_wino_wide', prev)\n"
        "    print(cin, cout, *row)\n"
    ) % os.path.join(ROOT, "fa-vae_amd")
    fine, coarse = (64 // 8) * (96 // 16), (64 // 16) * (96 // 16)
    for env_extra, wide_default in (({}, 1), ({"FAVAE_WINO_WIDE": "0"}, 0)):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        rows = [[int(v) for v in line.split()] for line in out.stdout.strip().splitlines()]
        assert len(rows) == 4
        for cin, cout, st, gb, st4, gb4, f44_ok, st_off, gb_off, prev in rows:
            wide = wide_default and cout % 128 == 0
            assert prev == wide_default
            assert st == gb == (fine if wide else coarse), (env_extra, cin, cout, st, gb)
            assert st_off == gb_off == coarse, (env_extra, cin, cout)
            if f44_ok:                                       # F(4x4) records: that kernel's 16 x 16 partials whatever the F(2x2) tiling
                assert st4 == gb4 == coarse, (env_extra, cin, cout, st4, gb4)
            else:
                assert st4 == st and gb4 == gb


def test_package_asks_for_eight_hardware_queues_before_hip_is_initialised():
    """Round 5 finding (profiles/r05_dist_overhead.txt): with HIP's default of 4 hardware queues an initialised RCCL process group takes
    the overlap of the weight-gradient stream away (+8.6 % step time at every N >= 2).  Importing favae_hip before the first HIP call
    exports GPU_MAX_HW_QUEUES=8 (an explicit setting of the user is kept); bench.py does the same at the top of main()."""
    import subprocess
    import sys
    code = "import os, sys; sys.path.insert(0, %r); import favae_hip; print(os.environ.get('GPU_MAX_HW_QUEUES'))" % os.path.join(ROOT, "fa-vae_amd")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "8", out.stdout + out.stderr[-1000:]
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="16"), capture_output=True, text=True, timeout=300)
    assert out.stdout.strip() == "16"
    # what TrainStep's warning looks at: the value HIP read (a number), or None when it cannot be known -- a non-numeric value must not
    # raise (int() on the raw environment variable did)
    code2 = "import sys; sys.path.insert(0, %r); import favae_hip; print(favae_hip.HW_QUEUES_AT_INIT)" % os.path.join(ROOT, "fa-vae_amd")
    for val, want in (("8", "8"), ("lots", "None"), (None, "8")):
        e = dict(env) if val is None else dict(env, GPU_MAX_HW_QUEUES=val)
        out = subprocess.run([sys.executable, "-c", code2], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.strip() == want, (val, out.stdout, out.stderr[-500:])
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")') < src.index('dist.init_process_group("nccl"')


def test_direct_accumulation_asks_the_engine_what_the_pass_does_with_a_gradient():
    """ops._engine_accumulates (host logic, runs on the CPU): True in .backward() and for the named leaves of backward(inputs=...), False
    inside torch.autograd.grad() -- for the differentiated leaf (the engine captures the returned tensor there) and for every other one
    (its gradient is not wanted): favae_step.FlatAdam's direct mode returns ordinary tensors and leaves .grad alone in those cases"""
    import torch
    from favae_hip import ops as K
    w = torch.ones(3, requires_grad=True)
    b = torch.ones(3, requires_grad=True)
    seen = []

    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, y):
            return x * 2 + y

        @staticmethod
        def backward(ctx, g):
            seen.append((K._engine_accumulates(w), K._engine_accumulates(b)))
            return g * 2, g
    F.apply(w, b).sum().backward()
    torch.autograd.grad(F.apply(w, b).sum(), w)
    torch.autograd.grad(F.apply(w, b).sum(), [w, b])
    F.apply(w, b).sum().backward(inputs=[w])
    assert seen == [(True, True), (False, False), (False, False), (True, False)], seen


def test_identity_nodes_created_first_run_last_in_backward():
    """what the late gradients lean on (favae_hip/ops.py _LateGradFn): the autograd engine runs ready nodes latest-created first, so an
    identity node made before anything else of the forward pass runs after every other node of the backward pass -- a weight gradient
    parked at it is not looked at until the rest of backward has been queued.  (If a torch release changed that order the gradients
    would still be right -- the node waits for the side stream before it hands its tensor on -- but the overlap would be gone.)"""
    import torch
    from favae_hip import ops as K
    order = []

    class Mark(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, tag):
            ctx.tag = tag
            return x * w.sum()

        @staticmethod
        def backward(ctx, g):
            order.append(ctx.tag)
            return g, g.sum().expand(3), None

    ws = [torch.ones(3, requires_grad=True) for _ in range(4)]
    seen = []
    for i, w in enumerate(ws):
        w.register_hook(lambda g, i=i: seen.append((i, list(order))))
    alias = [K._LateGradFn.apply(w) for w in ws]              # start of the forward pass
    assert all(a.data_ptr() == w.data_ptr() and a.requires_grad for a, w in zip(alias, ws))
    h = torch.ones(3, requires_grad=True)
    x = h * 1.0
    for i, a in enumerate(alias):
        x = Mark.apply(x, a, i)
    x.sum().backward()
    assert order == [3, 2, 1, 0]
    # every parameter's gradient arrived only after ALL the conv-like nodes had run, the alias made last first
    assert [i for i, _ in seen] == [3, 2, 1, 0] and all(o == [3, 2, 1, 0] for _, o in seen), seen
    assert all(w.grad is not None for w in ws) and h.grad is not None
