"""GPU: the multi-GPU code path of bench.py / TrainStep (RCCL process group, codebook all-reduces inside the quantizer
forward, flat-gradient all-reduce, max-over-ranks timing) runs under torch.distributed.run.  The GPU box has one MI355X, so
the launch is world_size 1 with FAVAE_FORCE_DIST=1; numerical equivalence of the exchange pattern for world_size 2 is
covered on CPU by tests/test_distributed_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_rccl():
    env = dict(os.environ, FAVAE_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--codebook", "1024", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["config"]["parallelism"] == "dp1"
    assert res["config"]["loss_g_last"] == res["config"]["loss_g_last"]          # not NaN
    # a distributed bench line diagnoses itself -- both arms of the gradient exchange (collectives queued behind
    # backward / started where backward finishes a segment) with the per-segment overlap table and the two codebook all-reduces
    c = res["comm"]
    assert c["default_arm"] in ("defer", "eager") and c["faster_arm"] in ("defer", "eager") and set(c["arms"]) == {"defer", "eager"}
    for arm, a in c["arms"].items():
        assert a["ms_per_step"] > 0 and a["backward_ms"] > 0, arm
        assert len(a["segments"]) == 4 and all(len(r) == 5 for r in a["segments"]), a["segments"]
        assert abs(sum(r[0] for r in a["segments"]) - 4e-6 * 82.7e6) < 0.05 * 4e-6 * 82.7e6, "the segments tile the 82.7 M-float gradient buffer"
        assert len(a["codebook_allreduce"]) == 2 and all(len(r) == 2 and r[1] >= 0 for r in a["codebook_allreduce"])
        assert abs(a["codebook_allreduce"][1][0] - 1024 * 256 * 4e-6) < 1e-3          # embedding sums: codebook 1024 x 256 floats
        for mb, t0, t1, ov, ex in a["segments"]:
            assert t1 >= t0 and ov >= 0 and ex >= -1e-3 and abs((t1 - t0) - (ov + ex)) < 1e-2
    for mb, t0, t1, ov, ex in c["arms"]["defer"]["segments"]:
        assert ov < 0.05, "deferred arm: nothing starts before backward has ended"
    # the arm the timed region ran was MEASURED on the job's first steps (favae_step.CommArmProbe), world 1 over RCCL here
    ch = c["chosen"]
    assert ch["arm"] in ("defer", "eager") and ch["arm"] == c["default_arm"] and ch["how"].startswith("measured at start"), ch
    assert set(ch["ms_per_step"]) == {"defer", "eager"} and min(ch["ms_per_step"].values()) > 0
    assert ("queued behind backward" if ch["arm"] == "defer" else "overlapped with backward") in res["config"]["gradient_exchange"]


def test_model_under_torch_ddp():
    """The drop-in VQGANFCM wrapped in torch DDP(find_unused_parameters=True, broadcast_buffers=True) -- what accelerate.prepare does
    to it in the reference's train_favae.py -- trains one iteration with torch.optim.Adam and lands on the parameters of the
    unwrapped run (child process: it owns a process group)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ddp_probe.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
    assert "DDP PROBE OK" in out.stdout


def _run_probe(world, variant, overlap, port, backend="nccl", extra_env=None):
    """One launch, no repetition: a run whose ranks disagree numerically FAILS.  Rounds 3-4 repeated the
    two-processes-on-one-GPU runs up to twice because the FFT kernels were not bit-reproducible next to another process's MFMA waves
    (profiles/HISTORY.md: round-5 DESIGN section 6); `ffl.hip` has been built without SLP vectorisation since (0 of 1200 two-process steps differ), so a disagreement is a
    finding again.  The probe's DIAG lines are printed with the failure."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FAVAE_PROBE_VARIANT=variant, FAVAE_OVERLAP_COMM="1" if overlap else "0",
               FAVAE_PROBE_BACKEND=backend, **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_probe.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    diag = "\n".join(l for l in out.stderr.splitlines() if "DIAG" in l or "Error" in l)
    assert out.returncode == 0, out.stdout[-1500:] + diag[-4000:] + out.stderr[-3000:]
    assert "DIST PROBE OK world=%d" % world in out.stdout
    return out.stdout


@pytest.mark.parametrize("variant,overlap", [("gauss_resblock", True), ("same_conv_gauss", True), ("gauss_resblock", False)])
def test_product_trainstep_distributed_world1(variant, overlap):
    """TrainStep(distributed=True) through RCCL at world size 1 (what this box has): initial broadcast, codebook all-reduces,
    gradient marks and the overlapped bucketed all-reduce all run, and the result is bit-identical to the non-distributed step."""
    port = 29551 + 2 * [("gauss_resblock", True), ("same_conv_gauss", True), ("gauss_resblock", False)].index((variant, overlap))
    _run_probe(1, variant, overlap, port)


@pytest.mark.parametrize("defer", ["0", "1"])
def test_gradient_exchange_overlap_table(defer):
    """the probe prints, per gradient segment, when its collective started and ended relative to the backward pass
    (FAVAE_COMM_TIMING=1) -- eagerly queued (FAVAE_COMM_DEFER=0) and deferred to the end of backward (FAVAE_COMM_DEFER=1, the default; the two arms for the
    first real multi-GPU run).  World 1 over RCCL here; the same command prints the table at world N.  Results stay bit-identical."""
    out = _run_probe(1, "gauss_resblock", True, 29565 + int(defer), extra_env={"FAVAE_COMM_TIMING": "1", "FAVAE_COMM_DEFER": defer})
    assert "COMM TABLE world=1 defer=%s" % (defer == "1") in out
    rows = [l for l in out.splitlines() if l.strip().startswith("segment ")]
    assert len(rows) == 4, out[-1500:]
    if defer == "1":                                  # nothing may start before backward has ended
        for l in rows:
            assert float(l.split("overlapped")[1].split("ms")[0]) < 0.05, l
    print(out[out.index("COMM TABLE"):])


@pytest.mark.parametrize("variant", ["gauss_resblock", "same_conv_gauss"])
def test_product_trainstep_two_ranks_equal_global_batch(variant):
    """Two ranks on two GPUs against one rank on the concatenated batch: codebooks identical, gradients within 2e-5."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs on the node (this box has %d)" % torch.cuda.device_count())
    _run_probe(2, variant, True, 29581 + 2 * ["gauss_resblock", "same_conv_gauss"].index(variant))


@pytest.mark.parametrize("variant,overlap", [("gauss_resblock", True), ("same_conv_gauss", True), ("gauss_resblock", False)])
def test_product_trainstep_two_ranks_on_one_gpu(variant, overlap):
    """World size 2 of the REAL product path on the single-GPU box: two fresh child processes share cuda:0 and exchange through gloo
    on device tensors -- real VQGANFCM + HIP kernels + gradient marks / GradExchange + the two codebook all-reduces
    (models/l2_quantize.py:419,427; favae_scripts/train_favae.py:344-347), 2 ranks x batch 2 against one rank on the concatenated
    batch: gradients <= 2e-5 of the max, codebooks <= 1e-6, cluster sizes equal, parameters identical on both ranks after step()."""
    # one rendezvous port per variant: consecutive launches on the same port raced with the previous store's teardown once (round 4)
    port = 29571 + 2 * [("gauss_resblock", True), ("same_conv_gauss", True), ("gauss_resblock", False)].index((variant, overlap))
    out = _run_probe(2, variant, overlap, port, backend="gloo")
    assert "backend=gloo" in out


def test_bench_gpus_flag_launches_ranks():
    """`python bench.py --gpus N` without a launcher starts N ranks itself; with more ranks than GPUs it refuses (exit 2) before
    touching a GPU.  N = the number of GPUs of the node (1 here: the in-process path) and N = that + 1 (the refusal)."""
    import torch
    n = torch.cuda.device_count()
    bench = os.path.join(ROOT, "bench.py")
    small = ["--steps", "1", "--warmup", "1", "--batch", "2", "--codebook", "256", "--res", "64", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, bench, "--gpus", str(n + 1)] + small, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "only %d GPU" % n in out.stderr
    if n >= 2:
        out = subprocess.run([sys.executable, bench, "--gpus", str(n)] + small, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert res["n_gpus"] == n and res["rccl_world_size"] == n and len(res["per_rank_images_per_s"]) == n


def test_bench_line_schema():
    """The JSON line of bench.py carries what the driver's contract and the tier's measurement section name: metric / value / unit /
    n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, `roofline`
    {bound, achieved, peak, unit, frac, traffic} for the dominant kernel measured by the in-library profiler, `cpu_baseline`
    {value, unit, cores, kind, sample}.  Small batch, same model."""
    detail = os.path.join(ROOT, "gpurun_out", "bench_detail_schema_test.json")
    env = dict(os.environ, FAVAE_BENCH_DETAIL=detail)
    env.pop("WORLD_SIZE", None)
    os.makedirs(os.path.dirname(detail), exist_ok=True)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--no-extras",
                          "--cpu-batch", "1", "--cpu-steps", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    assert len(lines[0]) < 4096, "the driver keeps an 8 KB tail: tables and notes belong in the side file, not the line (%d bytes)" % len(lines[0])
    res = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in res, k
    assert res["unit"] == "images/s" and res["n_gpus"] == 1 and res["steps"] == 2 and res["warmup"] == 1 and res["dtype"] == "f32"
    assert res["higher_is_better"] is True and res["scaling"] == "weak" and res["vs_baseline"] is None and res["data"] == "synthetic"
    assert "workload" in res["config"] and "model" not in res["config"]
    assert abs(res["value"] - 2 * 2 / (res["ms_per_step"] * 2e-3)) < 1e-6 * res["value"]
    r = res["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "avg_launch_us", "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 * r["frac"]
    # checkable against the guide: peak is the hardware peak of the pipe, the executed MFMA work is priced next to the algorithmic one
    assert r["peak"] in (2500.0, 157.3)
    for k in ("executed_gflop_per_launch", "frac_of_pipe_peak", "peak_fp32_equivalent", "frac_fp32_equivalent"):
        assert k in r, k
    ex = r["executed_gflop_per_launch"] / r["avg_algorithmic_gflop_per_launch"]
    assert abs(r["frac_of_pipe_peak"] - ex * r["frac"]) < 2e-3 * r["frac_of_pipe_peak"]
    assert abs(r["frac_fp32_equivalent"] - r["achieved"] / r["peak_fp32_equivalent"]) < 2e-3 * r["frac_fp32_equivalent"]
    assert "median" in res["cpu_baseline"]["sample"] and len(res["cpu_baseline"]["step_s"]) == 1
    assert "<" in r["kernel"], "the kernel is named by its full instantiation (the name a rocprofv3 trace shows)"
    assert not any(isinstance(v, str) and len(v) > 80 for v in r.values()), "no prose inside the machine-readable roofline object"
    for k in ("kernel_table", "roofline_others"):
        assert k not in res, "%s belongs in the side file" % k
    assert os.path.exists(detail) and "roofline" in json.load(open(detail))
    c = res["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "images/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
