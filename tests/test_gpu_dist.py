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


def test_model_under_torch_ddp():
    """The drop-in VQGANFCM wrapped in torch DDP(find_unused_parameters=True, broadcast_buffers=True) -- what accelerate.prepare does
    to it in the reference's train_favae.py -- trains one iteration with torch.optim.Adam and lands on the parameters of the
    unwrapped run (child process: it owns a process group)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ddp_probe.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
    assert "DDP PROBE OK" in out.stdout
