"""CPU (cross-compile only): no 128-bit buffer store of the library is followed, in the next issue slot, by an instruction that writes
its data registers.  gfx950 needs a wait state there (the ISA manuals' store-data hazard); the compiler's hazard recogniser leaves it out
when the store takes its soffset from an SGPR, as every `__builtin_amdgcn_raw_buffer_store_b128` of this library does -- round 4 found
wrong first components in gn_bwd_apply_rows_kernel<., false> next to the weight-gradient stream (tools/experiments/
apply_race.py; common.h bstore carries the wait state).  The scanner must also still SEE the pattern in a build without the wait state."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fa-vae_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _asm(src, out, *defs):
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-w", "-S", "--cuda-device-only", "-I" + CSRC, "-o", out,
           os.path.join(CSRC, src)] + list(defs)
    subprocess.run(cmd, check=True, timeout=900)


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_unguarded_128bit_buffer_store(tmp_path):
    import store_hazard_scan as S
    S.BUFFER128 = True
    # only translation units that can contain such a store at all: those including a header (or holding code) with the builtin
    users = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        text = open(os.path.join(CSRC, f)).read()
        incs = [l.split('"')[1] for l in text.splitlines() if l.startswith('#include "') and l.split('"')[1].endswith(".h")]
        blob = text + "".join(open(os.path.join(CSRC, h)).read() for h in incs if h != "common.h" and os.path.exists(os.path.join(CSRC, h)))
        if "bstore(" in blob or "raw_buffer_store_b128(" in blob:
            users.append(f)
    assert "norm.hip" in users and "conv.hip" in users
    hits = []
    for f in users:
        out = str(tmp_path / (f + ".s"))
        _asm(f, out)
        hits += S.scan(out)
    assert not hits, hits[:5]
    # the scanner still recognises the pattern: the apply kernels without the wait state have it
    out = str(tmp_path / "norm_nonop.s")
    _asm("norm.hip", out, "-DFAVAE_NO_STORE_NOP")
    assert S.scan(out), "the build without the wait state no longer shows the pattern: the scanner (or the kernel) changed"
