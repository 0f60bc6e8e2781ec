"""CPU (cross-compile only): no 128-bit buffer store of the library is followed, in the next issue slot, by an instruction that writes
its data registers.  gfx950 needs a wait state there (the ISA manuals' store-data hazard); the compiler's hazard recogniser leaves it out
when the store takes its soffset from an SGPR, as every `__builtin_amdgcn_raw_buffer_store_b128` of this library does -- round 4 found
wrong first components in gn_bwd_apply_rows_kernel<., false> next to the weight-gradient stream (profiles/HISTORY.md, round-5 DESIGN section 6, tools/experiments/
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


def _product_flags():
    """optimisation-relevant flags of csrc/Makefile (the assembly scanned here must be the product's: e.g. -fno-slp-vectorize)"""
    flags = []
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("CXXFLAGS"):
            flags += [t for t in line.split("=", 1)[1].split() if t.startswith("-f") and t != "-fPIC"]
    return flags


def _asm(src, out, *defs, flags=None):
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-w", "-S", "--cuda-device-only", "-I" + CSRC, "-o", out,
           os.path.join(CSRC, src)] + (_product_flags() if flags is None else flags) + list(defs)
    subprocess.run(cmd, check=True, timeout=900)


@pytest.fixture(scope="module")
def unit_asm(tmp_path_factory):
    """every translation unit cross-compiled to assembly once, with the product's flags (in parallel)"""
    from concurrent.futures import ThreadPoolExecutor
    d = tmp_path_factory.mktemp("asm")
    units = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    outs = {f: str(d / (f + ".s")) for f in units}
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(lambda f: _asm(f, outs[f]), units))
    return outs


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_unguarded_128bit_buffer_store(unit_asm, tmp_path):
    import store_hazard_scan as S
    S.BUFFER128 = True
    # only translation units that can contain such a store at all: those including a header (or holding code) with the builtin
    users = []
    for f in sorted(unit_asm):
        text = open(os.path.join(CSRC, f)).read()
        incs = [l.split('"')[1] for l in text.splitlines() if l.startswith('#include "') and l.split('"')[1].endswith(".h")]
        blob = text + "".join(open(os.path.join(CSRC, h)).read() for h in incs if h != "common.h" and os.path.exists(os.path.join(CSRC, h)))
        if "bstore(" in blob or "act_store4<" in blob or "raw_buffer_store_b128(" in blob:      # act_store4<float> = bstore (common.h)
            users.append(f)
    assert "norm.hip" in users        # (conv.hip lost its last 128-bit buffer store with the operand-plane path, round 6; still scanned below)
    users = sorted(set(users) | {"conv.hip"})
    hits = []
    for f in users:
        hits += S.scan(unit_asm[f])
    assert not hits, hits[:5]
    # the scanner still recognises the pattern: the apply kernels without the wait state have it
    out = str(tmp_path / "norm_nonop.s")
    _asm("norm.hip", out, "-DFAVAE_NO_STORE_NOP")
    assert S.scan(out), "the build without the wait state no longer shows the pattern: the scanner (or the kernel) changed"


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_half_dead_packed_fp32_result_is_overwritten(unit_asm, tmp_path):
    """Round 4's FFT finding (profiles/HISTORY.md: round-5 DESIGN section 6): SLP vectorisation made `v_pk_fma_f32 v[6:7], ...` of which only v7 was used, followed within
    three instructions by an unpacked write of v6 -- and exactly those low halves came out wrong in 2-10 % of the launches next to MFMA
    waves on the same SIMD.  Round 5 builds EVERY unit without SLP (profiles/r05_slp_ab.txt: neutral in time); no unit's assembly may
    show the pattern, the Makefile must carry the flag, and the scanner must still find the pattern in an SLP build of ffl.hip."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "experiments"))
    import importlib
    argv, sys.argv = sys.argv, [sys.argv[0]]              # the scanner is a script: it scans sys.argv[1:] at import
    try:
        P = importlib.import_module("pk_waw_scan")
    finally:
        sys.argv = argv
    assert "-fno-slp-vectorize" in _product_flags()
    for f, path in unit_asm.items():
        hits = P.scan(path)
        assert not hits, (f, hits[:3])
    out = str(tmp_path / "ffl_slp.s")
    _asm("ffl.hip", out, flags=[])
    assert P.scan(out), "the SLP build of ffl.hip no longer shows the pattern: the scanner (or the compiler) changed"
