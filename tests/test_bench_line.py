"""CPU: the machine-readable line of bench.py stays small (the driver keeps an 8 KB tail of stdout; round 2's 39 KB line was
unparseable) and the PMC traffic file is only used for the build it was collected on."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _rec(n, us, gflop, gb):
    return {"launches": n, "total_us": n * us, "min_us": us, "max_us": us, "flops": n * gflop * 1e9, "bytes": n * gb * 1e9}


def test_compact_roofline_is_small_and_numeric():
    name = "conv_wgrad_nine_sp_kernel<2, 2, 128, false, 1, float>"
    r = _rec(268, 927.4123, 177.41, 0.534)
    full = bench.roofline_entry(name, r, 4 * 174000.0, {name: {"hbm_bytes_per_launch_corrected": 1.873e9}}, {name: _rec(134, 672.0, 177.41, 0.534)})
    c = bench.compact_roofline(full)
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "algorithmic_bytes_per_launch",
              "traffic_over_algorithmic", "frac_single_stream"):
        assert k in c, k
    # peak = the hardware peak of the 16-bit matrix pipe (MI355X_MICROARCH.md); the h3 kernels issue 3 MFMA products per fp32 multiply-add
    assert c["bound"] == "mfma" and c["unit"] == "TFLOP/s" and c["peak"] == 2500.0 and abs(c["peak_fp32_equivalent"] - 2500.0 / 3) < 0.01
    assert abs(c["frac"] - c["achieved"] / c["peak"]) < 1e-4
    assert abs(c["frac_fp32_equivalent"] - 3 * c["frac"]) < 1e-4
    assert abs(c["executed_gflop_per_launch"] - 3 * 177.41) < 0.01 and abs(c["frac_of_pipe_peak"] - 3 * c["frac"]) < 1e-4
    assert abs(c["traffic_over_algorithmic"] - 1.873 / 0.534) < 1e-3
    assert not any(isinstance(v, str) and len(v) > 80 for v in c.values())
    assert len(json.dumps(c)) < 700


def test_winograd_kernel_is_priced_on_executed_flops():
    """the Winograd kernel's line shows the fraction of the real 2.5 PFLOP/s pipe next to the fp32-equivalent one --
    executed = algorithmic x 3 products x 4/9 (the numbers of the round-3 driver run: 177.41 GFLOP in 631 us -> 0.150 and 0.337)."""
    name = "conv3x3_wino_sp_kernel<0, true, false>"
    e = bench.compact_roofline(bench.roofline_entry(name, _rec(268, 631.0, 177.41, 0.534), 4 * 135000.0, {}, None))
    assert abs(e["achieved"] - 281.2) < 0.5 and e["peak"] == 2500.0 and abs(e["frac"] - 0.1125) < 1e-3
    assert abs(e["executed_gflop_per_launch"] - 177.41 * 3 * 4 / 9) < 0.01
    assert abs(e["frac_of_pipe_peak"] - 0.150) < 1e-3 and abs(e["frac_fp32_equivalent"] - 0.337) < 1e-3


def test_one_plane_winograd_kernel_is_priced_with_one_product():
    """round 6: the kernel names carry the operand scheme and the activation storage type -- conv3x3_wino_sp_kernel<XFORM, GB, SE, WIDE,
    PLN, storage>: PLN 4 (one bf16 plane, b1) executes algorithmic x 1 product x 4/9, PLN 2 (h3) x 3 x 4/9"""
    for name, prod in (("conv3x3_wino_sp_kernel<0, true, false, true, 4, unsigned short>", 1), ("conv3x3_wino_sp_kernel<0, true, false, true, 2, float>", 3),
                       ("conv3x3_wino_sp_kernel<2, false, true, true, 1, float>", 1)):
        assert bench.kernel_planes(name) == {1: 4, 3: 2}[prod] or (prod == 1 and bench.kernel_planes(name) in (1, 4))
        e = bench.roofline_entry(name, _rec(67, 400.0, 177.41, 0.534), 4 * 135000.0, {}, None)
        assert abs(e["executed_gflop_per_launch"] - 177.41 * prod * 4 / 9) < 0.01, (name, e["executed_gflop_per_launch"])
    assert bench.kernel_planes("conv3x3_halo_sp_kernel<2, 4, 3, false, false, true, unsigned short>") == 4
    assert bench.kernel_planes("conv_wgrad_nine_sp_kernel<2, 4, 128, false, 1, unsigned short>") == 4


def test_f44_winograd_kernel_is_priced_on_executed_flops():
    """round 5: the F(4x4, 3x3) kernel executes 36 of the direct conv's 144 products per 16 outputs, x 3 split products: executed =
    algorithmic x 3 / 4, against the same 2.5 PFLOP/s pipe (it once showed up at frac 1.64 of the fp32-MFMA peak: no planes matched)."""
    name = "conv3x3_wino4_sp_kernel<0, true, false>"
    e = bench.compact_roofline(bench.roofline_entry(name, _rec(42, 1018.0, 618.5, 2.7), 2 * 135000.0, {}, None))
    assert e["peak"] == 2500.0 and abs(e["frac"] - 618.5e9 / 1018e-6 * 1e-12 / 2500.0) < 1e-4 and e["frac"] < 0.5
    assert abs(e["executed_gflop_per_launch"] - 618.5 * 3 / 4) < 0.01 and abs(e["frac_of_pipe_peak"] - 0.75 * e["frac"]) < 1e-4


def test_no_matrix_kernel_is_priced_above_its_pipe():
    """The h3 kernels without a scheme template argument (codebook scores, SDPA batched GEMM) run on the 16-bit pipe with 3 products per
    multiply-add: priced against 2500, not against the fp32-MFMA 157.3 (the round-4 table once showed the VQ product at frac 1.52)."""
    for name in ("vq_dist_top2_h3_kernel", "bgemm_sp_kernel<0, 1>"):
        e = bench.roofline_entry(name, _rec(2, 286.8, 68.7, 0.025), 2 * 135000.0, {}, None)       # 2*8192*16384*256 FLOP in 287 us
        assert e["peak"] == 2500.0 and e["products_per_fma"] == 3 and e["frac"] < 0.2 and abs(e["frac_of_pipe_peak"] - 3 * e["frac"]) < 1e-9
    e = bench.roofline_entry("vq_dist_top2_kernel", _rec(2, 780.0, 68.7, 0.025), 2 * 135000.0, {}, None)   # the fp32-MFMA arm
    assert e["peak"] == 157.3 and e["frac"] < 1.0


def test_line_is_trimmed_not_asserted():
    res = {"metric": "m", "value": 1.0, "config": {"workload": "w" * 300}, "roofline": {"kernel": "k"}, "with_lpips": {"x": "y" * 5000},
           "cpu_baseline": {"value": 1}}
    line = bench.fit_line(res)
    out = json.loads(line)
    assert len(line) < bench.MAX_LINE_BYTES and "with_lpips" not in out and "roofline" in out and "cpu_baseline" in out


def test_hbm_bound_entry_uses_gb_per_s():
    e = bench.roofline_entry("gn_bwd_apply_rows_kernel", _rec(75, 316.2, 0.0, 0.876), 174000.0, {}, None)
    assert e["bound"] == "hbm" and e["unit"] == "GB/s" and e["peak"] == 8000.0
    assert abs(e["achieved"] - 0.876e9 / 316.2e-6 * 1e-9) < 1.0 and abs(e["frac"] - e["achieved"] / 8000.0) < 1e-9


def test_traffic_file_is_bound_to_the_build(tmp_path, monkeypatch):
    stamp = bench.build_stamp()
    assert len(stamp) == 16 and stamp == bench.build_stamp()
    p = tmp_path / "t.json"
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(p))
    assert bench.load_traffic()[1]["status"] == "missing"
    p.write_text(json.dumps({"_build": "0" * 16, "k<1>": {"hbm_bytes_per_launch_corrected": 1.0}}))
    t, meta = bench.load_traffic()
    assert t == {} and meta["status"] == "stale"
    p.write_text(json.dumps({"_build": stamp, "k<1>": {"hbm_bytes_per_launch_corrected": 1.0}}))
    t, meta = bench.load_traffic()
    assert meta["status"] == "ok" and t["k<1>"]["hbm_bytes_per_launch_corrected"] == 1.0


def test_rounding_keeps_five_digits():
    assert bench.rnd(184.6647291815485) == 184.66 and bench.rnd(None) is None and bench.rnd(3) == 3
