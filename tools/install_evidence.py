#!/usr/bin/env python3
"""Copy the outputs of one `bash tools/r06_run.sh <tag> tests_all bench prof traffic configs` call from gpurun_out/<tag>/ into profiles/
under the round's names (the bench line alone, without the banner lines a launcher may print in front of it).
usage: python tools/install_evidence.py <tag> [round prefix, default r06]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r06"
O, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


def line_of(path):
    return [l for l in open(path).read().splitlines() if l.startswith("{")][-1]


d = json.loads(line_of(os.path.join(O, "bench.json")))
open(os.path.join(P, rnd + "_bench.json"), "w").write(json.dumps(d) + "\n")
print("bench: %.2f images/s, %.2f ms/step, peak %.2f GiB, reference_loop %s, dominant kernel %.1f us, frac %.4f" %
      (d["value"], d["ms_per_step"], d.get("peak_mem_gib", 0), d.get("reference_loop", {}).get("images_per_s"), d["roofline"]["avg_launch_us"],
       d["roofline"]["frac"]))
for a, b in (("bench_detail.json", "_bench_detail.json"), ("kernel_stats.csv", "_kernel_stats.csv"), ("timeline.txt", "_timeline.txt"),
             ("hbm_traffic.json", "_hbm_traffic.json"), ("pytest.log", "_gpu_tests.log"), ("parity_margins.txt", "_parity_margins.txt")):
    if os.path.exists(os.path.join(O, a)):
        shutil.copy(os.path.join(O, a), os.path.join(P, rnd + b))
for f in ("fp16", "bf16", "cfg5_fp32", "cfg5_fp16", "cfg5_bf16", "f4"):
    src = os.path.join(O, "bench_%s.json" % f)
    if os.path.exists(src):
        e = json.loads(line_of(src))
        open(os.path.join(P, "%s_bench_%s.json" % (rnd, f)), "w").write(json.dumps(e) + "\n")
        print("%-10s %.1f images/s, %.1f ms/step, peak %s GiB" % (f, e["value"], e["ms_per_step"], e.get("peak_mem_gib")))
t = json.load(open(os.path.join(P, rnd + "_hbm_traffic.json")))
print("traffic file build %s, this tree %s" % (t.get("_build"), bench.build_stamp()))
for l in open(os.path.join(P, rnd + "_gpu_tests.log")):
    if " passed" in l or " failed" in l:
        print(l.strip())
