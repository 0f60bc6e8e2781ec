#!/usr/bin/env python3
"""Print the kernel table of a bench.py side file (bench_detail.json): per kernel launches/step, average launch time in step and with the
weight-gradient stream off, ms/step and the achieved fraction of its roofline.  usage: tools/kernel_table.py <bench_detail.json> [md]"""
import json
import sys

d = json.load(open(sys.argv[1]))
md = len(sys.argv) > 2 and sys.argv[2] == "md"
kt = d.get("kernel_table")
if not kt:
    sys.exit("no kernel_table in %s (bench.py --no-extras?)" % sys.argv[1])
print("# two-stream %.1f ms/step, single-stream %.1f ms/step" % (kt["ms_per_step"], kt["ms_per_step_single_stream"]))
fmt = "| %s | %d | %.1f | %.1f | %.2f | %s | %s |" if md else "%-62s n=%4d avg=%8.1f excl=%8.1f ms/step=%6.2f %-22s %s"
if md:
    print("| kernel | launches / step | avg µs in step | avg µs exclusive | ms / step | achieved in step | exclusive |")
    print("|---|---|---|---|---|---|---|")
tot = 0.0
for k in kt["kernels"]:
    n = k["launches"] / 2.0
    ms = n * k["avg_launch_us"] / 1e3
    tot += ms
    a = "%.0f %s = %.0f %%" % (k["achieved"], k["unit"], 100 * k["frac"]) if "achieved" in k else ""
    x = "%.0f = %.0f %%" % (k["achieved_single_stream"], 100 * k["frac_single_stream"]) if "achieved_single_stream" in k else ""
    name = ("`%s`" % k["kernel"]) if md else k["kernel"][:62]
    print(fmt % (name, n, k["avg_launch_us"], k.get("avg_launch_us_single_stream", 0.0), ms, a, x))
print("# sum of launch durations: %.1f ms/step" % tot)
