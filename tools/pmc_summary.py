#!/usr/bin/env python3
"""Markdown table of rocprofv3 --pmc passes (tools/pmc_conv.sh): per kernel the counters averaged over its dispatches, and the derived
matrix-pipe / vector-ALU / LDS utilisation and effective clock.  usage: tools/pmc_summary.py <dir with p*/**/*_counter_collection.csv>"""
import collections
import csv
import glob
import os
import re
import sys

csv.field_size_limit(1 << 30)
KEEP = ("conv3x3_halo_sp_kernel", "conv3x3_wino_sp_kernel", "conv3x3_wino4_sp_kernel", "conv3x3_winow_sp_kernel", "conv_wgrad_nine_sp_kernel", "conv_wgrad_row3_sp_kernel", "conv_fwd_sp_kernel", "conv_wgrad_sp_kernel")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<.*?>)?)\(", name)
    return (m.group(1) if m else name)[:80]


vals = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> [values]
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*_counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k.startswith(KEEP):
            continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
kernels = sorted(vals, key=lambda k: -sum(dur[k]))


def avg(k, c):
    v = vals[k].get(c)
    return sum(v) / len(v) if v else float("nan")


rows = [("dispatches / avg µs (profiled passes)", lambda k: "%d / %.0f" % (len(dur[k]) // 4 or len(dur[k]), sum(dur[k]) / len(dur[k])))]


def derived(k):
    cyc = avg(k, "GRBM_GUI_ACTIVE") / 8.0                      # per-XCD active cycles
    us = sum(dur[k]) / len(dur[k])
    out = collections.OrderedDict()
    out["effective clock (GRBM_GUI_ACTIVE / 8 / duration)"] = "%.2f GHz" % (cyc / us * 1e-3)
    out["MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMD x cycles)"] = "%.1f %%" % (100 * avg(k, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cyc))
    out["VALU busy = 4 x SQ_ACTIVE_INST_VALU / (1024 x cycles)"] = "%.1f %%" % (100 * 4 * avg(k, "SQ_ACTIVE_INST_VALU") / (1024 * cyc))
    out["resident waves / SIMD = 4 x SQ_WAVE_CYCLES / (1024 x cycles)"] = "%.1f" % (4 * avg(k, "SQ_WAVE_CYCLES") / (1024 * cyc))
    wc = avg(k, "SQ_WAVE_CYCLES")
    out["wave time: issue-stalled (SQ_WAIT_INST_ANY) / parked (SQ_WAIT_ANY) / issuing (SQ_ACTIVE_INST_ANY)"] = "%.0f / %.0f / %.0f %%" % (
        100 * avg(k, "SQ_WAIT_INST_ANY") / wc, 100 * avg(k, "SQ_WAIT_ANY") / wc, 100 * avg(k, "SQ_ACTIVE_INST_ANY") / wc)
    out["LDS-issue stall share of wave time (SQ_WAIT_INST_LDS)"] = "%.1f %%" % (100 * avg(k, "SQ_WAIT_INST_LDS") / wc)
    m = avg(k, "SQ_INSTS_MFMA")
    out["instructions per MFMA: VALU / SALU / LDS / VMEM_RD"] = "%.2f / %.2f / %.2f / %.2f" % (
        avg(k, "SQ_INSTS_VALU") / m, avg(k, "SQ_INSTS_SALU") / m, avg(k, "SQ_INSTS_LDS") / m, avg(k, "SQ_INSTS_VMEM_RD") / m)
    out["LDS bank-conflict cycles / LDS index cycles"] = "%.1f %%" % (100 * avg(k, "SQ_LDS_BANK_CONFLICT") / max(avg(k, "SQ_LDS_IDX_ACTIVE"), 1))
    out["LDS array busy = SQ_LDS_IDX_ACTIVE / (256 CU x cycles)"] = "%.1f %%" % (100 * avg(k, "SQ_LDS_IDX_ACTIVE") / (256 * cyc))
    out["SQ_INSTS_MFMA (x 32768 FLOP)"] = "%.3e" % m
    return out


print("| | " + " | ".join("`%s`" % k for k in kernels) + " |")
print("|---|" + "---|" * len(kernels))
print("| dispatches, avg µs under the profiler | " + " | ".join("%d, %.0f" % (len(dur[k]), sum(dur[k]) / len(dur[k])) for k in kernels) + " |")
d = {k: derived(k) for k in kernels}
for name in d[kernels[0]]:
    print("| %s | " % name + " | ".join(d[k][name] for k in kernels) + " |")
