#!/usr/bin/env python3
"""Per-kernel HBM traffic from two separate rocprofv3 --pmc passes over the same bench command (FETCH_SIZE, WRITE_SIZE; KiB per
dispatch summed over the XCDs).  bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- the gfx950 FETCH_SIZE x2
correction of MI355X_MICROARCH.md -- averaged over the launches of the profiled run.
usage: tools/hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps in the run> <out.json> [by_grid.txt]
The output carries "_build" = bench.build_stamp() of the tree it was collected on; bench.py refuses a file whose stamp differs from
the build it runs (roofline.traffic is then null).

collect with (separately, no other tracing):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <dir> -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline"""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<.*?>)?)\(", name)
    return (m.group(1) if m else name)[:110]


def collect(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def by_grid(fpath, wpath, out_path, pattern="wino_sp_kernel"):
    """per launch shape (grid size = pixel tiles x channel tiles) of the kernels matching `pattern`: launches, FETCH (x2) and WRITE MB --
    tells which layers make a kernel's per-launch average"""
    rows = {}
    for path, counter, slot in ((fpath, "FETCH_SIZE", 0), (wpath, "WRITE_SIZE", 1)):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and pattern in r["Kernel_Name"]:
                key = (short(r["Kernel_Name"]), int(r.get("Grid_Size") or 0) // max(int(r.get("Workgroup_Size") or 1), 1))
                rows.setdefault(key, [[], []])[slot].append(float(r["Counter_Value"]))
    with open(out_path, "w") as o:
        o.write("# kernel, workgroups per launch, launches in the pass, FETCH_SIZE x2 MB / launch, WRITE_SIZE MB / launch\n")
        for (k, g), (fv, wv) in sorted(rows.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
            fa = 2.0 * 1024 * sum(fv) / max(len(fv), 1) / 1e6
            wa = 1024 * sum(wv) / max(len(wv), 1) / 1e6
            o.write("%-44s %7d wg  n=%3d  fetch %8.1f MB  write %8.1f MB\n" % (k[:44], g, max(len(fv), len(wv)), fa, wa))


def main():
    if len(sys.argv) > 5:
        by_grid(sys.argv[1], sys.argv[2], sys.argv[5])
    f, w = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    steps = int(sys.argv[3])
    out = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) + sum(w.get(k, [0])))):
        fv, wv = f.get(k, [0.0]), w.get(k, [0.0])
        fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
        out[k] = {"launches_per_step": max(len(fv), len(wv)) / steps, "fetch_size_kb_avg": fa, "write_size_kb_avg": wa,
                  "hbm_bytes_per_launch_corrected": (2.0 * fa + wa) * 1024.0}
    import bench
    final = {"_build": bench.build_stamp()}
    final.update(out)
    json.dump(final, open(sys.argv[4], "w"), indent=1)
    for k in list(out)[:12]:
        print("%-70s %8.1f MB/launch" % (k[:70], out[k]["hbm_bytes_per_launch_corrected"] / 1e6))


if __name__ == "__main__":
    main()
