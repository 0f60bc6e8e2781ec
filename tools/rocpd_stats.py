#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table (name, calls, total/avg/min/max us, %).
usage: tools/rocpd_stats.py <results.db> [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namec}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
    for k, (c, t, mn, mx) in out:
        w.writerow([k, c, f"{t:.1f}", f"{t / c:.1f}", f"{mn:.1f}", f"{mx:.1f}", f"{100 * t / tot:.2f}"])


if __name__ == "__main__":
    main()
