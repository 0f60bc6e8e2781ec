#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 kernel trace of bench.py (rocpd sqlite): per training step, the busy time of each HIP
queue, the overlap between the main and the weight-gradient stream, idle gaps, and per-kernel durations in a two-stream step
next to the single-stream step of the same run.  usage: tools/timeline.py <results.db>"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:60]


db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id, vgpr_count, accum_vgpr_count, lds_size, grid_x, workgroup_x from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
# a step ends with its last adam launch (two per step when the model has its own sigma group; detect by spacing)
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or rows[adam[i + 1]][1] - rows[adam[i]][2] > 5e6]
steps, prev = [], 0
for e in ends:
    steps.append(rows[prev:e + 1])
    prev = e + 1
print("steps found:", len(steps))


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


for si, st in enumerate(steps):
    t0, t1 = st[0][1], max(r[2] for r in st)
    qs = defaultdict(list)
    for r in st:
        qs[r[3]].append((r[1], r[2]))
    allb = union([(r[1], r[2]) for r in st])
    line = f"step {si}: wall {(t1 - t0) / 1e6:8.2f} ms  busy(any) {allb / 1e6:8.2f}  idle {(t1 - t0 - allb) / 1e6:6.2f}"
    for q, iv in sorted(qs.items()):
        line += f"  q{q}: busy {union(iv) / 1e6:7.2f} sum {sum(e - s for s, e in iv) / 1e6:7.2f} n {len(iv)}"
    if len(qs) == 2:
        a, b = [union(v) for _, v in sorted(qs.items())]
        line += f"  overlap {(a + b - allb) / 1e6:7.2f}"
    print(line)

if len(steps) >= 4:
    two, one = steps[-3], steps[-1]          # last two-stream step, last single-stream step
    def agg(st):
        d = defaultdict(lambda: [0, 0.0])
        for r in st:
            k = short(r[0])
            d[k][0] += 1
            d[k][1] += (r[2] - r[1]) / 1e6
        return d
    a2, a1 = agg(two), agg(one)
    print("\nper kernel: ms in the two-stream step | single-stream step | inflation   (vgpr+agpr, lds)")
    res = {short(r[0]): (r[4], r[5], r[6]) for r in two}
    for k, (n, ms) in sorted(a2.items(), key=lambda kv: -kv[1][1])[:28]:
        m1 = a1.get(k, [0, 0.0])[1]
        print(f"  {k:60s} n={n:4d}  {ms:7.2f} | {m1:7.2f} | {ms / m1 if m1 else 0:5.2f}   v{res[k][0]}+a{res[k][1]} lds {res[k][2]}")
    print(f"  sum {sum(v[1] for v in a2.values()):.2f} | {sum(v[1] for v in a1.values()):.2f}")

# largest idle gaps of the main queue in the last two-stream step: which kernels surround them
if steps:
    st = steps[min(len(steps) - 1, 2)] if len(steps) < 4 else steps[-3]
    mainq = max(set(r[3] for r in st), key=lambda q: sum(1 for r in st if r[3] == q))
    rows_q = sorted([r for r in st if r[3] == mainq], key=lambda r: r[1])
    gaps = []
    for a, b in zip(rows_q[:-1], rows_q[1:]):
        g = b[1] - a[2]
        if g > 0:
            gaps.append((g, short(a[0]), short(b[0])))
    tot = sum(g for g, _, _ in gaps)
    print(f"\nmain queue: {len(gaps)} gaps, {tot / 1e6:.2f} ms in total; > 20 us: {sum(g for g, _, _ in gaps if g > 20e3) / 1e6:.2f} ms")
    agg = defaultdict(lambda: [0, 0.0])
    for g, ka, kb in gaps:
        agg[(ka, kb)][0] += 1
        agg[(ka, kb)][1] += g / 1e3
    for (ka, kb), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {us:9.1f} us in {n:4d} gaps   after {ka[:44]:44s} before {kb[:44]}")

# phases of the last two-stream step on the main queue: forward | backward before the first dense 3x3 data gradient | the conv chain
# (first .. last dense data gradient) | tail; per phase the wall time and the main-queue time by kernel
if steps:
    st = steps[min(len(steps) - 1, 2)] if len(steps) < 4 else steps[-3]
    mainq = max(set(r[3] for r in st), key=lambda q: sum(1 for r in st if r[3] == q))
    rows_q = sorted([r for r in st if r[3] == mainq], key=lambda r: r[1])
    names = [short(r[0]) for r in rows_q]
    def first(pred, default=None):
        for i, n in enumerate(names):
            if pred(n):
                return i
        return default
    bwd0 = first(lambda n: n.startswith(("absdiff_bwd", "diff_bwd", "fft_lines_kernel<2", "blur9_stream_kernel<1", "sqdiff_bwd")), len(names) - 1)
    def is_dg(n):       # dense data gradient with the GroupNorm-backward epilogue: halo <0, S, 3, false, true, ...> / wino <0, true, ...>
        if n.startswith("conv3x3_halo_sp_kernel<0"):
            a = [v.strip() for v in n[n.index("<") + 1:n.rindex(">")].split(",")]
            return len(a) >= 5 and a[4] == "true"
        return n.startswith(("conv3x3_wino_sp_kernel<0, true", "conv3x3_winow_sp_kernel<0, true"))
    dg = [i for i, n in enumerate(names) if is_dg(n)]
    cuts = [0, bwd0, dg[0] if dg else bwd0, dg[-1] + 1 if dg else bwd0, len(names)]
    labels = ["forward", "backward head (losses, blur / FFT backward, decoder tail)", "conv chain (first..last dense data gradient)", "backward tail + Adam"]
    print("\nphases of the main queue (last two-stream step):")
    for (a, b), lab in zip(zip(cuts[:-1], cuts[1:]), labels):
        if b <= a:
            continue
        seg = rows_q[a:b]
        wall = (seg[-1][2] - seg[0][1]) / 1e6
        by = defaultdict(float)
        for r in seg:
            by[short(r[0])[:48]] += (r[2] - r[1]) / 1e6
        top = sorted(by.items(), key=lambda kv: -kv[1])[:(40 if lab == "forward" else 12)]
        print(f"  {lab}: wall {wall:.2f} ms, kernels {sum(by.values()):.2f} ms, n={len(seg)}")
        cnt = defaultdict(int)
        for r in seg:
            cnt[short(r[0])[:48]] += 1
        print("      " + "; ".join(f"{k} {v:.2f} (n={cnt[k]})" for k, v in top))
