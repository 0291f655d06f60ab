#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace run.

    python scripts/prof_summary.py <run_results.db | *_kernel_trace.csv> [--tail-tokens N]
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    db = sqlite3.connect(path)
    return [(n, s, e) for n, s, e in db.execute("select name, start, end from kernels order by start")]


def rows_from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out.sort(key=lambda r: r[1])
    return out


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:80]


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = defaultdict(lambda: [0, 0])
    for n, s, e in rows:
        a = agg[short(n)]
        a[0] += 1
        a[1] += e - s
    total = sum(a[1] for a in agg.values())
    print(f"{'kernel':80s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>8s} {'share':>6s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{k:80s} {c:7d} {t / 1e6:9.3f} {t / c / 1e3:8.2f} {100 * t / total:5.1f}%")
    # steady-state decode: chained greedy steps, each ending with argmax_advance_kernel.  Averages over the last
    # (up to) 32 complete steps, M = 1 launches only (the per-kernel table above also contains the prompt chunks).
    fused = [r for r in rows if "fused_step" in r[0]]  # fused_step_ring_kernel (default) / fused_step_kernel (LDS-DMA)
    if len(fused) >= 8:
        # the chained loop: consecutive launches less than 30 us apart (the bench also launches the step one at a time,
        # with host synchronisation in between, for its roofline object)
        pairs = [(fused[i], fused[i + 1][1] - fused[i][2]) for i in range(len(fused) - 1)]
        chained = [(r, g) for r, g in pairs if g < 30e3]
        if chained:
            dur = sum(e - s_ for (_, s_, e), _ in chained) / len(chained)
            gap = sum(g for _, g in chained) / len(chained)
            print(f"\nsteady-state decode on the fused step: 1 launch per token; {len(chained)} chained launches, mean duration "
                  f"{dur / 1e3:.1f} us, mean gap to the next launch {gap / 1e3:.2f} us -> {1e9 / (dur + gap):.1f} tokens/s")
    ends = [i for i, r in enumerate(rows) if "argmax_advance_kernel" in r[0]]
    if len(ends) >= 3:
        ends = ends[-33:]
        steps = [rows[ends[i] + 1:ends[i + 1] + 1] for i in range(len(ends) - 1)]
        n = len(steps)
        span = sum(st[-1][2] - st[0][1] for st in steps) / n
        busy = sum(sum(e - s_ for _, s_, e in st) for st in steps) / n
        print(f"\nsteady-state decode, mean of the last {n} chained steps: {len(steps[-1])} launches per step, "
              f"span {span / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
        per = defaultdict(lambda: [0, 0])
        for st in steps:
            for nme, s_, e in st:
                per[short(nme)][0] += 1
                per[short(nme)][1] += e - s_
        for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            print(f"  {k:78s} {c / n:6.1f} x {t / c / 1e3:7.2f} us = {t / n / 1e3:8.1f} us per step")


if __name__ == "__main__":
    main()
