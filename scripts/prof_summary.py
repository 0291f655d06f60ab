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
    # steady-state decode: the last complete token (from one embedding_kernel to the next)
    idx = [i for i, r in enumerate(rows) if "embedding_kernel" in r[0]]
    if len(idx) >= 3:
        a, b = idx[-3], idx[-2]
        tok = rows[a:b]
        span = tok[-1][2] - tok[0][1]
        busy = sum(e - s for _, s, e in tok)
        print(f"\nlast full decode step: {len(tok)} launches, span {span / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
        per = defaultdict(lambda: [0, 0])
        for n, s, e in tok:
            per[short(n)][0] += 1
            per[short(n)][1] += e - s
        for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            print(f"  {k:78s} {c:4d} x {t / c / 1e3:7.2f} us = {t / 1e3:8.1f} us")


if __name__ == "__main__":
    main()
