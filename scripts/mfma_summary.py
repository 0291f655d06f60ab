#!/usr/bin/env python
"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.

SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs
(/opt/skills/guides/MI355X_MICROARCH.md: 32 cycles per v_mfma_f32_32x32x16_bf16; measured here: exactly 16 per
v_mfma_f32_16x16x32 — 360 710 144 for the 22 544 384 MFMAs of the c_fc1/c_fc2 GEMM at T = 2048).  GRBM_GUI_ACTIVE comes
back SUMMED OVER THE 8 XCDs (5.36 M cycles for a 355-us kernel = 8 x 670 k at the 1.9 GHz a profiled pass runs at), so
MfmaUtil = busy / (active / 8 x 1024 SIMDs).  (ROCm 7.2 ships no gfx950 section in derived_counters.xml; the gfx94x
formula without the / 8 gives 8x too little.)  Utilisation is per ACTUAL cycle: the fraction of the 2.5 PFLOP/s peak
(quoted at 2.4 GHz) is lower by the clock ratio.
    python scripts/mfma_summary.py <counter_collection.csv>
"""
import csv
import sys
from collections import defaultdict


def main():
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(sys.argv[1]) as f:
        rd = csv.DictReader(f)
        cols = rd.fieldnames
        name_c = next(c for c in cols if c.lower() in ("kernel_name", "kernel name"))
        cn_c = next(c for c in cols if c.lower() in ("counter_name", "counter name"))
        cv_c = next(c for c in cols if c.lower() in ("counter_value", "counter value"))
        for r in rd:
            n = r[name_c].replace("(anonymous namespace)::", "").replace("void ", "")
            n = n[: n.find("(")] if "(" in n else n
            a = agg[n][r[cn_c]]
            a[0] += 1
            a[1] += float(r[cv_c])
    rows = []
    for k, c in agg.items():
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0])
        act = c.get("GRBM_GUI_ACTIVE", [0, 0.0])
        if act[0] == 0 or busy[1] == 0:
            continue
        rows.append((busy[1], k, busy[0], busy[1] / busy[0], act[1] / act[0]))
    print(f"{'kernel':64s} {'launches':>8s} {'MFMA busy cyc/launch':>22s} {'active cyc x 8 XCDs':>18s} {'MfmaUtil':>9s}")
    for _, k, n, b, a in sorted(rows, reverse=True)[:12]:
        print(f"{k[:64]:64s} {n:8d} {b:22.0f} {a:18.0f} {b / (a / 8 * 1024):9.3f}")


if __name__ == "__main__":
    main()
