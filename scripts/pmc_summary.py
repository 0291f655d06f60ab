#!/usr/bin/env python
"""Per-kernel HBM read traffic from a `rocprofv3 --pmc FETCH_SIZE` pass (counter_collection CSV).

FETCH_SIZE is reported in KiB of 64-B read requests at the L2's memory side; on gfx950 a wide coalesced streaming
read (16 B per lane) is tallied at HALF its bytes (/opt/skills/guides/MI355X_MICROARCH.md, section HBM), so the
figure is doubled for the weight-streaming kernels, as the guide prescribes.  Writes the dominant kernel's
bytes per launch to profiles/pmc_traffic.json (read back by bench.py as roofline.traffic).

    python scripts/pmc_summary.py <counter_collection.csv> [--json profiles/pmc_traffic.json]
"""
import csv
import json
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        rd = csv.DictReader(f)
        cols = rd.fieldnames
        name_c = next(c for c in cols if c.lower() in ("kernel_name", "kernel name"))
        cn_c = next(c for c in cols if c.lower() in ("counter_name", "counter name"))
        cv_c = next(c for c in cols if c.lower() in ("counter_value", "counter value"))
        for r in rd:
            if r[cn_c] != "FETCH_SIZE":
                continue
            n = r[name_c].replace("(anonymous namespace)::", "").replace("void ", "")
            n = n[: n.find("(")] if "(" in n else n
            a = agg[n]
            a[0] += 1
            a[1] += float(r[cv_c])
    print(f"{'kernel':70s} {'launches':>8s} {'FETCH_SIZE KiB/launch':>22s} {'x2 corrected MB':>16s}")
    res = {}
    for k, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        per = v / c
        print(f"{k[:70]:70s} {c:8d} {per:22.1f} {2 * per * 1024 / 1e6:16.2f}")
        res[k] = {"launches": c, "fetch_kib_per_launch": per, "corrected_bytes_per_launch": 2 * per * 1024}
    # the c_fc1/c_fc2 + SwiGLU launch: gemv_kernel<FMT = 0 (Q4), R = 2, P, EPI = 2, VMODE, MULTI>; prefer the decode
    # specialisation (MULTI = false) over the prompt-chunk one
    def targs(k):
        return [t.strip() for t in k[k.find("<") + 1:k.rfind(">")].split(",")]

    cands = [(k, v) for k, v in res.items() if k.startswith("gemv_kernel<") and targs(k)[:2] == ["0", "2"] and targs(k)[3] == "2"]
    cands.sort(key=lambda kv: (targs(kv[0])[5:6] != ["false"], -kv[1]["launches"]))
    fc = cands[0][1] if cands else None
    fused = next((v for k, v in res.items() if k.startswith("fused_step")), None)  # fused_step_ring_kernel<GRP, FMT>: the busiest instantiation
    if out_json and (fc or fused):
        out = {"note": "rocprofv3 --pmc FETCH_SIZE, KiB x 1024 x 2 (gfx950 wide-read correction)", "kernels": res}
        if "--merge" in sys.argv:
            # keep the entries of kernels this pass did not launch (bench.py looks the TIMED instantiation up by name: a pass over
            # MI355_FUSED_F8=0 and one over the default each contribute their fused_step_ring_kernel<false, FMT>)
            try:
                old = json.load(open(out_json))
                out["kernels"] = {**old.get("kernels", {}), **res}
                for k_ in ("fc_swiglu_bytes_per_launch", "fused_step_bytes_per_launch"):
                    if k_ in old:
                        out[k_] = old[k_]
            except (OSError, ValueError):
                pass
        if fc:
            out["fc_swiglu_bytes_per_launch"] = round(fc["corrected_bytes_per_launch"])
        if fused:  # one launch = one decode step
            out["fused_step_bytes_per_launch"] = round(fused["corrected_bytes_per_launch"])
        json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
