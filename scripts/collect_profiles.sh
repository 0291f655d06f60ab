#!/bin/bash
# Copy the summaries of the last `gpu_session.sh` run (gpurun_out/, scratch) into profiles/ (tracked), named per round.
# Usage: bash scripts/collect_profiles.sh r01
set -eu
cd "$(dirname "$0")/.."
R=${1:-r01}
O=gpurun_out
cp $O/pmc_traffic.json profiles/pmc_traffic.json
cp $O/bench.json profiles/${R}_bench.json
cp $O/kernel_stats.csv profiles/${R}_rocprofv3_kernel_stats.csv
cp $O/prof_summary.txt profiles/${R}_rocprofv3_kernel_trace_summary.txt
cp $O/pmc_summary.txt profiles/${R}_rocprofv3_pmc_fetch_size.txt
[ -s $O/int8_timeline.txt ] && grep -v amdgpu $O/int8_timeline.txt > profiles/${R}_int8_phase_timeline.txt
[ -s $O/timeline.txt ] && grep -v amdgpu $O/timeline.txt > profiles/${R}_gemv_phase_timeline.txt
for f in none llm.int8 13B 30B 65B; do [ -s $O/bench_cfg_$f.json ] && cp $O/bench_cfg_$f.json profiles/${R}_bench_cfg_$f.json; done
[ -s $O/bench_longctx.json ] && cp $O/bench_longctx.json profiles/${R}_bench_longctx.json
python - <<PY
import json
for f in ["${R}_bench","${R}_bench_cfg_none","${R}_bench_cfg_llm.int8","${R}_bench_cfg_13B","${R}_bench_cfg_30B","${R}_bench_cfg_65B","${R}_bench_longctx"]:
    try:
        d=json.load(open(f"profiles/{f}.json"))
    except Exception as e:
        print(f, "missing", e); continue
    r=d["decode_roofline"]; fk=[k for k in r if k.startswith("frac_of")][0]
    print(f"{f:28s} {d['value']:8.2f} tok/s {d['ms_per_step']:7.4f} ms  roofline {r['tokens_per_s_at_100pct']:7.1f} frac {r[fk]:.4f}  fc {d['roofline']['avg_launch_us']:6.2f} us {d['roofline']['achieved']:7.1f} GB/s  prefill {d['prefill_s']}")
PY
