#!/bin/bash
# One gpurun call: the int4 headline through both operand paths of the persistent step on ONE box (bench.py lines), then the rocprofv3
# kernel trace of the fp8-operand run (MI355_FUSED_F8=1).  Output: gpurun_out/bench_f8.json, bench_f16.json, prof_f8_summary.txt.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
MI355_FUSED_F8=1 timeout 50 python bench.py --steps 128 --no-cpu-baseline --no-tp > $OUT/bench_f8.json 2> $OUT/bench_f8.err
echo "f8 rc $?"; cut -c1-400 $OUT/bench_f8.json
timeout 50 python bench.py --steps 128 --no-cpu-baseline --no-tp > $OUT/bench_f16.json 2> $OUT/bench_f16.err
echo "f16 rc $?"; cut -c1-400 $OUT/bench_f16.json
rm -rf $OUT/prof_f8
MI355_FUSED_F8=1 timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f8 -o run -- python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-tp > $OUT/prof_f8_bench.json 2> $OUT/prof_f8.err
echo "rocprof rc $?"
t=$(find $OUT/prof_f8 -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && python scripts/prof_summary.py "$t" > $OUT/prof_f8_summary.txt 2>&1 && head -20 $OUT/prof_f8_summary.txt
f=$(find $OUT/prof_f8 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/prof_f8_kernel_stats.csv
rm -rf $OUT/prof_f8
