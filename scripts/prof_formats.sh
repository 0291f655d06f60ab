#!/bin/bash
# rocprofv3 evidence for the persistent step over BF16 / LLM.int8 streams: kernel trace + stats, then a separate FETCH_SIZE pass.
#   gpurun --timeout 1500 -- 'bash scripts/prof_formats.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; OUT=gpurun_out; mkdir -p $OUT
for q in none llm.int8; do
  rm -rf $OUT/prof_$q $OUT/pmc_$q
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$q -o run -- python bench.py --quantize $q --steps 64 --warmup 8 --no-cpu-baseline --no-tp > $OUT/prof_bench_$q.json 2> $OUT/prof_$q.err
  t=$(find $OUT/prof_$q -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && python scripts/prof_summary.py "$t" > $OUT/prof_summary_$q.txt 2>&1; tail -4 $OUT/prof_summary_$q.txt
  find $OUT/prof_$q -name '*kernel_trace.csv' -size +30M -delete
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$q -o fetch -- python bench.py --quantize $q --steps 8 --warmup 2 --no-cpu-baseline --no-tp > $OUT/pmc_bench_$q.json 2> $OUT/pmc_$q.err
  f=$(find $OUT/pmc_$q -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python scripts/pmc_summary.py "$f" > $OUT/pmc_summary_$q.txt 2>&1; head -4 $OUT/pmc_summary_$q.txt
  find $OUT/pmc_$q -name '*.csv' -size +20M -delete
done
