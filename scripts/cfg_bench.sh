#!/bin/bash
# The bench lines of the two other 7B BASELINE configurations on the persistent step (profiles/rNN_bench_cfg_none.json, _llm.int8.json).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for q in none llm.int8; do
  timeout 400 python bench.py --quantize $q --steps 64 --no-cpu-baseline --no-tp > gpurun_out/bench_cfg_$q.json 2>> gpurun_out/bench.err
  tail -1 gpurun_out/bench_cfg_$q.json | cut -c1-160
done
