#!/bin/bash
# Launch-shape sweep of the wide GEMM for short prompts: every (waves, tokens per block, K-slices) the kernel is instantiated
# for, per layer linear (c_attn, attn.c_proj, c_fc1/c_fc2 pair, mlp.c_proj: us per launch incl. the staging pass), through
# scripts/bench_gemm.py (MI355_GEMM_FORCE overrides the rule in csrc/gemm.hip).
#   gpurun -- 'bash scripts/sweep_gemm_shapes.sh 128 384'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
us() { grep -oE ": +[0-9.]+ us" | grep -oE "[0-9.]+" | tr '\n' ' '; echo; }
for M in "$@"; do
  echo -n "M=$M rule     "; timeout 60 python scripts/bench_gemm.py --M $M 2>&1 | us
  for shape in 8:128 8:64 2:128; do
    for ks in 1 2 4 8; do
      echo -n "M=$M $shape:$ks  "
      MI355_GEMM_FORCE=$shape:$ks timeout 60 python scripts/bench_gemm.py --M $M 2>&1 | us
    done
  done
done
