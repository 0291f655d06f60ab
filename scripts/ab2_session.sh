#!/bin/bash
# One gpurun call: selected parity tests of the default build, then ROUNDS alternating rounds of scripts/ab_fused.py over the default
# library and every lit_llama_amd/_variants/*.so (timeline + budget in round 1 only).
#   gpurun --timeout 1200 -- 'bash scripts/ab2_session.sh [ROUNDS] [pytest -k expression]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
ROUNDS=${1:-2}
KEXPR=${2:-}
: > $OUT/ab.log
if [ -n "$KEXPR" ]; then
  timeout 600 python -m pytest tests/test_zz_fused_f8_gpu.py tests/test_fused_step_gpu.py -m gpu -q -x -p no:cacheprovider --timeout=600 -s -k "$KEXPR" > $OUT/ab_tests.log 2>&1
  echo "pytest exit $?" | tee -a $OUT/ab.log; grep -E "passed|failed|error|Error|assert|fmt|launch" $OUT/ab_tests.log | tail -12 | tee -a $OUT/ab.log
fi
for r in $(seq 1 $ROUNDS); do
  for f in default lit_llama_amd/_variants/*.so; do
    if [ "$f" = default ]; then t=default; unset MI355_LLAMA_LIB; else [ -e "$f" ] || continue; t=$(basename $f .so); t=${t#libmi355llama_}; export MI355_LLAMA_LIB=$PWD/$f; fi
    echo "== $t (round $r)" | tee -a $OUT/ab.log
    timeout 300 python scripts/ab_fused.py --tag $t $( [ $r -eq 1 ] && echo --timeline || echo --no-parity ) 2>&1 | grep -E "^AB|timeline|^  [GS] |^  layer period|^    |Error|error|abort" | tee -a $OUT/ab.log
  done
done
unset MI355_LLAMA_LIB
echo "=== done $(date +%T)" | tee -a $OUT/ab.log
