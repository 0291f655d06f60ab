#!/bin/bash
# One gpurun call: A / B of the persistent decode step over the default library and every lit_llama_amd/_variants/*.so on
# ONE box (scripts/ab_fused.py per library), optionally preceded by parity tests of the default build.
#   gpurun --timeout 1500 -- 'bash scripts/ab_session.sh [tests] [golden] [timeline] [rounds N]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
TESTS=""; TL=""; ROUNDS=1
while [ $# -gt 0 ]; do
  case $1 in
    tests) TESTS="$TESTS tests/test_fused_step_gpu.py" ;;
    golden) TESTS="$TESTS tests/test_golden_7b_gpu.py" ;;
    f8tests) TESTS="$TESTS tests/test_zz_fused_f8_gpu.py" ;;
    real) TESTS="$TESTS tests/test_zz_golden_7b_real_gpu.py" ;;
    timeline) TL="--timeline" ;;
    rounds) ROUNDS=$2; shift ;;
  esac
  shift
done
: > $OUT/ab.log
if [ -n "$TESTS" ]; then
  timeout 1200 python -m pytest $TESTS -m gpu -q -x -p no:cacheprovider --timeout=900 -s > $OUT/ab_tests.log 2>&1
  echo "pytest exit $?" | tee -a $OUT/ab.log; grep -E "passed|failed|error|Error|assert|fused|launch" $OUT/ab_tests.log | tail -12 | tee -a $OUT/ab.log
fi
for r in $(seq 1 $ROUNDS); do
  echo "== default (round $r)" | tee -a $OUT/ab.log
  timeout 300 python scripts/ab_fused.py --tag default $TL 2>&1 | grep -E "^AB|timeline|^  [GS] |^  layer period|^    |Error|error|abort" | tee -a $OUT/ab.log
  for f in lit_llama_amd/_variants/*.so; do
    [ -e "$f" ] || continue
    t=$(basename $f .so); t=${t#libmi355llama_}
    echo "== $t (round $r)" | tee -a $OUT/ab.log
    MI355_LLAMA_LIB=$PWD/$f timeout 300 python scripts/ab_fused.py --tag $t $TL $( [ $r -gt 1 ] && echo --no-parity ) 2>&1 | grep -E "^AB|timeline|^  [GS] |^  layer period|^    |Error|error|abort" | tee -a $OUT/ab.log
  done
done
echo "=== done $(date +%T)" | tee -a $OUT/ab.log
