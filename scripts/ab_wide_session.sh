#!/bin/bash
# One gpurun call: A / B of the wide-shape persistent step over the default library and every lit_llama_amd/_variants/*.so on ONE box.
#   gpurun --timeout 1500 -- 'bash scripts/ab_wide_session.sh [rounds N] [layers L]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
ROUNDS=2; LAYERS=16
while [ $# -gt 0 ]; do
  case $1 in
    rounds) ROUNDS=$2; shift ;;
    layers) LAYERS=$2; shift ;;
  esac
  shift
done
: > $OUT/ab_wide.log
for r in $(seq 1 $ROUNDS); do
  echo "== default (round $r)" | tee -a $OUT/ab_wide.log
  timeout 300 python scripts/ab_wide.py --tag default --layers $LAYERS $( [ $r -gt 1 ] && echo --no-parity ) 2>&1 | grep -E "^AB|Error|error|abort" | tee -a $OUT/ab_wide.log
  for f in lit_llama_amd/_variants/*.so; do
    [ -e "$f" ] || continue
    t=$(basename $f .so); t=${t#libmi355llama_}
    echo "== $t (round $r)" | tee -a $OUT/ab_wide.log
    MI355_LLAMA_LIB=$PWD/$f timeout 300 python scripts/ab_wide.py --tag $t --layers $LAYERS $( [ $r -gt 1 ] && echo --no-parity ) 2>&1 | grep -E "^AB|Error|error|abort" | tee -a $OUT/ab_wide.log
  done
done
echo "=== done $(date +%T)" | tee -a $OUT/ab_wide.log
