#!/bin/bash
# One gpurun call: A / B of the persistent step's other stream formats (BASELINE configs[1] bf16, configs[3] llm.int8) over the default
# library and every lit_llama_amd/_variants/*.so on ONE box, through bench.py (three blocks of 64 steps each).
#   gpurun --timeout 1500 -- 'bash scripts/ab_formats_session.sh [rounds N]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
ROUNDS=2
[ "$1" = "rounds" ] && ROUNDS=$2
: > $OUT/ab_formats.log
one() {  # tag, quantize, lib
  ( [ -n "$3" ] && export MI355_LLAMA_LIB=$3; timeout 400 python bench.py --quantize $2 --steps 64 --no-cpu-baseline --no-tp 2>/dev/null ) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('AB', '$1', '$2', d['value'], 'tok/s', d['blocks_ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'])" | tee -a $OUT/ab_formats.log
}
for r in $(seq 1 $ROUNDS); do
  for q in none llm.int8; do
    one default $q ""
    for f in lit_llama_amd/_variants/*.so; do
      [ -e "$f" ] || continue
      t=$(basename $f .so); t=${t#libmi355llama_}
      one $t $q $PWD/$f
    done
  done
done
echo "=== done $(date +%T)" | tee -a $OUT/ab_formats.log
