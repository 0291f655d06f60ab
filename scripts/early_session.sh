#!/bin/bash
# One gpurun call: the BF16 / LLM.int8 persistent steps with the next phase's ring turn requested in FRONT of the publish barrier
# (MI355_FUSED_EARLY_BURST=1, lit_llama_amd/_variants/libmi355llama_early.so) against the default, two alternating rounds of bench.py.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
V=$PWD/lit_llama_amd/_variants/libmi355llama_early.so
: > $OUT/early.log
MI355_LLAMA_LIB=$V timeout 600 python -m pytest tests/test_fused_step_gpu.py -m gpu -q -x -p no:cacheprovider --timeout=600 -s -k "bf16 or int8" > $OUT/early_tests.log 2>&1
echo "pytest(early) exit $?" | tee -a $OUT/early.log; grep -E "passed|failed|error|Error|assert|std" $OUT/early_tests.log | tail -8 | tee -a $OUT/early.log
for r in 1 2; do
  for q in none llm.int8; do
    for lib in default early; do
      if [ $lib = early ]; then export MI355_LLAMA_LIB=$V; else unset MI355_LLAMA_LIB; fi
      timeout 300 python bench.py --quantize $q --steps 64 --no-cpu-baseline --no-tp > $OUT/early_${q}_${lib}_$r.json 2>> $OUT/early.err
      python - "$OUT/early_${q}_${lib}_$r.json" $q $lib $r <<'PY' | tee -a $OUT/early.log
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["decode_roofline"]; fk = [k for k in r if k.startswith("frac_of")][0]
    print(f"EARLY {sys.argv[2]:9s} {sys.argv[3]:8s} round {sys.argv[4]}: {d['value']:8.2f} tok/s  blocks {d['blocks_ms_per_step']}  {fk} {r[fk]}  kernel {d['roofline']['avg_launch_us']} us")
except Exception as e:
    print("EARLY", sys.argv[2:], "failed", e)
PY
    done
  done
done
unset MI355_LLAMA_LIB
echo "=== done" | tee -a $OUT/early.log
