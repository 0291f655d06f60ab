#!/bin/bash
# One gpurun call = tests + smoke + bench + rocprof summary + tuning sweep; everything lands in gpurun_out/.
# Usage (from the build container):  gpurun --timeout 2400 -- 'bash scripts/gpu_session.sh [stage ...]'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
STAGES="${*:-tests smoke bench prof sweep}"
echo "stages: $STAGES" | tee $OUT/session.log
rocm-smi --showproductname 2>/dev/null | head -8 >> $OUT/session.log
for st in $STAGES; do
  echo "=== $st $(date +%T)" | tee -a $OUT/session.log
  case $st in
    tests)
      timeout 2700 python -m pytest tests -m gpu -q -rA -p no:cacheprovider --timeout=900 > $OUT/pytest_gpu.log 2>&1
      echo "pytest exit $?" | tee -a $OUT/session.log; tail -60 $OUT/pytest_gpu.log ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
      echo "smoke exit $?" | tee -a $OUT/session.log; tail -5 $OUT/smoke.log ;;
    bench)
      timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
      echo "bench exit $?" | tee -a $OUT/session.log; cat $OUT/bench.json; tail -5 $OUT/bench.err
      timeout 600 python bench.py --no-graph --no-cpu-baseline --no-tp --steps 64 > $OUT/bench_nograph.json 2>> $OUT/bench.err
      cat $OUT/bench_nograph.json ;;
    prof)
      rm -rf $OUT/prof
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-tp > $OUT/prof_bench.json 2> $OUT/prof.err
      echo "rocprof exit $?" | tee -a $OUT/session.log
      ls $OUT/prof | head
      f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
      t=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
      [ -n "$t" ] && python scripts/prof_summary.py "$t" > $OUT/prof_summary.txt 2>&1 && cat $OUT/prof_summary.txt
      find $OUT/prof -name '*kernel_trace.csv' -size +30M -delete ;;
    pmc)
      rm -rf $OUT/pmc
      timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o fetch -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-tp > $OUT/pmc_bench.json 2> $OUT/pmc.err
      echo "pmc exit $?" | tee -a $OUT/session.log
      f=$(find $OUT/pmc -name '*counter_collection.csv' | head -1)
      ls $OUT/pmc | head; [ -n "$f" ] && head -3 "$f"
      [ -n "$f" ] && python scripts/pmc_summary.py "$f" --json $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
      find $OUT/pmc -name '*.csv' -size +20M -delete ;;
    cfgs)
      # the other BASELINE configurations and the long-context point (profiles/rNN_bench_cfg_*.json, rNN_bench_longctx.json)
      timeout 400 python bench.py --quantize none --steps 64 --no-cpu-baseline --no-tp > $OUT/bench_cfg_none.json 2>> $OUT/bench.err
      timeout 400 python bench.py --quantize llm.int8 --steps 64 --no-cpu-baseline --no-tp > $OUT/bench_cfg_llm.int8.json 2>> $OUT/bench.err
      timeout 500 python bench.py --model 13B --steps 64 --no-cpu-baseline --no-tp > $OUT/bench_cfg_13B.json 2>> $OUT/bench.err
      timeout 700 python bench.py --model 30B --steps 48 --no-cpu-baseline --no-tp > $OUT/bench_cfg_30B.json 2>> $OUT/bench.err
      timeout 900 python bench.py --model 65B --steps 32 --no-cpu-baseline --no-tp > $OUT/bench_cfg_65B.json 2>> $OUT/bench.err
      timeout 400 python bench.py --prompt-len 1900 --steps 64 --warmup 16 --no-cpu-baseline --no-tp > $OUT/bench_longctx.json 2>> $OUT/bench.err
      for f in none llm.int8 13B 30B 65B; do tail -1 $OUT/bench_cfg_$f.json | cut -c1-200; done; tail -1 $OUT/bench_longctx.json | cut -c1-200
      echo "cfgs done" | tee -a $OUT/session.log ;;
    mfma)
      rm -rf $OUT/mfma
      timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o mfma -- python scripts/prefill_run.py 2048 > $OUT/mfma_run.txt 2> $OUT/mfma.err
      echo "mfma pmc exit $?" | tee -a $OUT/session.log
      f=$(find $OUT/mfma -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python scripts/mfma_summary.py "$f" > $OUT/mfma_summary.txt 2>&1; cat $OUT/mfma_summary.txt; tail -3 $OUT/mfma_run.txt
      find $OUT/mfma -name '*.csv' -size +20M -delete ;;
    sweep)
      timeout 1200 python scripts/sweep_gemv.py --out $OUT/sweep_best.json > $OUT/sweep.log 2>&1
      echo "sweep exit $?" | tee -a $OUT/session.log; grep BEST $OUT/sweep.log ;;
    sweepres)
      timeout 600 python scripts/sweep_gemv.py --quick --bufs 1 > $OUT/sweep_resident.log 2>&1
      echo "sweep(resident) exit $?" | tee -a $OUT/session.log; grep BEST $OUT/sweep_resident.log ;;
    sweepq)
      timeout 600 python scripts/sweep_gemv.py --quick --out $OUT/sweep_best.json > $OUT/sweep.log 2>&1
      echo "sweep exit $?" | tee -a $OUT/session.log; grep BEST $OUT/sweep.log ;;
  esac
done
echo "=== done $(date +%T)" | tee -a $OUT/session.log
