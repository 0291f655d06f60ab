#!/bin/bash
# One gpurun call at the end of a session: kernel + prefill parity tests, then same-box A / B of the knobs of this session.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_kernels_gpu.py tests/test_prefill_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/final_tests.log 2>&1
echo "pytest rc $?"; tail -2 gpurun_out/final_tests.log
V=$PWD/lit_llama_amd/_variants
for t in 2048 128; do
  echo -n "prefill $t default: "; timeout 100 python scripts/prefill_run.py $t 2>&1 | tail -1
  echo -n "prefill $t v_and+v_or: "; MI355_LLAMA_LIB=$V/libmi355llama_noandor.so timeout 100 python scripts/prefill_run.py $t 2>&1 | tail -1
done
echo -n "13B default: "; timeout 200 python bench.py --model 13B --steps 64 --no-cpu-baseline --no-tp 2>/dev/null | tail -1 | cut -c1-160
echo -n "13B gemv v_and+v_or: "; MI355_LLAMA_LIB=$V/libmi355llama_gemvnoandor.so timeout 200 python bench.py --model 13B --steps 64 --no-cpu-baseline --no-tp 2>/dev/null | tail -1 | cut -c1-160
