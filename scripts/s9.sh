cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_step_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -s -k "int8" > gpurun_out/s9_int8_tests.log 2>&1; echo "int8 tests exit $?"; grep -E "passed|failed|Error|assert |AssertionError|int8 outliers" gpurun_out/s9_int8_tests.log | tail -15
timeout 400 python bench.py --quantize llm.int8 --steps 64 --no-cpu-baseline --no-tp > gpurun_out/s9_bench_int8.json 2> gpurun_out/s9_bench.err; echo "bench int8 exit $?"; tail -1 gpurun_out/s9_bench_int8.json | cut -c1-260
