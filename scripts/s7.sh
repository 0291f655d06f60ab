cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_step_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -k "bf16 or int8" > gpurun_out/s7_new_tests.log 2>&1; echo "bf16/int8 tests exit $?"; grep -E "passed|failed|Error|assert|AssertionError" gpurun_out/s7_new_tests.log | tail -15
timeout 400 python bench.py --quantize none --steps 64 --no-cpu-baseline --no-tp > gpurun_out/s7_bench_none.json 2> gpurun_out/s7_bench.err; echo "bench none exit $?"; tail -1 gpurun_out/s7_bench_none.json | cut -c1-260
timeout 400 python bench.py --quantize llm.int8 --steps 64 --no-cpu-baseline --no-tp > gpurun_out/s7_bench_int8.json 2>> gpurun_out/s7_bench.err; echo "bench int8 exit $?"; tail -1 gpurun_out/s7_bench_int8.json | cut -c1-260
MI355_FUSED_INT8=0 timeout 400 python bench.py --quantize llm.int8 --steps 64 --no-cpu-baseline --no-tp > gpurun_out/s7_bench_int8_launch.json 2>> gpurun_out/s7_bench.err; tail -1 gpurun_out/s7_bench_int8_launch.json | cut -c1-260
tail -5 gpurun_out/s7_bench.err
