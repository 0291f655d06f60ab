cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_step_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -s -k "int8 or bf16" > gpurun_out/s18_tests.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|int8 outliers" gpurun_out/s18_tests.log | tail -5
timeout 400 python bench.py --quantize llm.int8 --steps 64 --no-cpu-baseline --no-tp > gpurun_out/bench_cfg_llm.int8.json 2> gpurun_out/s18.err; tail -1 gpurun_out/bench_cfg_llm.int8.json | cut -c1-200
python scripts/fused_timeline.py --quantize llm.int8 2>&1 | grep -v amdgpu | head -23 > gpurun_out/tl_int8.txt; cat gpurun_out/tl_int8.txt
