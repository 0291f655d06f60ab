cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_fused_step_gpu.py tests/test_model_gpu.py tests/test_zz_fused_f8_gpu.py -m gpu -q -x -p no:cacheprovider --timeout=900 -k "clipped or ladder or full_7b or f8" > gpurun_out/s1_pytest.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/s1_pytest.log
timeout 900 python bench.py --steps 20 --warmup 8 > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err; echo "bench exit $?"; cat gpurun_out/s1_bench.json; tail -3 gpurun_out/s1_bench.err
