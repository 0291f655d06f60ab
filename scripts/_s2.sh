cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fused_wide_gpu.py -m gpu -q -x -s -p no:cacheprovider --timeout=600 -k "7b_shape or refuses" > gpurun_out/s2_pytest_a.log 2>&1; echo "pytest A exit $?"; tail -25 gpurun_out/s2_pytest_a.log
timeout -k 10 1200 python -m pytest tests/test_fused_wide_gpu.py -m gpu -q -x -s -p no:cacheprovider --timeout=900 -k "65b_width_against_launch" > gpurun_out/s2_pytest_b.log 2>&1; echo "pytest B exit $?"; tail -25 gpurun_out/s2_pytest_b.log
