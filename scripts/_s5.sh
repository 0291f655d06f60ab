cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fused_wide_gpu.py -m gpu -q -x -s -p no:cacheprovider --timeout=600 -k "7b_shape or launch_path" > gpurun_out/s5_pytest.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/s5_pytest.log
timeout -k 10 600 python scripts/fused_timeline.py --heads 64 --layers 8 --layer 4 --prompt 128 > gpurun_out/s5_timeline_65b.txt 2>&1; echo "timeline exit $?"; grep -v amdgpu gpurun_out/s5_timeline_65b.txt | grep -v "^  [GS] " | head -40
timeout -k 10 900 python bench.py --model 65B --steps 32 --no-cpu-baseline --no-tp > gpurun_out/s5_bench_65B.json 2> gpurun_out/s5_bench_65B.err; echo "bench65 exit $?"; cut -c1-330 gpurun_out/s5_bench_65B.json; tail -3 gpurun_out/s5_bench_65B.err
