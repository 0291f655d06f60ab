cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_adapter_gpu.py -m gpu -q -p no:cacheprovider --timeout=500 -k "v2_on_llm_int8 or in_place" > gpurun_out/s11_adapter.log 2>&1; echo "adapter tests exit $?"; grep -E "passed|failed|^E " gpurun_out/s11_adapter.log | tail -8
bash scripts/gpu_session.sh bench prof pmc cfgs 2>&1 | tail -40
timeout 300 python scripts/fused_timeline.py > gpurun_out/fused_timeline.txt 2>&1; grep -v amdgpu gpurun_out/fused_timeline.txt | head -30
timeout 300 python bench.py --quantize none --steps 64 --no-cpu-baseline --no-tp --no-graph > /dev/null 2>&1
