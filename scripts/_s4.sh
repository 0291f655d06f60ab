cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 600 python scripts/fused_timeline.py --heads 64 --layers 8 --layer 4 --prompt 128 > gpurun_out/s4_timeline_65b.txt 2>&1; echo "timeline exit $?"; grep -v amdgpu gpurun_out/s4_timeline_65b.txt
