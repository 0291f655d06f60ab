cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/ab.log
echo "== default" | tee -a gpurun_out/ab.log
timeout 300 python scripts/ab_fused.py --tag default --timeline 2>&1 | grep -E "^AB|timeline|^  [GS] |Error|error|abort" | tee -a gpurun_out/ab.log
for t in noconv halfmfma; do
  echo "== $t (timing proxy, wrong numerics)" | tee -a gpurun_out/ab.log
  MI355_LLAMA_LIB=$PWD/lit_llama_amd/_variants/libmi355llama_$t.so timeout 300 python scripts/ab_fused.py --tag $t --timeline --no-parity 2>&1 | grep -E "^AB|timeline|^  [GS] |Error|error|abort" | tee -a gpurun_out/ab.log
done
