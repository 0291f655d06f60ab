#!/bin/bash
# A / B of the persistent decode step on one box: the round-2 register-ring kernel vs the LDS-DMA kernel (and any
# compiled variants under lit_llama_amd/_variants/).   gpurun -- 'bash scripts/run_variants.sh [layers] [steps]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
L=${1:-32}; N=${2:-24}
echo "== ring (round 2)"
MI355_FUSED_IMPL=ring timeout 200 python scripts/fused_debug.py --layers $L --steps $N 2>&1 | grep -E "fused:|equal|dlogit|Error|error|abort"
echo "== lds-dma (default build)"
timeout 200 python scripts/fused_debug.py --layers $L --steps $N 2>&1 | grep -E "fused:|equal|dlogit|Error|error|abort"
for f in lit_llama_amd/_variants/*.so; do
  [ -e "$f" ] || continue
  echo "== variant $f"
  MI355_LLAMA_LIB=$PWD/$f timeout 200 python scripts/fused_debug.py --layers $L --steps $N 2>&1 | grep -E "fused:|equal|dlogit|Error|error|abort"
done
