for W in 2 4 6 8 12; do
  if [ $W = 4 ]; then L=lit_llama_amd/libmi355llama.so; else L=lit_llama_amd/_variants/libmi355llama_w$W.so; fi
  echo "== window $W"
  MI355_LLAMA_LIB=$PWD/$L timeout 100 python scripts/fused_debug.py --layers 32 --steps 24 2>&1 | grep -E "fused:|equal"
done
