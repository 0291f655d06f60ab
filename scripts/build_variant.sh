#!/bin/bash
# Build lit_llama_amd/_variants/libmi355llama_<tag>.so with extra defines for csrc/fused_step.hip (A / B of tuning knobs on
# one box: MI355_LLAMA_LIB=<path> selects the library).   bash scripts/build_variant.sh w8 -DMI355_FUSED_WRUN=8 ...
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p lit_llama_amd/_variants /tmp/variants
obj=/tmp/variants/fused_step_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c lit_llama_amd/csrc/fused_step.hip -o $obj
objs=$(ls lit_llama_amd/csrc/_obj/*.o | grep -v "/fused_step.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lit_llama_amd/_variants/libmi355llama_$tag.so $objs $obj
echo built lit_llama_amd/_variants/libmi355llama_$tag.so
