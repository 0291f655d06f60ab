#!/bin/bash
# Build lit_llama_amd/_variants/libmi355llama_<tag>.so with extra defines for one source of csrc/ (A / B of tuning knobs on
# one box: MI355_LLAMA_LIB=<path> selects the library).
#   bash scripts/build_variant.sh x [-s fused_step_wide.hip] -DSOME_EXPERIMENT=1 ...     (default source: fused_step_ring.hip; round 6: the
#   kernels carry no standing knobs any more — an experiment adds its own #ifdef for the duration of its A / B and leaves with its evidence)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
src=fused_step_ring.hip
if [ "$1" = "-s" ]; then src=$2; shift 2; fi
mkdir -p lit_llama_amd/_variants /tmp/variants
obj=/tmp/variants/${src%.hip}_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c lit_llama_amd/csrc/$src -o $obj
objs=$(ls lit_llama_amd/csrc/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lit_llama_amd/_variants/libmi355llama_$tag.so $objs $obj
echo built lit_llama_amd/_variants/libmi355llama_$tag.so
