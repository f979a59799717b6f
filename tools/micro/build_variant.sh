#!/bin/bash
# diagnostic builds of the whole library with ONE source compiled under extra definitions:
#   tools/micro/build_variant.sh <source.hip> <name> "<defs>"   -> tools/micro/_bin/libcodetr_<name>.so (CODETR_LIB for the tools/)
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/_bin
src=$1; name=$2; defs=$3
base=$(basename $src .hip)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ico-detr-tensorrt_amd/csrc -Wno-inline-asm -fno-slp-vectorize"
case $base in gemm_f16|gemm_sk|window_attention|decoder_layer) FLAGS="$FLAGS -mllvm -amdgpu-mfma-vgpr-form";; esac
OTHERS=$(ls co-detr-tensorrt_amd/csrc/_obj/*.o | grep -v "/$base.o" | grep -v amdgcn)
/opt/rocm/bin/hipcc $FLAGS $defs -c co-detr-tensorrt_amd/csrc/$base.hip -o tools/micro/_bin/${base}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/_bin/libcodetr_$name.so tools/micro/_bin/${base}_$name.o $OTHERS
ls -la tools/micro/_bin/libcodetr_$name.so
