#!/bin/bash
# diagnostic builds of the whole library with ONE source compiled under extra definitions:
#   tools/micro/build_variant.sh <source.hip> <name> "<defs>"   -> tools/micro/_bin/libcodetr_<name>.so (CODETR_LIB for the tools/)
# Compiler, flags and the probed -amdgpu-mfma-vgpr-form option come from the product Makefile (make -s print-*), so that a
# variant differs from the shipped object by its definitions only.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/_bin
src=$1; name=$2; defs=$3
base=$(basename $src .hip)
MK=co-detr-tensorrt_amd/csrc
HIPCC=$(make -s -C $MK print-hipcc)
FLAGS="$(make -s -C $MK print-flags) -Ico-detr-tensorrt_amd/csrc"
case $base in gemm_f16|gemm_sk|gemm_pp|window_attention|decoder_layer|decoder_layer_bf16) FLAGS="$FLAGS $(make -s -C $MK print-vgprform)";; esac
OTHERS=$(ls $MK/_obj/*.o | grep -v "/$base.o" | grep -v amdgcn)
$HIPCC $FLAGS $defs -c $MK/$base.hip -o tools/micro/_bin/${base}_$name.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o tools/micro/_bin/libcodetr_$name.so tools/micro/_bin/${base}_$name.o $OTHERS
ls -la tools/micro/_bin/libcodetr_$name.so
