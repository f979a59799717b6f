#!/bin/bash
# diagnostic builds of the whole library with csrc/decoder_layer.hip compiled under extra definitions (timing experiments):
#   tools/micro/build_dec_variants.sh stamps:-DCODETR_DEC_STAMPS abl1:-DCODETR_DEC_ABL=1
# -> tools/micro/_bin/libcodetr_dec_<name>.so, loaded by tools/bench_decoder.py via CODETR_LIB
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/_bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ico-detr-tensorrt_amd/csrc -Wno-inline-asm -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form"
OTHERS=$(ls co-detr-tensorrt_amd/csrc/_obj/*.o | grep -v decoder_layer.o | grep -v amdgcn)
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c co-detr-tensorrt_amd/csrc/decoder_layer.hip -o tools/micro/_bin/dec_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/_bin/libcodetr_dec_$name.so tools/micro/_bin/dec_$name.o $OTHERS ) &
done
wait
ls -la tools/micro/_bin/libcodetr_dec_*.so
