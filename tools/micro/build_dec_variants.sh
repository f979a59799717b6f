#!/bin/bash
# diagnostic builds of the whole library with csrc/decoder_layer.hip compiled under -DCODETR_DEC_ABL=<mask> (timing
# experiments, WRONG results): tools/micro/_bin/libcodetr_dec<mask>.so, loaded by tools/bench_decoder.py via CODETR_LIB
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/_bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ico-detr-tensorrt_amd/csrc -Wno-inline-asm -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form"
OTHERS=$(ls co-detr-tensorrt_amd/csrc/_obj/*.o | grep -v decoder_layer.o | grep -v amdgcn)
for m in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DCODETR_DEC_ABL=$m -c co-detr-tensorrt_amd/csrc/decoder_layer.hip -o tools/micro/_bin/dec$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/_bin/libcodetr_dec$m.so tools/micro/_bin/dec$m.o $OTHERS ) &
done
wait
ls -la tools/micro/_bin/libcodetr_dec*.so
