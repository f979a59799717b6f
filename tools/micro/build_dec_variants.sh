#!/bin/bash
# diagnostic builds of the whole library with csrc/decoder_layer.hip compiled under extra definitions (timing experiments):
#   tools/micro/build_dec_variants.sh stamps:-DCODETR_DEC_STAMPS abl1:-DCODETR_DEC_ABL=1
# -> tools/micro/_bin/libcodetr_dec_<name>.so, loaded by tools/bench_decoder.py via CODETR_LIB
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/_bin
# (compiler, flags and the probed LLVM option of the product build: make -s print-*)
MK=co-detr-tensorrt_amd/csrc
HIPCC=$(make -s -C $MK print-hipcc)
FLAGS="$(make -s -C $MK print-flags) -Ico-detr-tensorrt_amd/csrc $(make -s -C $MK print-vgprform)"
OTHERS=$(ls co-detr-tensorrt_amd/csrc/_obj/*.o | grep -v decoder_layer.o | grep -v amdgcn)
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( $HIPCC $FLAGS $defs -c co-detr-tensorrt_amd/csrc/decoder_layer.hip -o tools/micro/_bin/dec_$name.o &&
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o tools/micro/_bin/libcodetr_dec_$name.so tools/micro/_bin/dec_$name.o $OTHERS ) &
done
wait
ls -la tools/micro/_bin/libcodetr_dec_*.so
