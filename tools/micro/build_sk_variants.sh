#!/bin/bash
# diagnostic builds of csrc/gemm_sk.hip for tools/micro/gemm_sk_bench (--sklib name=path): ablation masks and experiments
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/_bin
# (compiler, flags and the probed LLVM option of the product build: make -s print-*)
MK=co-detr-tensorrt_amd/csrc
HIPCC=$(make -s -C $MK print-hipcc)
FLAGS="$(make -s -C $MK print-flags) -Ico-detr-tensorrt_amd/csrc $(make -s -C $MK print-vgprform) -shared"
for spec in "$@"; do   # e.g. abl1:-DCODETR_SK_ABL=1
  name=${spec%%:*}; defs=${spec#*:}
  $HIPCC $FLAGS $defs co-detr-tensorrt_amd/csrc/gemm_sk.hip -o tools/micro/_bin/libsk_$name.so &
done
wait
ls -la tools/micro/_bin/libsk_*.so
