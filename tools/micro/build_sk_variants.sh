#!/bin/bash
# diagnostic builds of csrc/gemm_sk.hip for tools/micro/gemm_sk_bench (--sklib name=path): ablation masks and experiments
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/_bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ico-detr-tensorrt_amd/csrc -Wno-inline-asm -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -shared"
for spec in "$@"; do   # e.g. abl1:-DCODETR_SK_ABL=1
  name=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs co-detr-tensorrt_amd/csrc/gemm_sk.hip -o tools/micro/_bin/libsk_$name.so &
done
wait
ls -la tools/micro/_bin/libsk_*.so
