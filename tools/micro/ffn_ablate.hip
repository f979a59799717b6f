// Diagnostic builds of the fused fp16 FFN (never part of libcodetr_hip.so): parts of the chunk loop compiled out.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -fno-slp-vectorize -DCODETR_FFN_ABL=mask tools/micro/ffn_ablate.hip -o ...
//   mask: 1 = no LDS-DMA inside the chunk loop, 2 = no MFMAs, 4 = no W fragment reads inside the chunk loop
#include "../../co-detr-tensorrt_amd/csrc/ffn_fused.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 818400, C = 256, Hd = 2048;
  unsigned short *X, *W1, *W2, *W2p, *B1, *B2, *Y;
  hipMalloc(&X, M * C * 2); hipMalloc(&Y, M * C * 2); hipMalloc(&W1, Hd * C * 2); hipMalloc(&W2, C * Hd * 2); hipMalloc(&W2p, C * Hd * 2);
  hipMalloc(&B1, Hd * 2); hipMalloc(&B2, C * 2);
  std::vector<unsigned short> h(M * C);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x2c00 + (unsigned short)((i * 2654435761u) >> 23);   // ~0.06-0.1 fp16
  hipMemcpy(X, h.data(), M * C * 2, hipMemcpyHostToDevice);
  hipMemcpy(W1, h.data(), Hd * C * 2, hipMemcpyHostToDevice);
  hipMemcpy(W2, h.data(), C * Hd * 2, hipMemcpyHostToDevice);
  hipMemset(B1, 0, Hd * 2); hipMemset(B2, 0, C * 2);
  codetr_ffn_pack_w2_f16(nullptr, W2, W2p, C, Hd);
  for (int it = 0; it < 3; ++it) codetr_ffn_relu_f16(nullptr, X, W1, B1, W2p, B2, Y, M, C, Hd);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) codetr_ffn_relu_f16(nullptr, X, W1, B1, W2p, B2, Y, M, C, Hd);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 5;
  printf("ffn_fused fp16 M %lld: %.1f us per launch, %.1f TF/s, ablation mask %d (%s)\n", (long long)M, us,
         4.0 * M * C * Hd / us / 1e6, CODETR_FFN_ABL_MASK, hipGetErrorString(hipGetLastError()));
  return 0;
}
