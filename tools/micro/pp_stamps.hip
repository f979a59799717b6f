// Diagnostic build of the ping-pong GEMM (csrc/gemm_pp.hip) with in-kernel cycle stamps (never part of libcodetr_hip.so):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ico-detr-tensorrt_amd/csrc -fno-slp-vectorize \
//         -mllvm -amdgpu-mfma-vgpr-form tools/micro/pp_stamps.hip -o tools/micro/_bin/pp_stamps
//   tools/micro/_bin/pp_stamps M N K [flags ...]     (flags as codetr_linear_pp_f16; include tools/micro/experiments/gemm_pp_variants.hip instead for the variants)
// Per wave, sums over the main loops of: LOAD segment issue | wait for staged data (vmcnt) | wait for own LDS operations |
// barrier behind the LOAD segment | MFMA segment | barrier behind it | epilogue | whole kernel -- averaged over the waves
// of each group and over the workgroups, per half-stage.
#define CODETR_PP_STAMPS
#include "../../co-detr-tensorrt_amd/csrc/gemm_pp.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 38400, N = argc > 2 ? atoll(argv[2]) : 768, K = argc > 3 ? atoll(argv[3]) : 3072;
  std::vector<int> flags;
  for (int i = 4; i < argc; ++i) flags.push_back(atoi(argv[i]));
  if (flags.empty()) flags = {0};
  unsigned short *X, *W, *B, *Y;
  hipMalloc(&X, M * K * 2); hipMalloc(&W, N * K * 2); hipMalloc(&B, N * 2); hipMalloc(&Y, M * N * 2);
  std::vector<unsigned short> h((size_t)M * K);
  unsigned r = 12345u;
  for (size_t i = 0; i < h.size(); ++i) {   // random halves in [-2, 2): sign, exponent 0x3800-0x3fff range, random mantissa
    r = r * 1664525u + 1013904223u;
    h[i] = (unsigned short)(((r >> 16) & 0x8000u) | 0x3400u | ((r >> 4) & 0x0fffu));
  }
  hipMemcpy(X, h.data(), M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), N * K * 2, hipMemcpyHostToDevice);
  hipMemset(B, 0, N * 2);
  const int G = 256;
  unsigned long long* stamps;
  hipMalloc(&stamps, (size_t)G * 64 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_pp_stamps), &stamps, sizeof(stamps));
  const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
  const double hs_per_wg = (double)tiles / G * (K / 32);   // half-stages an average workgroup runs
  const char* names[8] = {"LOAD issue", "wait data", "wait LDS", "barrier A", "MFMA seg", "barrier B", "epilogue", "kernel"};
  for (int f : flags) {
    for (int it = 0; it < 3; ++it) {
      const int rc = codetr_linear_pp_f16(nullptr, X, W, B, nullptr, Y, M, N, K, 0, f);
      if (rc) { printf("rc %d\n", rc); return 1; }
    }
    hipDeviceSynchronize();
    hipMemset(stamps, 0, (size_t)G * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    codetr_linear_pp_f16(nullptr, X, W, B, nullptr, Y, M, N, K, 0, f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> s((size_t)G * 64);
    hipMemcpy(s.data(), stamps, s.size() * 8, hipMemcpyDeviceToHost);
    printf("M %lld N %lld K %lld flags %d: kernel %.1f us (event, stamped build), %.1f half-stages per workgroup\n", (long long)M,
           (long long)N, (long long)K, f, ms * 1e3, hs_per_wg);
    for (int grp = 0; grp < 2; ++grp) {
      double acc[8] = {0};
      int n = 0;
      for (int wg = 0; wg < G; ++wg)
        for (int w = grp * 4; w < grp * 4 + 4; ++w) {
          const unsigned long long* o = &s[((size_t)wg * 8 + w) * 8];
          if (o[7] == 0) continue;
          for (int i = 0; i < 8; ++i) acc[i] += (double)o[i];
          ++n;
        }
      printf("  group %d (%d waves), cycles per half-stage:", grp, n);
      for (int i = 0; i < 6; ++i) printf("  %s %.0f", names[i], acc[i] / n / hs_per_wg);
      printf("  | per workgroup: epilogue %.0f, kernel %.0f\n", acc[6] / n, acc[7] / n);
    }
  }
  return 0;
}
