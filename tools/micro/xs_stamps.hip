// Diagnostic build of the X-stationary GEMM with in-kernel s_memtime stamps (never part of libcodetr_hip.so):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form tools/micro/xs_stamps.hip -o /tmp/xs_stamps
//   /tmp/xs_stamps M N1 N2      (N2 = 0: codetr_linear_xadd_f16 with N = N1; else codetr_encoder_projections_f16)
// Per workgroup, wave 0: cycles in the six sections of the chunk loop (waiting for the W chunk | at the barrier | flushing a
// staged pair / requesting residual rows | issuing the LDS-DMA pieces | bias + fragment reads + MFMAs | activation + pack +
// staging writes), the prologue, the whole workgroup; realtime (100 MHz) at start / end.
#ifndef NOSTAMPS
#define CODETR_XS_STAMPS
#endif
#include "../../co-detr-tensorrt_amd/csrc/gemm_f16.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 818400, N1 = argc > 2 ? atoll(argv[2]) : 512, N2 = argc > 3 ? atoll(argv[3]) : 0, K = 256;
  const int64_t N = N1 + N2;
  unsigned short *X, *P, *W, *B, *Y, *Y2;
  hipMalloc(&X, M * K * 2); hipMalloc(&P, M * K * 2); hipMalloc(&W, N * K * 2); hipMalloc(&B, N * 2); hipMalloc(&Y, M * N * 2);
  Y2 = Y + M * N1;
  std::vector<unsigned short> h(M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3000 + (unsigned short)((i * 2654435761u) >> 22);
  hipMemcpy(X, h.data(), M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(P, h.data(), M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), N * K * 2, hipMemcpyHostToDevice);
  hipMemset(B, 0, N * 2);
  const int wgs = (int)((M + 127) / 128);
  unsigned long long* stamps;
  hipMalloc(&stamps, (size_t)wgs * 12 * 8);
#ifdef CODETR_XS_STAMPS
  hipMemcpyToSymbol(HIP_SYMBOL(g_xs_stamps), &stamps, sizeof(stamps));
#endif
  auto run = [&]() {
    return N2 ? codetr_encoder_projections_f16(nullptr, X, P, W, B, nullptr, Y, Y2, M, N1, N2, K, 0, 0)
              : codetr_linear_xadd_f16(nullptr, X, P, W, B, Y, M, N, K);
  };
  for (int it = 0; it < 5; ++it) {
    const int rc = run();
    if (rc) { printf("rc %d\n", rc); return 1; }
  }
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  run();
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("M %lld N %lld (+%lld) K 256: kernel %.1f us (event), %d workgroups\n", (long long)M, (long long)N1, (long long)N2, ms * 1e3, wgs);
#ifdef CODETR_XS_STAMPS
  std::vector<unsigned long long> s((size_t)wgs * 12);
  hipMemcpy(s.data(), stamps, s.size() * 8, hipMemcpyDeviceToHost);
  double acc[8] = {0}, clk = 0;
  for (int t = 0; t < wgs; ++t) {
    const unsigned long long* o = &s[(size_t)t * 12];
    for (int i = 0; i < 8; ++i) acc[i] += (double)o[i];
    clk += (double)o[7] / (double)(o[9] - o[8]) * 100.0;
  }
  const char* names[8] = {"wait W chunk", "barrier", "flush / residual", "LDS-DMA issue", "bias + reads + MFMAs", "act + pack + stage", "prologue", "workgroup"};
  const int nch = (int)((N + 31) / 32);
  for (int i = 0; i < 8; ++i)
    printf("  %-22s %9.0f cyc per workgroup%s\n", names[i], acc[i] / wgs, i < 6 ? "" : "");
  printf("  per chunk: ");
  for (int i = 0; i < 6; ++i) printf("%s %.0f  ", names[i], acc[i] / wgs / nch);
  printf("\n  clock %.0f MHz, %d chunks\n", clk / wgs, nch);
#endif
  return 0;
}
