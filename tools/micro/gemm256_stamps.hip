// Diagnostic build of the 256-tile GEMM with in-kernel s_memtime stamps (never part of libcodetr_hip.so):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -fno-slp-vectorize tools/micro/gemm256_stamps.hip -o /tmp/gemm_stamps
//   /tmp/gemm_stamps M N K [act] [residual]
// Per workgroup: start, end of the main loop, epilogue stores issued, stores completed (shader cycles) + realtime
// (100 MHz) at start / end -> main-loop and epilogue time per tile, the clock, and how the tiles of the 256 CUs line up.
#ifndef NOSTAMPS   // -DNOSTAMPS -DCODETR_GEMM_ABL=mask: ablation builds timed by events only (the stamps themselves cost ~15 %)
#define CODETR_GEMM_STAMPS
#endif
#include "../../co-detr-tensorrt_amd/csrc/gemm_f16.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 80640, N = argc > 2 ? atoll(argv[2]) : 2304, K = argc > 3 ? atoll(argv[3]) : 768;
  const int act = argc > 4 ? atoi(argv[4]) : 0, res = argc > 5 ? atoi(argv[5]) : 0;
  unsigned short *X, *W, *B, *R, *Y;
  hipMalloc(&X, M * K * 2); hipMalloc(&W, N * K * 2); hipMalloc(&B, N * 2); hipMalloc(&R, M * N * 2); hipMalloc(&Y, M * N * 2);
  std::vector<unsigned short> h(std::max(M * K, N * K));
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3000 + (unsigned short)((i * 2654435761u) >> 22);  // ~0.1-0.25, fp16
  hipMemcpy(X, h.data(), M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), N * K * 2, hipMemcpyHostToDevice);
  hipMemset(B, 0, N * 2); hipMemset(R, 0, M * N * 2);
  const int tiles = (int)(((M + 255) / 256) * ((N + 255) / 256));
  unsigned long long* stamps;
  hipMalloc(&stamps, (size_t)tiles * 64);
#ifdef CODETR_GEMM_STAMPS
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps));
#endif
  for (int it = 0; it < 5; ++it) codetr_linear_f16(nullptr, X, W, B, res ? R : nullptr, nullptr, Y, M, N, K, act, 0, 0);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  codetr_linear_f16(nullptr, X, W, B, res ? R : nullptr, nullptr, Y, M, N, K, act, 0, 0);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
#ifndef CODETR_GEMM_STAMPS
  printf("M %lld N %lld K %lld act %d res %d: kernel %.1f us (event), ablation mask %d\n", (long long)M, (long long)N, (long long)K, act, res, ms * 1e3, kGemmAbl);
  return 0;
#endif
  std::vector<unsigned long long> s((size_t)tiles * 8);
  hipMemcpy(s.data(), stamps, s.size() * 8, hipMemcpyDeviceToHost);
  double main_c = 0, epi_issue = 0, epi_done = 0, clk = 0, wdma = 0, wbar = 0;
  unsigned long long r0 = ~0ull, r1 = 0;
  for (int t = 0; t < tiles; ++t) {
    const unsigned long long* o = &s[(size_t)t * 8];
    wdma += (double)o[6]; wbar += (double)o[7];
    main_c += (double)(o[1] - o[0]); epi_issue += (double)(o[2] - o[1]); epi_done += (double)(o[3] - o[1]);
    clk += (double)(o[3] - o[0]) / (double)(o[5] - o[4]) * 100.0;   // MHz
    r0 = std::min(r0, o[4]); r1 = std::max(r1, o[5]);
  }
  printf("M %lld N %lld K %lld act %d res %d: %d tiles, kernel %.1f us (event), span %.1f us (realtime)\n", (long long)M, (long long)N,
         (long long)K, act, res, tiles, ms * 1e3, (double)(r1 - r0) / 100.0);
  printf("  per tile (wave 0): main loop %.0f cyc (of which waiting for the DMA %.0f, at the barrier %.0f; %lld k-tiles), epilogue until stores issued %.0f cyc, until stores completed %.0f cyc; clock %.0f MHz\n",
         main_c / tiles, wdma / tiles, wbar / tiles, (long long)(K / 64), epi_issue / tiles, epi_done / tiles, clk / tiles);
  // how synchronised are the epilogues?  histogram of (main-loop end time, realtime-equivalent) modulo the mean tile time
  const double tile_c = (main_c + epi_done) / tiles;
  int hist[8] = {0};
  for (int t = 0; t < tiles; ++t) {
    const unsigned long long* o = &s[(size_t)t * 8];
    const double rt_end_main = (double)(o[4] - r0) + (double)(o[1] - o[0]) / (clk / tiles) * 100.0;   // in 10-ns ticks
    const double tile_ticks = tile_c / (clk / tiles) * 100.0;
    const double ph = rt_end_main / tile_ticks;
    hist[(int)((ph - (long long)ph) * 8) & 7]++;
  }
  printf("  phase histogram of main-loop ends (8 bins of a tile time): ");
  for (int i = 0; i < 8; ++i) printf("%d ", hist[i]);
  printf("\n");
  return 0;
}
