// Diagnostic build of the e4m3 fused FFN with parts of its main loop removed (never part of libcodetr_hip.so):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -fno-slp-vectorize -DFFN8_ABLATE=<bits> tools/micro/ffn8_ablate.hip -o ...
//   bits: 1 no LDS-DMA in the loop, 2 no activation, 4 no product 2, 8 no product 1, 16 no barriers, 32 no LDS operand reads
// Prints the launch time at the encoder's shape; the difference to the full kernel prices each part.
#include "../../co-detr-tensorrt_amd/csrc/ffn_fp8.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 204600, Hd = 2048;
  const int opt = argc > 2 ? atoi(argv[2]) : 3;  // 1 LayerNorm of the input, 2 LayerNorm of the output, 4 + pos second output
  void *X, *W1, *W2, *B1, *B2, *Y, *G, *Y2;
  float *S1, *S2;
  hipMalloc(&X, M * 512); hipMalloc(&Y, M * 512); hipMalloc(&Y2, M * 512); hipMalloc(&W1, Hd * 256); hipMalloc(&W2, Hd * 256);
  hipMalloc(&B1, Hd * 2); hipMalloc(&B2, 512); hipMalloc(&G, 512); hipMalloc(&S1, Hd * 4); hipMalloc(&S2, 1024);
  std::vector<unsigned short> h(M * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3000 + (unsigned short)((i * 2654435761u) >> 22);
  hipMemcpy(X, h.data(), M * 512, hipMemcpyHostToDevice);
  hipMemset(W1, 0x38, Hd * 256); hipMemset(W2, 0x30, Hd * 256); hipMemset(B1, 0, Hd * 2); hipMemset(B2, 0, 512);
  std::vector<float> s(Hd, 1e-3f);
  hipMemcpy(S1, s.data(), Hd * 4, hipMemcpyHostToDevice); hipMemcpy(S2, s.data(), 1024, hipMemcpyHostToDevice);
  std::vector<unsigned short> g(256, 0x3c00);
  hipMemcpy(G, g.data(), 512, hipMemcpyHostToDevice);
  auto run = [&] { return codetr_ffn_fp8(nullptr, X, W1, S1, B1, W2, S2, B2, Y, M, 256, Hd, 0.01f, 0.02f, (opt & 1) ? G : nullptr, (opt & 1) ? B2 : nullptr, 1e-5f,
                                           (opt & 2) ? G : nullptr, (opt & 2) ? B2 : nullptr, 1e-5f, (opt & 4) ? Y : nullptr, (opt & 4) ? Y2 : nullptr); };
  for (int i = 0; i < 5; ++i) if (run() != 0) { printf("launch failed\n"); return 1; }
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f, sum = 0;
  for (int i = 0; i < 20; ++i) {
    hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
  }
#ifdef FFN8_STAMPS
  {
    unsigned long long* st; hipMalloc(&st, 256 * 64); hipMemset(st, 0, 256 * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ffn8_stamps), &st, sizeof(st));
    run(); hipDeviceSynchronize();
    std::vector<unsigned long long> hs(256 * 8);
    hipMemcpy(hs.data(), st, 256 * 64, hipMemcpyDeviceToHost);
    double acc[8] = {0}; int n = 0;
    for (int b = 0; b < 256; ++b) if (hs[b * 8 + 7]) { for (int i = 0; i < 8; ++i) acc[i] += (double)hs[b * 8 + i]; ++n; }
    const double ch = acc[7] / n;
    printf("stamps (s_memtime ticks, mean over %d workgroups, %.0f chunks each): total %.0f | per chunk: T wait+barrier %.0f, product 1 %.0f, "
           "M wait+barrier %.0f, product 2 %.0f | per tile: quantise %.0f, epilogue(+next rows) %.0f\n", n, ch, acc[6] / n,
           acc[1] / n / ch, acc[2] / n / ch, acc[3] / n / ch, acc[4] / n / ch, acc[0] / n / (ch / 16), acc[5] / n / (ch / 16));
  }
#endif
  printf("FFN8_ABLATE=%d opt=%d M=%lld: best %.1f us, mean %.1f us\n", FFN8_ABLATE, opt, (long long)M, best * 1e3f, sum / 20 * 1e3f);
  return 0;
}
