// Micro-benchmark: how fast can one 512-thread workgroup per CU write a 256 x 256 fp16 tile of a row-major [M, N]
// matrix, as a function of how many contiguous bytes of one row a wave instruction covers?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef short s16x8 __attribute__((ext_vector_type(8)));

// SEG = contiguous bytes of one row per wave instruction (128, 256, 512); every lane stores 16 B
template <int SEG>
__global__ __launch_bounds__(512) void tile_store(unsigned short* Y, int N, int tiles_n, int reps) {
  const int tile = blockIdx.x, tn = tile % tiles_n, tm = tile / tiles_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16;        // lanes per row segment
  constexpr int RPI = 64 / LPR;        // rows per instruction
  constexpr int WCOLS = SEG / 2;       // columns a wave owns
  constexpr int WN = 256 / WCOLS;      // waves along n
  const int wn = wave % WN, wm = wave / WN;
  constexpr int ROWS_PER_WAVE = 256 / (8 / WN);
  s16x8 v = {1, 2, 3, 4, 5, 6, 7, (short)lane};
  for (int r = 0; r < reps; ++r) {
    for (int it = 0; it < ROWS_PER_WAVE / RPI; ++it) {
      const int row = tm * 256 + wm * ROWS_PER_WAVE + it * RPI + lane / LPR;
      const int col = tn * 256 + wn * WCOLS + (lane % LPR) * 8;
      *reinterpret_cast<s16x8*>(Y + (size_t)row * N + col) = v;
    }
  }
}

template <int SEG>
void run(unsigned short* Y, int M, int N) {
  const int tiles_n = N / 256, tiles = (M / 256) * tiles_n;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(tile_store<SEG>, dim3(tiles), dim3(512), 0, 0, Y, N, tiles_n, 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(tile_store<SEG>, dim3(tiles), dim3(512), 0, 0, Y, N, tiles_n, 1);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  printf("segment %3d B per row per instruction: %7.1f us  %6.2f TB/s (%d tiles of 128 KiB)\n", SEG, ms * 1e3,
         (double)M * N * 2 / ms / 1e9, tiles);
}

int main() {
  const int M = 80640 / 256 * 256, N = 2304;
  unsigned short* Y;
  hipMalloc(&Y, (size_t)M * N * 2);
  run<128>(Y, M, N);
  run<256>(Y, M, N);
  run<512>(Y, M, N);
  run<128>(Y, M, N);
  return 0;
}
