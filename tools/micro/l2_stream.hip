// How fast can a CU take in operand tiles that sit in L2 / MALL, by which instruction and with how many waves?
// The GEMM kernels of this repo (128-tile, 256-tile, X-stationary, fused FFN) all measure ~15 B/clk/CU of L2 -> LDS
// traffic in their main loops while hipBLASLt's 256x256x64 kernels reach ~21; this probe separates
// the transport from the kernels: every workgroup streams GEMM-shaped tiles (rows of 128 contiguous bytes, row stride
// K * 2 bytes) of a small matrix that stays cache-resident, and nothing is computed.
//   mode 0: global_load_dwordx4 into registers (DEPTH loads in flight per thread, results xor-ed into a sink)
//   mode 1: LDS-DMA (global_load_lds_dwordx4) into an LDS ring, counted vmcnt wait, nobody reads the LDS
//   mode 2: global_load_dwordx4 + ds_write_b128 (register staging into the same LDS ring)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2_stream.hip -o tools/micro/_bin/l2_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

// One "k-tile" = ROWS rows x 128 bytes.  Thread t of NT handles pieces u = q * NT + t: row u / 8, 16-byte chunk u % 8.
template <int NT, int ROWS, int MODE, int DEPTH>
__global__ __launch_bounds__(NT) void stream_kernel(const unsigned char* __restrict__ base, int rows_total, int kbytes,
                                                    int iters, int* __restrict__ sink) {
  constexpr int kPieces = ROWS * 8 / NT;   // 16-byte pieces per thread per k-tile
  static_assert(kPieces >= 1, "tile too small for the block");
  constexpr int kTileBytes = ROWS * 128;
  constexpr int kSlots = MODE == 0 ? 1 : (131072 / kTileBytes < 4 ? 131072 / kTileBytes : 4);
  __shared__ __attribute__((aligned(16))) unsigned char lds[MODE == 0 ? 16 : kSlots * kTileBytes];
  const int tid = threadIdx.x;
  const int row_blocks = rows_total / ROWS, kt = kbytes / 128;
  i32x4 acc = {0, 0, 0, 0};
  unsigned rb = (blockIdx.x * 7u) % (unsigned)row_blocks, kc = blockIdx.x % (unsigned)kt;
  if (MODE == 0) {
    i32x4 buf[DEPTH][kPieces];
    // prologue: DEPTH tiles in flight
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int q = 0; q < kPieces; ++q) {
        const int u = q * NT + tid;
        buf[d][q] = *reinterpret_cast<const i32x4*>(base + ((size_t)rb * ROWS + (u >> 3)) * kbytes + kc * 128 + (u & 7) * 16);
      }
      kc = kc + 1 == (unsigned)kt ? 0 : kc + 1;
      if (kc == 0) rb = rb + 1 == (unsigned)row_blocks ? 0 : rb + 1;
    }
    for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
        for (int q = 0; q < kPieces; ++q) {
          acc ^= buf[d][q];
          const int u = q * NT + tid;
          buf[d][q] = *reinterpret_cast<const i32x4*>(base + ((size_t)rb * ROWS + (u >> 3)) * kbytes + kc * 128 + (u & 7) * 16);
        }
        kc = kc + 1 == (unsigned)kt ? 0 : kc + 1;
        if (kc == 0) rb = rb + 1 == (unsigned)row_blocks ? 0 : rb + 1;
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int q = 0; q < kPieces; ++q) acc ^= buf[d][q];
  } else if (MODE == 1) {
    // LDS-DMA: kSlots - 1 tiles in flight, wait for the oldest before re-using its slot
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < kPieces; ++q) {
        const int u = q * NT + tid;
        const unsigned char* g = base + ((size_t)rb * ROWS + (u >> 3)) * kbytes + kc * 128 + (u & 7) * 16;
        // destination: lane-linear inside the wave's 1-KiB piece (the builtin adds lane * 16 itself)
        unsigned char* l = lds + slot * kTileBytes + (q * NT + (tid & ~63)) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
      }
      kc = kc + 1 == (unsigned)kt ? 0 : kc + 1;
      if (kc == 0) rb = rb + 1 == (unsigned)row_blocks ? 0 : rb + 1;
      slot = slot + 1 == kSlots ? 0 : slot + 1;
      // all but the (kSlots - 1) youngest tiles landed
      if (kSlots == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * kPieces > 63 ? 63 : 3 * kPieces) : "memory");
      else if (kSlots == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPieces > 63 ? 63 : kPieces) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc[0] = lds[tid * 16];
  } else {
    i32x4 buf[DEPTH][kPieces];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int q = 0; q < kPieces; ++q) {
        const int u = q * NT + tid;
        buf[d][q] = *reinterpret_cast<const i32x4*>(base + ((size_t)rb * ROWS + (u >> 3)) * kbytes + kc * 128 + (u & 7) * 16);
      }
      kc = kc + 1 == (unsigned)kt ? 0 : kc + 1;
      if (kc == 0) rb = rb + 1 == (unsigned)row_blocks ? 0 : rb + 1;
    }
    int slot = 0;
    for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
        for (int q = 0; q < kPieces; ++q) {
          const int u = q * NT + tid;
          *reinterpret_cast<i32x4*>(lds + slot * kTileBytes + u * 16) = buf[d][q];
          buf[d][q] = *reinterpret_cast<const i32x4*>(base + ((size_t)rb * ROWS + (u >> 3)) * kbytes + kc * 128 + (u & 7) * 16);
        }
        kc = kc + 1 == (unsigned)kt ? 0 : kc + 1;
        if (kc == 0) rb = rb + 1 == (unsigned)row_blocks ? 0 : rb + 1;
        slot = slot + 1 == kSlots ? 0 : slot + 1;
      }
    }
    __syncthreads();
    acc[0] = lds[tid * 16];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int q = 0; q < kPieces; ++q) acc ^= buf[d][q];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) sink[0] = 1;
}

template <int NT, int ROWS, int MODE, int DEPTH>
static void run(const char* what, const unsigned char* d, int rows_total, int kbytes, int wgs_per_cu, int* sink) {
  const int iters = 2048 * 256 / ROWS;     // 64 MiB per workgroup... scaled: constant bytes per workgroup = 2048 * 32 KiB
  const int grid = 256 * wgs_per_cu;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((stream_kernel<NT, ROWS, MODE, DEPTH>), dim3(grid), dim3(NT), 0, 0, d, rows_total, kbytes, iters / 8, sink);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  hipLaunchKernelGGL((stream_kernel<NT, ROWS, MODE, DEPTH>), dim3(grid), dim3(NT), 0, 0, d, rows_total, kbytes, iters, sink);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)grid * iters * ROWS * 128;
  const hipError_t e = hipGetLastError();
  printf("%-58s matrix %5.1f MiB  wg/CU %d  %8.1f us  %6.2f TB/s  %5.1f GB/s/CU  %5.1f B/clk/CU at 2.1 GHz%s\n", what,
         rows_total * (double)kbytes / 1048576.0, wgs_per_cu, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256,
         bytes / (ms * 1e-3) / 256 / 2.1e9, e == hipSuccess ? "" : "  ERROR");
}

int main(int argc, char** argv) {
  int* sink;
  hipMalloc(&sink, 4);
  for (int pass = 0; pass < 3; ++pass) {
    // matrix sizes: 2 MiB (fits every XCD's 4-MiB L2), 16 MiB (MALL), 512 MiB (HBM)
    const int kbytes = 2048;                    // K = 1024 halves
    const int rows_total = pass == 0 ? 1024 : pass == 1 ? 8192 : 262144;
    unsigned char* d;
    hipMalloc(&d, (size_t)rows_total * kbytes);
    hipMemset(d, 1, (size_t)rows_total * kbytes);
    printf("== %s ==\n", pass == 0 ? "2 MiB matrix (L2-resident)" : pass == 1 ? "16 MiB matrix (MALL-resident)" : "512 MiB matrix (HBM)");
    run<256, 256, 0, 2>("global_load -> VGPR, 4 waves, 256-row tiles, depth 2", d, rows_total, kbytes, 1, sink);
    run<256, 256, 0, 4>("global_load -> VGPR, 4 waves, 256-row tiles, depth 4", d, rows_total, kbytes, 1, sink);
    run<256, 256, 0, 4>("global_load -> VGPR, 4 waves, 256-row tiles, depth 4", d, rows_total, kbytes, 2, sink);
    run<512, 256, 0, 4>("global_load -> VGPR, 8 waves, 256-row tiles, depth 4", d, rows_total, kbytes, 1, sink);
    run<512, 512, 0, 4>("global_load -> VGPR, 8 waves, 512-row tiles, depth 4", d, rows_total, kbytes, 1, sink);
    run<1024, 512, 0, 4>("global_load -> VGPR, 16 waves, 512-row tiles, depth 4", d, rows_total, kbytes, 1, sink);
    run<256, 256, 1, 1>("LDS-DMA, 4 waves, 256-row tiles (32 KiB), 4 slots", d, rows_total, kbytes, 1, sink);
    run<512, 256, 1, 1>("LDS-DMA, 8 waves, 256-row tiles (32 KiB), 4 slots", d, rows_total, kbytes, 1, sink);
    run<512, 512, 1, 1>("LDS-DMA, 8 waves, 512-row tiles (64 KiB), 2 slots", d, rows_total, kbytes, 1, sink);
    run<1024, 256, 1, 1>("LDS-DMA, 16 waves, 256-row tiles (32 KiB), 4 slots", d, rows_total, kbytes, 1, sink);
    run<256, 128, 1, 1>("LDS-DMA, 4 waves, 128-row tiles (16 KiB), 4 slots, 2 WG", d, rows_total, kbytes, 2, sink);
    run<256, 256, 2, 2>("global_load + ds_write, 4 waves, 256-row tiles, depth 2", d, rows_total, kbytes, 1, sink);
    run<512, 256, 2, 4>("global_load + ds_write, 8 waves, 256-row tiles, depth 4", d, rows_total, kbytes, 1, sink);
    run<512, 512, 2, 2>("global_load + ds_write, 8 waves, 512-row tiles, depth 2", d, rows_total, kbytes, 1, sink);
    hipFree(d);
  }
  return 0;
}
