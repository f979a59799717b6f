// Micro-benchmark: issue rate of the vector instructions the MSDA blend can be built from, at 1 / 2 / 4 waves per SIMD.
// For each instruction a wave runs ITER x 64 independent copies (8 accumulators x 8) and stamps s_memtime around the
// loop; reported: shader cycles per wave-instruction per SIMD (= loop cycles / instructions issued by ALL waves of the
// SIMD), median over workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o tools/micro/_bin/valu_rates && tools/micro/_bin/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

constexpr int ITER = 2000;

enum Op { FMA = 0, FMA_MIX, DOT2, PK_FMA_F32, DPP_MOV, PK_FMA_F16, DOT2_DPP, FMA_MIX_HI, DS_READ_B128, NOPS };
const char* kNames[] = {"v_fma_f32",         "v_fma_mix_f32 (f16 b)", "v_dot2_f32_f16",       "v_pk_fma_f32",
                        "v_mov_b32_dpp quad", "v_pk_fma_f16",          "v_dot2c_f32_f16 dpp",  "v_fma_mix_f32 op_sel hi",
                        "ds_read_b128",       "-"};

template <int OP>
__global__ __launch_bounds__(1024) void rate_kernel(uint64_t* __restrict__ out, float* __restrict__ sink, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  float b0 = 0.5f, b1 = 0.25f;
  unsigned h = 0x3c003800u + threadIdx.x;  // two halves
  unsigned addr = (threadIdx.x * 64) & 65535u;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 r0, r1, r2, r3;
  double p0 = 1.0, p1 = 2.0, p2 = 3.0, p3 = 4.0;  // 64-bit register pairs for the packed f32 form
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##i) : "v"(b0), "v"(b1));
        REP8(X)
#undef X
      } else if (OP == FMA_MIX) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(a##i) : "v"(b0), "v"(h));
        REP8(X)
#undef X
      } else if (OP == FMA_MIX_HI) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(a##i) : "v"(b0), "v"(h));
        REP8(X)
#undef X
      } else if (OP == DOT2) {
#define X(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a##i) : "v"(h), "v"(h));
        REP8(X)
#undef X
      } else if (OP == DOT2_DPP) {
#define X(i) asm volatile("v_dot2c_f32_f16_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(a##i) : "v"(h), "v"(h));
        REP8(X)
#undef X
      } else if (OP == PK_FMA_F32) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p2), "v"(p3));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "v"(p2), "v"(p3));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p2), "v"(p3));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "v"(p2), "v"(p3));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p2), "v"(p3));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "v"(p2), "v"(p3));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p2), "v"(p3));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "v"(p2), "v"(p3));
      } else if (OP == DPP_MOV) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "=v"(a##i) : "v"(b0));
        REP8(X)
#undef X
      } else if (OP == PK_FMA_F16) {
#define X(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a##i) : "v"(h), "v"(h));
        REP8(X)
#undef X
      } else if (OP == DS_READ_B128) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(r1) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(r2) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(r3) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(r0) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(r1) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(r2) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:7168" : "=v"(r3) : "v"(addr));
      }
    }
    if (OP == DS_READ_B128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (OP == DS_READ_B128) a0 += r0[0] + r1[1] + r2[2] + r3[3];
  if (OP == PK_FMA_F32) a0 += (float)(p0 + p1);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) sink[0] = a0 + lds[threadIdx.x];
}

template <int OP>
void run(int waves_per_simd, uint64_t* d_out, float* d_sink) {
  const int threads = 256 * waves_per_simd, blocks = 256;
  hipMemset(d_out, 0, blocks * 16 * 8);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, 10);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, ITER);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  std::vector<uint64_t> h(blocks * 16);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc;
  for (int b_ = 0; b_ < blocks; ++b_)
    for (int w = 0; w < threads / 64; ++w) cyc.push_back((double)h[b_ * 16 + w]);
  std::sort(cyc.begin(), cyc.end());
  const double med = cyc[cyc.size() / 2];
  const double instr_per_wave = (double)ITER * 64;
  printf("%-26s %d waves/SIMD: %6.2f cycles per wave-instruction per wave, %5.2f per SIMD   (wall %7.1f us, clock %.2f GHz)\n",
         kNames[OP], waves_per_simd, med / instr_per_wave, med / instr_per_wave / waves_per_simd, ms * 1e3,
         med / (ms * 1e3) / 1e3);
}

int main() {
  uint64_t* d_out;
  float* d_sink;
  hipMalloc(&d_out, 256 * 16 * 8);
  hipMalloc(&d_sink, 64);
  for (int w : {1, 2, 4}) {
    run<FMA>(w, d_out, d_sink);
    run<FMA_MIX>(w, d_out, d_sink);
    run<FMA_MIX_HI>(w, d_out, d_sink);
    run<DOT2>(w, d_out, d_sink);
    run<DOT2_DPP>(w, d_out, d_sink);
    run<PK_FMA_F32>(w, d_out, d_sink);
    run<PK_FMA_F16>(w, d_out, d_sink);
    run<DPP_MOV>(w, d_out, d_sink);
    run<DS_READ_B128>(w, d_out, d_sink);
  }
  return 0;
}
