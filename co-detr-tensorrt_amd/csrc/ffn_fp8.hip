// Fused transformer FFN on the e4m3 matrix path of MI355X (gfx950) -- BASELINE config 5, second half:
//   Y = LNout( xn + relu(q(xn) . W1q^T * s1 * sx + b1) -> q(.) . W2q^T * s2 * sh + b2 ),   xn = LNin(X)   (+ pos)
// (reference codetr/transformer_mmcv.py:484-500 FFN inside the post-norm encoder layer :709-749: norm, ffn, norm, and
// the next layer's `query + query_pos`; no reference counterpart for the 8-bit arithmetic: its dtypes stop at half).
//
// Same dataflow as ffn_fused.hip -- persistent 256-thread workgroups walk 128-row tiles, each wave keeps its 32 rows of
// the input as MFMA B fragments and its 32 x 256 slice of Y in accumulators, the hidden activation never leaves the CU --
// on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales; twice the fp16 MFMA rate):
//   * the (LayerNorm'ed) input rows are quantised once, in registers: xq = sat(xn / sx), sx a static per-tensor scale;
//   * the hidden dimension is walked in chunks of 128: H^T[h][m] = W1q_c . xq^T (K = 256: two MFMAs per 16 x 16 tile),
//     then relu(acc s1 sx + b1) / sh as ONE fma and ONE median (s1 sx / sh and b1 / sh sit in LDS; the median is the ReLU
//     and the +448 saturation at once) -- and the four results a lane holds of each of the eight 16-row tiles are
//     exactly the 32 bytes of its B fragment for the second product (k-slot 4 t + r of lane group g = hidden unit
//     16 t + 4 g + r; the same permutation is baked into W2q once, on the host);
//     Y^T[n][m] += W2q_c . hq^T (K = 128: one MFMA per tile);
//   * W1q / W2q chunks (32 KiB each) stream through two 2-stage LDS rings by LDS-DMA (inline asm, scalar base +
//     per-thread offset), XOR-swizzled on the source address, one barrier in front of each product with a counted
//     vmcnt wait for pieces issued half a chunk earlier; the rings keep streaming across tiles;
//   * epilogue out of the accumulators: scale s2[n] sh, + b2 -> fp16, lanes 16 apart swap halves so that every access is
//     16 bytes per lane, + identity (the fp16 LayerNorm'ed input, rebuilt from X and the lane's row statistics),
//     LayerNorm, optionally also row + pos.  The next tile's rows are requested first and arrive meanwhile.
// fp32 accumulation throughout; e4m3 conversions saturate at +-448.  Timeline / ablations: profiles/r02_ffn_stamps.txt.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

constexpr int C = 256;
constexpr int BH = 128;                 // hidden units per chunk
constexpr int kThreads = 256;
constexpr int kW1Bytes = BH * C;        // 32 KiB: [128 h][256 k] e4m3
constexpr int kW2Bytes = C * BH;        // 32 KiB: [256 n][128 h] e4m3
constexpr int kRingBytes = 2 * (kW1Bytes + kW2Bytes);   // [A0 A1][B0 B1]
constexpr int kMaxHidden = 2048;        // per-hidden-unit scale' and bias' (fp32) in LDS: 16 KiB
constexpr float kFp8Max = 448.0f;
// diagnostic builds only (tools/micro/ffn8_ablate.hip): bit 0 no LDS-DMA in the loop, 1 no activation, 2 no product 2,
// 3 no product 1, 4 no barriers, 5 operands read from LDS once per chunk only.  Results are wrong with any bit set.
#ifndef FFN8_ABLATE
#define FFN8_ABLATE 0
#endif
constexpr int kAbl = FFN8_ABLATE;
#ifdef FFN8_STAMPS   // diagnostic build only: per-workgroup cycle sums of the loop's phases (tools/micro/ffn8_ablate.hip)
__device__ unsigned long long* g_ffn8_stamps = nullptr;
#define FFN8_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define FFN8_T(i) do { } while (0)
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_fp8x4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -kFp8Max, kFp8Max);
  b = __builtin_amdgcn_fmed3f(b, -kFp8Max, kFp8Max);
  c = __builtin_amdgcn_fmed3f(c, -kFp8Max, kFp8Max);
  d = __builtin_amdgcn_fmed3f(d, -kFp8Max, kFp8Max);
  int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned)v;
}

// chunk swizzle of a 128-byte LDS row that keeps the 32-byte fragment reads conflict-free under ds_read_b128's lane
// groups (see gemm_fp8.hip)
__device__ __forceinline__ int sw128(int row) {
  const int q = (row >> 1) & 7;
  return q ^ ((q & 2) << 1);
}

// one LDS-DMA piece: 256 threads x 16 B = 4 KiB.  `src` is wave-uniform (kernel argument + scalar offsets), `voff` the
// thread's byte offset, the same for every piece of an operand -- so a piece costs no vector arithmetic; `dst` uniform.
// Issued as inline assembly, not through __builtin_amdgcn_global_load_lds: the compiler's wait-count pass files the
// builtin with the out-of-order LDS traffic and from then on turns every wait for a ds_read into lgkmcnt(0) -- each MFMA
// group then waits for the operand reads issued just before it (for the tiles three steps ahead) and the read-ahead is
// void.  LDS-DMA only counts in vmcnt, which this kernel waits on by hand; with the instruction opaque the compiler
// emits counted lgkmcnt(N) waits.
__device__ __forceinline__ void dma16(const unsigned char* src, unsigned voff, unsigned char* dst) {
  const unsigned lds_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)dst);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds_addr) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// per-output-channel constants of the epilogue, one 64-byte record per 4 channels (n = 4 i .. 4 i + 3)
struct EpiRec {
  float s2[4];          // w2_scale[n] * h_scale
  _Float16 b2[4];
  _Float16 gin_w[4], gin_b[4];   // LayerNorm of the input (identity path)
  _Float16 g_w[4], g_b[4];       // LayerNorm of the output
  _Float16 pad[4];
};
static_assert(sizeof(EpiRec) == 64, "record layout");

// Persistent: gridDim.x workgroups (one per CU) walk the 128-row tiles blockIdx.x, + gridDim.x, ...; the W ring keeps
// streaming across tiles (every tile reads the same W), the next tile's rows are requested while the current tile's
// last chunks compute, and the epilogue works on the accumulators as they stand (no LDS staging) while the next
// tile's first W chunks arrive.
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_fp8_kernel(
    const unsigned short* __restrict__ X, const unsigned char* __restrict__ W1q, const float* __restrict__ s1,
    const unsigned short* __restrict__ b1, const unsigned char* __restrict__ W2q, const float* __restrict__ s2,
    const unsigned short* __restrict__ b2, unsigned short* __restrict__ Y, int M, int Hd, float sx, float sh,
    const unsigned short* __restrict__ ln_g, const unsigned short* __restrict__ ln_b, float ln_eps,
    const unsigned short* __restrict__ pos, unsigned short* __restrict__ Y2, const unsigned short* __restrict__ lnin_g,
    const unsigned short* __restrict__ lnin_b, float lnin_eps, int ntiles) {
  __shared__ __attribute__((aligned(64))) unsigned char lds[kRingBytes + kMaxHidden * 8 + 64 * 64];  // 148 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, grp = lane >> 4;
  constexpr int MT = 2, WR = 32;
  const int nchunks = Hd / BH;
  unsigned char* ringA = lds;                    // W1 chunks: stage i at ringA + i * kW1Bytes
  unsigned char* ringB = lds + 2 * kW1Bytes;     // W2 chunks
  float* sS1 = reinterpret_cast<float*>(lds + kRingBytes);
  float* sB1 = sS1 + kMaxHidden;
  EpiRec* sEpi = reinterpret_cast<EpiRec*>(lds + kRingBytes + kMaxHidden * 8);

  // LDS-DMA geometry.  W1 chunk image [128 rows][16 x 16 B]: piece p (0..7) = rows 16 p + (tid >> 4), position
  // tid & 15 holds source chunk (tid & 15) ^ (row & 15).  Packed-W2 chunk image [256 rows][8 x 16 B]: piece q = rows
  // 32 q + (tid >> 3), position tid & 7 holds source chunk (tid & 7) ^ sw128(row).  Both keys are piece-independent.
  const unsigned w1_voff = (unsigned)((tid >> 4) * C + (((tid & 15) ^ ((tid >> 4) & 15)) * 16));
  const unsigned w2_voff = (unsigned)((tid >> 3) * Hd + (((tid & 7) ^ sw128(tid >> 3)) * 16));
  auto stage_w1 = [&](int p, int c, unsigned char* dst) {
    dma16(W1q + (size_t)c * kW1Bytes + p * 4096, w1_voff, dst + (p * kThreads + wave * 64) * 16);
  };
  auto stage_w2 = [&](int q, int c, unsigned char* dst) {
    dma16(W2q + (size_t)q * 32 * Hd + c * BH, w2_voff, dst + (q * kThreads + wave * 64) * 16);
  };
#pragma unroll
  for (int p = 0; p < 8; ++p) stage_w1(p, 0, ringA);
#pragma unroll
  for (int p = 0; p < 8; ++p) stage_w2(p, 0, ringB);

  // a tile's input rows: lane (m = l15, g) holds X[m][128 kb + 32 g .. + 31], kb = 0, 1 (fp16, 4 x 16 B each)
  f16x8 xf[MT][2][4];
  auto load_x = [&](int tile) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int m = tile * 128 + wave * WR + mt * 16 + l15;
      m = m < M ? m : M - 1;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          xf[mt][kb][v] = *reinterpret_cast<const f16x8*>(X + (size_t)m * C + kb * 128 + grp * 32 + v * 8);
    }
  };
  load_x(blockIdx.x);

  // relu(acc * s1 * sx + b1) / sh = max(acc * s1' + b1', 0) with s1' = s1 * sx / sh, b1' = b1 / sh: one FMA and one
  // median (ReLU and the e4m3 saturation at once) per hidden value in the loop
  const float inv_sh = 1.0f / sh;
  for (int i = tid; i < Hd; i += kThreads) {
    sS1[i] = s1[i] * sx * inv_sh;
    sB1[i] = (float)__builtin_bit_cast(_Float16, b1[i]) * inv_sh;
  }
  if (tid < 64) {
    EpiRec r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = tid * 4 + j;
      r.s2[j] = s2[n] * sh;
      r.b2[j] = __builtin_bit_cast(_Float16, b2[n]);
      r.gin_w[j] = lnin_g ? __builtin_bit_cast(_Float16, lnin_g[n]) : (_Float16)1.f;
      r.gin_b[j] = lnin_g ? __builtin_bit_cast(_Float16, lnin_b[n]) : (_Float16)0.f;
      r.g_w[j] = ln_g ? __builtin_bit_cast(_Float16, ln_g[n]) : (_Float16)1.f;
      r.g_b[j] = ln_g ? __builtin_bit_cast(_Float16, ln_b[n]) : (_Float16)0.f;
      r.pad[j] = (_Float16)0.f;
    }
    sEpi[tid] = r;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // tables written; the first barrier below publishes them

  const float inv_sx = 1.0f / sx;
  int gc = 0;  // chunks consumed so far: ring stage = gc & 1
#ifdef FFN8_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
  const unsigned long long st_begin = st_last;
#endif
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * 128 + wave * WR;
    // ---- optional LayerNorm of the input rows (statistics over the four lanes that share a row), then quantisation
    // to e4m3: xq[mt][kb] = the B fragment (32 bytes) of k-block kb.  The lane keeps its rows' statistics: in the
    // accumulator layout of the second product the same lane column (l15) is the same row.
    i32x8 xq[MT][2];
    float mean_in[MT], rstd_in[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float mean = 0.f, rstd = 1.f;
      if (lnin_g) {
        float sm = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int e = 0; e < 8; ++e) sm += (float)xf[mt][kb][v][e];
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        mean = sm * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float d = (float)xf[mt][kb][v][e] - mean;
              q = fmaf(d, d, q);
            }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        rstd = rsqrtf(q * (1.0f / C) + lnin_eps);
      }
      mean_in[mt] = mean;
      rstd_in[mt] = rstd;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float f[8];
          if (lnin_g) {
            const f16x8 gw = *reinterpret_cast<const f16x8*>(lnin_g + kb * 128 + grp * 32 + v * 8);
            const f16x8 gb = *reinterpret_cast<const f16x8*>(lnin_b + kb * 128 + grp * 32 + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e)   // the fp16 value the separate LayerNorm kernel would have written, then / sx
              f[e] = (float)(_Float16)fmaf(((float)xf[mt][kb][v][e] - mean) * rstd, (float)gw[e], (float)gb[e]) * inv_sx;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)xf[mt][kb][v][e] * inv_sx;
          }
          xq[mt][kb][2 * v] = (int)pk_fp8x4(f[0], f[1], f[2], f[3]);
          xq[mt][kb][2 * v + 1] = (int)pk_fp8x4(f[4], f[5], f[6], f[7]);
        }
    }

    f32x4 yacc[16][MT];
#pragma unroll
    for (int nt = 0; nt < 16; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) yacc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Schedule of one chunk c (ring stage gc & 1; vmcnt counts LDS-DMA pieces, loads and stores in issue order):
    //   T: wait until W1[c] landed (the 8 younger pieces are W2[c]'s), barrier.  Not in a tile's chunk 0: its W1 / W2
    //     pieces are older than the tile's input rows, which the quantisation above has waited for -- and a counted
    //     wait there would also wait for the previous tile's output stores.
    //   product 1 over 4 pairs of hidden tiles, issuing the 8 pieces of W1[c+1]; the activation of pair p runs beside
    //     the MFMAs of pair p + 1
    //   M: wait until W2[c] landed (the 8 younger pieces are W1[c+1]'s), barrier; product 2's first operands are
    //     requested, then the last pair's activation runs while they arrive
    //   product 2 over 16 output tiles (operands 3 tiles ahead), issuing the 8 pieces of W2[c+1]
    // Overwrites are safe: W1[c+1] goes to the stage product 1 of chunk c-1 read (every wave passed M of c-1),
    // W2[c+1] to the one product 2 of chunk c-1 read (every wave passed T of c).  Chunk nchunks wraps to chunk 0 of
    // the next tile (past the last tile: a fetch nobody reads, drained at the end).
    for (int c = 0; c < nchunks; ++c, ++gc) {
      const int cn = c + 1 < nchunks ? c + 1 : 0;
      const unsigned char* sW1 = ringA + (gc & 1) * kW1Bytes;
      const unsigned char* sW2 = ringB + (gc & 1) * kW2Bytes;
      unsigned char* nW1 = ringA + ((gc + 1) & 1) * kW1Bytes;
      unsigned char* nW2 = ringB + ((gc + 1) & 1) * kW2Bytes;
      auto read_w1 = [&](int ht, int kb) -> i32x8 {
        const int row = ht * 16 + l15;
        const unsigned char* rp = sW1 + row * C;
        const i32x4 lo = *reinterpret_cast<const i32x4*>(rp + (((8 * kb + 2 * grp) ^ l15) * 16));
        const i32x4 hi = *reinterpret_cast<const i32x4*>(rp + (((8 * kb + 2 * grp + 1) ^ l15) * 16));
        return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      };
      auto read_w2 = [&](int nt) -> i32x8 {
        const int n = nt * 16 + l15;
        const unsigned char* rp = sW2 + n * BH;
        const int sw = sw128(l15);
        const i32x4 lo = *reinterpret_cast<const i32x4*>(rp + (((2 * grp) ^ sw) * 16));
        const i32x4 hi = *reinterpret_cast<const i32x4*>(rp + (((2 * grp + 1) ^ sw) * 16));
        return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      };
      i32x8 pf[MT];   // B fragments of the second product: VGPR t = the lane's 4 hidden units of tile t
      f32x4 h[2][2][MT];
      f32x4 sc[2][2], bb[2][2];
      // scale', bias', ReLU + saturation, e4m3: tiles 2 p and 2 p + 1 from h[p & 1]
      auto activate = [&](int p) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if constexpr (kAbl & 2) {
              pf[mt][2 * p + u] = __builtin_bit_cast(int, h[p & 1][u][mt][0]);
              continue;
            }
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
              v[r] = __builtin_amdgcn_fmed3f(fmaf(h[p & 1][u][mt][r], sc[p & 1][u][r], bb[p & 1][u][r]), 0.f, kFp8Max);
            int w = pf[mt][2 * p + u];   // (both halves are overwritten: no zero to materialise)
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
            pf[mt][2 * p + u] = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
          }
      };

      FFN8_T(c == 0 ? 0 : 4);
      if (c > 0) wait_vmcnt<8>();
      if constexpr (!(kAbl & 16)) __builtin_amdgcn_s_barrier();  // T
      FFN8_T(1);
      i32x8 a1[2][2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) a1[0][u][kb] = read_w1(u, kb);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (p + 1 < 4) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
              if constexpr (kAbl & 32) a1[(p + 1) & 1][u][kb] = a1[p & 1][u][kb];
              else a1[(p + 1) & 1][u][kb] = read_w1(2 * (p + 1) + u, kb);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int hu = c * BH + (2 * p + u) * 16 + grp * 4;
          sc[p & 1][u] = *reinterpret_cast<const f32x4*>(sS1 + hu);
          bb[p & 1][u] = *reinterpret_cast<const f32x4*>(sB1 + hu);
        }
        if constexpr (!(kAbl & 1)) {   // (spread over the phase: as one burst behind the barrier or at the end of the
          stage_w1(2 * p, cn, nW1);     //  phase the same 8 pieces cost 200 cycles more, profiles/r02_ffn_stamps.txt)
          stage_w1(2 * p + 1, cn, nW1);
        }
        // 4 independent accumulators, k-block 0 then k-block 1: no MFMA waits for the one before it
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              if constexpr (kAbl & 8) {
                if (kb == 0) h[p & 1][u][mt] = f32x4{(float)a1[p & 1][u][0][0], (float)a1[p & 1][u][1][1], 0.f, 0.f};
              } else {
                h[p & 1][u][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                    a1[p & 1][u][kb], xq[mt][kb], kb ? h[p & 1][u][mt] : f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
              }
            }
        if (p > 0) {
          activate(p - 1);
          // one wave per SIMD: the activation's ~60 vector ops are spread between the 8 MFMAs (32 cycles each)
          // instead of running as one block while the matrix pipe idles
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            if (g < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            if (g < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      FFN8_T(2);
      if (c > 0) wait_vmcnt<8>();
      if constexpr (!(kAbl & 16)) __builtin_amdgcn_s_barrier();  // M
      FFN8_T(3);
      i32x8 a2[4];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) a2[nt] = read_w2(nt);
      activate(3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < 16; ++nt) {
        if (nt + 3 < 16) {
          if constexpr (kAbl & 32) a2[(nt + 3) & 3] = a2[nt & 3];
          else a2[(nt + 3) & 3] = read_w2(nt + 3);
        }
        if constexpr (!(kAbl & 1))
          if (nt & 1) stage_w2(nt >> 1, cn, nW2);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if constexpr (kAbl & 4) yacc[nt][mt][0] += (float)(a2[nt & 3][0] ^ pf[mt][nt & 7]);
          else yacc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a2[nt & 3], pf[mt], yacc[nt][mt], 0, 0, 0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    FFN8_T(4);
    // ---- epilogue out of the accumulators (no LDS staging): lane (column l15 = row m of the tile, group g) holds
    // Y^T[n = 16 nt + 4 g + r][m].  y = yacc * s2' + b2 -> fp16 there; then lanes g and g ^ 1 (16 lanes apart) swap
    // halves so that every lane owns 8 CONSECUTIVE channels of 8 of the 16 tiles (g even: the even tiles, g odd: the
    // odd ones; g >> 1 picks channels 0-7 or 8-15 of the tile) -- every load and store below is then 16 bytes per lane
    // and a wave-instruction touches 64 contiguous bytes per row instead of 32 (the epilogue is bound by the number
    // of vector-memory instructions: 4-channel accesses took 57k cycles per tile, as long as the MFMA loop).
    // In that layout: + identity (the fp16 LayerNorm'ed input, rebuilt from X and the row statistics) -> fp16,
    // LayerNorm over the row (64 in-lane values + the four lanes of a row), + pos.
    // The next tile's input rows are requested first and arrive while this runs.
    {
      const int next = tile + (int)gridDim.x;
      load_x(next < ntiles ? next : tile);
    }
    const int odd = grp & 1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + l15;
      const int mc = m < M ? m : M - 1;
      const int cbase = 16 * odd + 8 * (grp >> 1);            // the lane's channels of pair j: 32 j + cbase .. + 7
      const unsigned short* xrow = X + (size_t)mc * C + cbase;
      f16x8 xr[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[j] = *reinterpret_cast<const f16x8*>(xrow + 32 * j);
      // y -> fp16 pairs in the accumulator layout
      unsigned yp[16][2];
#pragma unroll
      for (int nt = 0; nt < 16; ++nt) {
        const EpiRec* rec = sEpi + nt * 4 + grp;
        const f32x4 s2v = *reinterpret_cast<const f32x4*>(rec->s2);
        const f16x4 b2v = *reinterpret_cast<const f16x4*>(rec->b2);
        _Float16 y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = (_Float16)fmaf(yacc[nt][mt][r], s2v[r], (float)b2v[r]);
        yp[nt][0] = (unsigned)__builtin_bit_cast(unsigned short, y[0]) | ((unsigned)__builtin_bit_cast(unsigned short, y[1]) << 16);
        yp[nt][1] = (unsigned)__builtin_bit_cast(unsigned short, y[2]) | ((unsigned)__builtin_bit_cast(unsigned short, y[3]) << 16);
      }
      float o[8][8];
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // the partner takes this lane's half of the tile the partner owns, and gives its half of this lane's tile
        const unsigned s0 = odd ? yp[2 * j][0] : yp[2 * j + 1][0], s1 = odd ? yp[2 * j][1] : yp[2 * j + 1][1];
        const unsigned k0 = odd ? yp[2 * j + 1][0] : yp[2 * j][0], k1 = odd ? yp[2 * j + 1][1] : yp[2 * j][1];
        const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16, 64), r1 = (unsigned)__shfl_xor((int)s1, 16, 64);
        const unsigned z[4] = {odd ? r0 : k0, odd ? r1 : k1, odd ? k0 : r0, odd ? k1 : r1};   // channels cbase .. + 7
        const EpiRec* rec = sEpi + 8 * j + (cbase >> 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f16x4 giw = *reinterpret_cast<const f16x4*>(rec[q].gin_w);
          const f16x4 gib = *reinterpret_cast<const f16x4*>(rec[q].gin_b);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned w = z[2 * q + (r >> 1)];
            const _Float16 y = __builtin_bit_cast(_Float16, (unsigned short)((r & 1) ? (w >> 16) : (w & 0xffffu)));
            _Float16 x1 = xr[j][4 * q + r];
            if (lnin_g) x1 = (_Float16)fmaf(((float)x1 - mean_in[mt]) * rstd_in[mt], (float)giw[r], (float)gib[r]);
            o[j][4 * q + r] = (float)(_Float16)((float)y + (float)x1);
            sm += o[j][4 * q + r];
          }
        }
      }
      if (ln_g) {
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        const float mean = sm * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = o[j][e] - mean;
            q = fmaf(d, d, q);
          }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.0f / C) + ln_eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const EpiRec* rec = sEpi + 8 * j + (cbase >> 2);
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const f16x4 gw = *reinterpret_cast<const f16x4*>(rec[qq].g_w);
            const f16x4 gb = *reinterpret_cast<const f16x4*>(rec[qq].g_b);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              o[j][4 * qq + r] = (float)(_Float16)fmaf((o[j][4 * qq + r] - mean) * rstd, (float)gw[r], (float)gb[r]);
          }
        }
      }
      if (m < M) {
        unsigned short* yrow = Y + (size_t)m * C + cbase;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          f16x8 ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = (_Float16)o[j][e];
          *reinterpret_cast<f16x8*>(yrow + 32 * j) = ov;
        }
        if (Y2) {
          const unsigned short* prow = pos + (size_t)m * C + cbase;
          unsigned short* y2row = Y2 + (size_t)m * C + cbase;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const f16x8 pv = *reinterpret_cast<const f16x8*>(prow + 32 * j);
            f16x8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (_Float16)(o[j][e] + (float)pv[e]);
            *reinterpret_cast<f16x8*>(y2row + 32 * j) = ov;
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef FFN8_STAMPS
  FFN8_T(5);
  if (tid == 0 && g_ffn8_stamps) {
    unsigned long long* o = g_ffn8_stamps + 8 * (size_t)blockIdx.x;
    for (int i = 0; i < 6; ++i) o[i] = st_acc[i];
    o[6] = st_last - st_begin;
    o[7] = (unsigned long long)gc;
  }
#endif
}

}  // namespace

extern "C" {

int codetr_ffn_fp8(void* stream, const void* x_f16_dev, const void* w1q_dev, const float* w1_scale_dev,
                   const void* b1_f16_dev, const void* w2q_packed_dev, const float* w2_scale_dev, const void* b2_f16_dev,
                   void* y_f16_dev, int64_t M, int64_t C_in, int64_t hidden, float x_scale, float h_scale,
                   const void* ln_in_gamma_dev, const void* ln_in_beta_dev, float ln_in_eps, const void* ln_gamma_dev,
                   const void* ln_beta_dev, float ln_eps, const void* pos_dev, void* y_plus_pos_dev) {
  if (!x_f16_dev || !w1q_dev || !w1_scale_dev || !b1_f16_dev || !w2q_packed_dev || !w2_scale_dev || !b2_f16_dev ||
      !y_f16_dev || M <= 0 || hidden <= 0 || !(x_scale > 0.f) || !(h_scale > 0.f))
    return CODETR_E_BADARG;
  if ((ln_gamma_dev == nullptr) != (ln_beta_dev == nullptr) || (pos_dev == nullptr) != (y_plus_pos_dev == nullptr) ||
      (ln_in_gamma_dev == nullptr) != (ln_in_beta_dev == nullptr))
    return CODETR_E_BADARG;
  if (C_in != C || hidden % BH != 0 || hidden > kMaxHidden) return CODETR_E_UNSUPPORTED;
  // rows, weights and LayerNorm parameters are read / written 16 bytes at a time
  if ((reinterpret_cast<uintptr_t>(x_f16_dev) | reinterpret_cast<uintptr_t>(w1q_dev) | reinterpret_cast<uintptr_t>(w2q_packed_dev) |
       reinterpret_cast<uintptr_t>(y_f16_dev) | reinterpret_cast<uintptr_t>(ln_in_gamma_dev) |
       reinterpret_cast<uintptr_t>(ln_in_beta_dev) | reinterpret_cast<uintptr_t>(pos_dev) |
       reinterpret_cast<uintptr_t>(y_plus_pos_dev)) & 15)
    return CODETR_E_BADARG;
  if (M > 0x7fffffffLL - 256) return CODETR_E_TOO_LARGE;
  const int ntiles = (int)((M + 127) / 128);
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const unsigned blocks = (unsigned)(ntiles < cus ? ntiles : cus);   // persistent: one workgroup per CU
  hipLaunchKernelGGL(ffn_fp8_kernel, dim3(blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(x_f16_dev), static_cast<const unsigned char*>(w1q_dev), w1_scale_dev,
                     static_cast<const unsigned short*>(b1_f16_dev), static_cast<const unsigned char*>(w2q_packed_dev),
                     w2_scale_dev, static_cast<const unsigned short*>(b2_f16_dev), static_cast<unsigned short*>(y_f16_dev),
                     (int)M, (int)hidden, x_scale, h_scale, static_cast<const unsigned short*>(ln_gamma_dev),
                     static_cast<const unsigned short*>(ln_beta_dev), ln_eps, static_cast<const unsigned short*>(pos_dev),
                     static_cast<unsigned short*>(y_plus_pos_dev), static_cast<const unsigned short*>(ln_in_gamma_dev),
                     static_cast<const unsigned short*>(ln_in_beta_dev), ln_in_eps, ntiles);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // extern "C"
