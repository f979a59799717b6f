// Fused linear layer for MI355X (gfx950):  Y[M,N] = act(X[M,K] . W[N,K]^T + bias[N]) (+ R[M,N])
// fp16 / bf16 storage, fp32 accumulation on the matrix cores (v_mfma_f32_16x16x32_{f16,bf16}).
//
// This is the GEMM behind every nn.Linear of the Co-DETR hot path (Swin qkv / proj / MLP, encoder
// and decoder projections and FFNs, prediction heads: reference codetr/swin.py:91-115,
// codetr/transformer_mmcv.py:484-500, codetr/multi_scale_deformable_attention.py:173-213), with the
// bias add, ReLU/GELU and residual add that the reference runs as separate elementwise kernels
// folded into the epilogue.
//
// Structure (cdna_hip_programming.md section 5, the 128x128 LDS-staged form):
//   * one 256-thread workgroup (4 waves, 2x2) owns a 128(M) x 128(N) output tile; each wave a
//     64x64 quadrant = 4x4 MFMA tiles of 16x16, K-step 64 (2 MFMAs deep per tile).
//   * both operands are K-contiguous (X rows, W rows), so both tiles are staged the same way:
//     global -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) into a ring
//     of 2-4 LDS buffers; up to 3 K-tiles stay in flight across the one raw s_barrier per K step
//     (counted s_waitcnt vmcnt), 64 KiB of LDS in every configuration -> 2 workgroups per CU.
//   * LDS image = [128 rows][8 x 16-B chunks]; the DMA destination is lane-linear, so the
//     bank-conflict swizzle is applied on the SOURCE address and undone on the fragment read:
//     chunk position = chunk ^ ((row >> 1) & 7)  -> the 16 rows of a ds_read_b128 lane group
//     land on 16 distinct 16-B slots of the 256-B bank row.
//   * operands are swapped (MFMA "A" = W tile, "B" = X tile) so that the accumulator layout puts
//     4 consecutive n of one output row m in each lane; the epilogue applies bias + activation on
//     the accumulators, passes the tile through LDS once and stores whole 128-byte lines
//     (16 B per lane) with the residual added on the way out.
//   * workgroup -> tile order is XCD-aware: each XCD walks a contiguous run of tiles, n fastest,
//     so the X rows shared by the tiles of one m-row come from one L2.
//
// Requirements: K % 64 == 0, row pitches == K / N (dense row-major), 16-byte aligned bases.
// M and N are arbitrary (edge tiles clamp their loads and mask their stores).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "codetr_hip.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int kThreads = 256;
constexpr int kStagePitch = 64 * 2 + 16;            // epilogue staging: bytes per staged row of a wave's 64x64 quadrant
constexpr int kStagingBytes = 4 * 64 * kStagePitch;  // 36,864 B for the 4 waves

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HalfT {
  using frag = f16x8;
  using elem = _Float16;
  __device__ static s16x4 pack4(const float (&v)[4]) {  // 2 x v_cvt_pk_f16_f32 (round-to-nearest-even)
    f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    s16x4 o;
    __builtin_memcpy(&o, &h, 8);
    return o;
  }
  __device__ static f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  __device__ static float to_f32(unsigned short bits) {
    _Float16 h;
    __builtin_memcpy(&h, &bits, 2);
    return (float)h;
  }
  __device__ static unsigned short from_f32(float v) {
    _Float16 h = (_Float16)v;
    unsigned short bits;
    __builtin_memcpy(&bits, &h, 2);
    return bits;
  }
};
struct BFloatT {
  using frag = bf16x8;
  using elem = __bf16;
  // fp32 -> bf16 on the hardware converter (v_cvt_pk_bf16_f32, round to nearest even, NaN stays quiet): the 5-instruction
  // integer rounding of rounds 1-4 made every bf16 epilogue ~30 vector instructions per 4 outputs longer than its fp16 twin
  __device__ static s16x4 pack4(const float (&v)[4]) {
    typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
    const bf16x4v h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return __builtin_bit_cast(s16x4, h);
  }
  __device__ static f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  __device__ static float to_f32(unsigned short bits) { return __uint_as_float(((unsigned)bits) << 16); }
  __device__ static unsigned short from_f32(float v) {
    const __bf16 h = (__bf16)v;
    return __builtin_bit_cast(unsigned short, h);
  }
};

__device__ __forceinline__ unsigned xcd_tile(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// nn.GELU (erf form): 0.5 x (1 + erf(x / sqrt 2)).  libm's erff costs ~50 VALU ops per element and made the
// GELU epilogues VALU-bound; erf is evaluated with Abramowitz-Stegun 7.1.26 instead
// (|erf error| <= 1.5e-7, i.e. < 2^-22 relative on the output: three orders of magnitude below the fp16 / bf16
// rounding of the result).
__device__ __forceinline__ float gelu_erf(float x) {
  // 0.5 x (1 + erf(x / sqrt 2)) = 0.5 x + |x| (0.5 - (0.5 p(t) t) exp(-x^2 / 2)),  t = 1 / (1 + 0.3275911 |x| / sqrt 2):
  // the sign of erf folds into |x|, the halves into the coefficients -- 11 plain ops + v_rcp + v_exp (15 + 2 before)
  const float u = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, u, 1.0f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  const float ez = __builtin_amdgcn_exp2f(u * u * (-0.5f * 1.4426950408889634f));
  return fmaf(u, fmaf(-(p * t), ez, 0.5f), 0.5f * x);
}

// LDS image of one operand tile: [128 rows][CH = BKT/8 chunks of 16 B].  The DMA destination is lane-linear, so
// the bank swizzle lives on the SOURCE address: LDS position p of row r holds source chunk p ^ sw(r), and the
// fragment read applies the same XOR.  sw(r) spreads the 16 rows of a ds_read_b128 lane group over distinct
// 16-B slots of the 256-B bank row: (r>>1)&7 for 128-B rows, (r>>2)&3 for 64-B rows.
template <int BKT>
__device__ __forceinline__ int sw(int row) {
  // 64-byte rows: key {0, 3, 2, 1}[(row >> 2) & 3] -- the plain (row >> 2) & 3 is 2-way conflicted under ds_read_b128's
  // lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (tests/test_lds_bank_model.py); 128-byte rows are fine as they are
  return BKT == 64 ? (row >> 1) & 7 : ((row >> 2) & 3) ^ (((row >> 2) & 1) << 1);
}

// Stage one 128 x BKT operand tile (rows row0.. of a [rows_total, K] matrix, columns k0..k0+BKT-1) into LDS with
// BKT/16 LDS-DMA instructions per thread (16 B per lane each).
template <int BKT>
__device__ __forceinline__ void stage_tile(const unsigned short* __restrict__ src, int rows_total, int K, int row0, int k0,
                                           unsigned char* lds_tile, int tid) {
  constexpr int CH = BKT / 8;
  const int wave = tid >> 6;
#pragma unroll
  for (int q = 0; q < CH / 2; ++q) {
    const int u = q * kThreads + tid;
    const int r = u / CH, pos = u % CH;
    const int chunk = pos ^ sw<BKT>(r);
    int grow = row0 + r;
    grow = grow < rows_total ? grow : rows_total - 1;  // edge tiles: re-read the last row, results masked later
    const unsigned short* g = src + (size_t)grow * K + k0 + chunk * 8;
    // wave-uniform LDS base of this instruction; the hardware adds lane*16
    unsigned char* l = lds_tile + (q * kThreads + wave * 64) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  }
}

// Per-thread source pointer (at k = 0) of LDS-DMA piece q of an operand tile: the same addressing as stage_tile,
// computed once per kernel so that a piece inside the K loop is one pointer add and the DMA instruction.
template <int BKT>
__device__ __forceinline__ const unsigned short* piece_src(const unsigned short* __restrict__ src, int rows_total, int K,
                                                           int row0, int q, int tid) {
  constexpr int CH = BKT / 8;
  const int u = q * kThreads + tid;
  const int r = u / CH, pos = u % CH;
  const int chunk = pos ^ sw<BKT>(r);
  int grow = row0 + r;
  grow = grow < rows_total ? grow : rows_total - 1;
  return src + (size_t)grow * K + chunk * 8;
}

__device__ __forceinline__ void dma_piece(const unsigned short* g, unsigned char* lds_tile, int q, int wave) {
  unsigned char* l = lds_tile + (q * kThreads + wave * 64) * 16;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <class T, int BKT>
__device__ __forceinline__ typename T::frag read_frag(const unsigned char* lds_tile, int row, int chunk) {
  const int pos = chunk ^ sw<BKT>(row);
  return *reinterpret_cast<const typename T::frag*>(lds_tile + row * (BKT * 2) + pos * 16);
}

// one LDS-DMA piece (16 B per lane, 1 KiB per wave-instruction) with a wave-uniform source base in SGPRs and a per-thread
// byte offset; `dst` is the wave's uniform LDS destination.  Inline assembly: the compiler's wait-count pass files
// __builtin_amdgcn_global_load_lds with out-of-order LDS traffic and turns every later wait for a ds_read into
// lgkmcnt(0); the instruction itself only counts in vmcnt, which the callers wait on by hand.
__device__ __forceinline__ void lds_dma16s(const unsigned char* src, unsigned voff, unsigned char* dst) {
  const unsigned lds_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)dst);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds_addr) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_n() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N < 0, "add the count");
}

// ACT: 0 none, 1 relu, 2 gelu(erf)
// ACT: 0 none, 1 relu, 2 gelu(erf).  BKT x STAGES = the K pipeline: STAGES LDS buffers of one (W tile, X tile)
// pair each, STAGES-1 tiles of LDS-DMA in flight across the per-K-step barrier (raw s_barrier + counted
// s_waitcnt vmcnt: a __syncthreads() would drain the DMA queue, cdna_hip_programming.md section 5).
// SPLITK: blockIdx.y picks a range of `kps` K tiles; the block's fp32 partial tile goes to Y viewed as
// float[gridDim.y][M][N] (no bias / activation / residual: splitk_reduce_kernel applies them to the sum).
template <class T, int ACT, bool HAS_BIAS, bool HAS_RES, int BKT, int STAGES, bool SPLITK = false>
__global__ __launch_bounds__(kThreads) void linear_kernel(const unsigned short* __restrict__ X,
                                                          const unsigned short* __restrict__ W,
                                                          const unsigned short* __restrict__ bias,
                                                          const unsigned short* __restrict__ R,
                                                          unsigned short* __restrict__ Y,
                                                          const unsigned char* __restrict__ row_mask, int M, int N,
                                                          int K, int tiles_n, int hm_rows, int hm_hd, int kps) {
  constexpr int kTileBytes = 128 * BKT * 2;        // one operand tile
  constexpr int kStageBytes = 2 * kTileBytes;      // W tile + X tile
  constexpr int LPS = 2 * (BKT / 16);              // LDS-DMA instructions per thread per stage
  constexpr int kLdsBytes = STAGES * kStageBytes > kStagingBytes ? STAGES * kStageBytes : kStagingBytes;
  static_assert(kLdsBytes <= 64 * 1024, "pipeline does not fit 64 KiB");
  static_assert(STAGES >= 2 && STAGES <= 4, "");
  // one object only (a second __shared__ object de-pipelines LDS-DMA kernels); the epilogue staging reuses it
  __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsBytes];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;  // wave quadrant inside the 128x128 tile

  const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // The accumulators start at the bias (D = A.B + C with C = bias broadcast along m): the bias add costs nothing
  // and its load latency hides under the first K stage.  A lane's 4 registers of tile (i, j) are the 4 consecutive
  // output columns n = wn*64 + i*16 + 4*(lane>>4) + r, the same for every j.
  const int ncol = 4 * (lane >> 4);
  const bool vec_n = (N & 7) == 0;
  f32x4 acc[4][4];  // [n-tile][m-tile]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (HAS_BIAS) {
      const int n = n0 + wn * 64 + i * 16 + ncol;
      if (vec_n) {
        if (n < N) {
          const s16x4 bb = *reinterpret_cast<const s16x4*>(bias + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) b4[r] = T::to_f32((unsigned short)bb[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) b4[r] = T::to_f32(bias[n + r]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = b4;
  }

  const int kt0 = SPLITK ? (int)blockIdx.y * kps : 0;
  const int nk = SPLITK ? (K / BKT - kt0 < kps ? K / BKT - kt0 : kps) : K / BKT;
  const int frow = lane & 15, fchunk = lane >> 4;
  constexpr int KS = BKT / 32;
  {
  auto issue = [&](int t) {
    unsigned char* buf = lds + (t % STAGES) * kStageBytes;
    stage_tile<BKT>(W, N, K, n0, (kt0 + t) * BKT, buf, tid);
    stage_tile<BKT>(X, M, K, m0, (kt0 + t) * BKT, buf + kTileBytes, tid);
  };
#pragma unroll
  for (int s0 = 0; s0 < STAGES - 1; ++s0)
    if (s0 < nk) issue(s0);

  for (int t = 0; t < nk; ++t) {
    // tiles issued beyond t: t+1 .. min(nk-1, t+STAGES-2); tile t itself must have landed
    const int ahead = (nk - 1 < t + STAGES - 2 ? nk - 1 : t + STAGES - 2) - t;
    if (STAGES >= 4 && ahead == 2) wait_vmcnt<2 * LPS>();
    else if (STAGES >= 3 && ahead == 1) wait_vmcnt<LPS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // everyone's part of tile t is in LDS; everyone is done reading tile t-1
    if (t + STAGES - 1 < nk) issue(t + STAGES - 1);  // overwrites the buffer of tile t-1
    const unsigned char* bufW = lds + (t % STAGES) * kStageBytes;
    const unsigned char* bufX = bufW + kTileBytes;
    typename T::frag a[2][4], b[2][4];
    auto read_frags = [&](int ks, int buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[buf][i] = read_frag<T, BKT>(bufW, wn * 64 + i * 16 + frow, ks * 4 + fchunk);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[buf][j] = read_frag<T, BKT>(bufX, wm * 64 + j * 16 + frow, ks * 4 + fchunk);
    };
    read_frags(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) read_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = T::mfma(a[ks & 1][i], b[ks & 1][j], acc[i][j]);
      if (ks + 1 < KS) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
    }
  }
  }
  __builtin_amdgcn_s_barrier();  // all fragment reads done (no DMA is in flight any more): LDS is free for the epilogue

  if (SPLITK) {
    float* part = reinterpret_cast<float*>(Y) + (size_t)blockIdx.y * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + ncol;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + frow;
        if (m >= M || n >= N) continue;
        if (vec_n) {
          *reinterpret_cast<f32x4*>(part + (size_t)m * N + n) = acc[i][j];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < N) part[(size_t)m * N + n + r] = acc[i][j][r];
        }
      }
    }
    return;
  }

  // ---- epilogue -------------------------------------------------------------------------------
  // accumulator layout: for MFMA tile (i, j) a lane holds n = wn*64 + i*16 + 4*(lane>>4) + r (r = 0..3),
  // m = wm*64 + j*16 + (lane&15), i.e. 8 contiguous output bytes per lane and 32-byte row segments per
  // store instruction.  Partial-line stores of that shape run at ~1 TB/s, so the tile goes through LDS
  // once and leaves as whole 128-byte lines (16 B per lane): bias + activation in fp32 on the
  // accumulators -> fp16 image of the wave's 64x64 quadrant in its private LDS region -> read back by
  // rows, add the residual (fp16 + fp16 -> fp16, the same two roundings as `identity + linear(x)`
  // in the reference's fp16 path) -> global_store_dwordx4.
  if (vec_n) {
    constexpr int kPitch = kStagePitch;  // bytes per staged row (+16: rows 0/8 do not share a bank pair)
    unsigned char* stage = lds + wave * (64 * kPitch);  // main loop is done with LDS (barrier above)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[i][j][r];
          if (ACT == 1) x = x < 0.f ? 0.f : x;  // NaN-propagating, like torch.relu
          if (ACT == 2) x = gelu_erf(x);
          v[r] = x;
        }
        *reinterpret_cast<s16x4*>(stage + (j * 16 + frow) * kPitch + (i * 16 + ncol) * 2) = T::pack4(v);
      }
    }
    __builtin_amdgcn_wave_barrier();  // same wave writes then reads: DS ops retire in order
    const int srow = lane >> 3, schunk = lane & 7;
    const int n = n0 + wn * 64 + schunk * 8;
    // residual rows, all requested before the first use (one exposed latency instead of eight); the head-major
    // destination never carries a residual
    s16x8 rres[8];
    if (HAS_RES) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        int m = m0 + wm * 64 + it * 8 + srow;
        m = m < M ? m : M - 1;
        const int nn = n + 8 <= N ? n : (N >= 8 ? N - 8 : 0);
        rres[it] = *reinterpret_cast<const s16x8*>(R + (size_t)m * N + nn);
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int ml = it * 8 + srow;
      const int m = m0 + wm * 64 + ml;
      if (m < M && n < N) {
        s16x8 v = *reinterpret_cast<const s16x8*>(stage + ml * kPitch + schunk * 16);
        if (row_mask) {
          const unsigned char mk = row_mask[m];
          if (mk == 2) {
            // the INPUT row counts as zeros (`memory * keep` ahead of enc_output): y = act(bias)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float x = HAS_BIAS ? T::to_f32(bias[n + e]) : 0.f;
              if (ACT == 1) x = x < 0.f ? 0.f : x;
              if (ACT == 2) x = gelu_erf(x);
              v[e] = (short)T::from_f32(x);
            }
          } else if (mk) {
            v = s16x8{0, 0, 0, 0, 0, 0, 0, 0};  // masked_fill(mask[..., None], 0) on the linear's output
          }
        }
        size_t off = (size_t)m * N + n;
        if (hm_hd > 0) {
          // head-major destination y[b][head][position][channel]: rows m = (b, position), columns n = (head, channel);
          // an 8-column chunk never straddles heads (hm_hd % 8 == 0), 8 consecutive rows of one head are 8*hm_hd*2
          // contiguous bytes
          const int bb = m / hm_rows, pos = m - bb * hm_rows;
          const int head = n / hm_hd, ch = n - head * hm_hd;
          off = (((size_t)bb * (N / hm_hd) + head) * hm_rows + pos) * hm_hd + ch;
        }
        if (HAS_RES) {
          const s16x8 rr = rres[it];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] = (short)T::from_f32(T::to_f32((unsigned short)v[e]) + T::to_f32((unsigned short)rr[e]));
        }
        *reinterpret_cast<s16x8*>(Y + off) = v;
      }
    }
    return;
  }
  // ragged N (N % 8 != 0: the 4-wide box head, test shapes): direct stores from the accumulator layout
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + wn * 64 + i * 16 + ncol;
    if (n >= N) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * 64 + j * 16 + frow;
      if (m >= M) continue;
      const size_t off = (size_t)m * N + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r < N) {
          float v = acc[i][j][r];
          if (ACT == 1) v = v < 0.f ? 0.f : v;
          if (ACT == 2) v = gelu_erf(v);
          unsigned short h = T::from_f32(v);
          if (row_mask && row_mask[m]) {
            h = 0;
            if (row_mask[m] == 2) {
              float x = HAS_BIAS ? T::to_f32(bias[n + r]) : 0.f;
              if (ACT == 1) x = x < 0.f ? 0.f : x;
              if (ACT == 2) x = gelu_erf(x);
              h = T::from_f32(x);
            }
          }
          if (HAS_RES) h = T::from_f32(T::to_f32(h) + T::to_f32(R[off + r]));
          Y[off + r] = h;
        }
      }
    }
  }
}

// Pipeline configuration per problem (A/B on MI355X over the model's 22 layer shapes, tools/bench_linear.py):
//   K <= 256 (every 256-wide transformer layer, Swin stage 0): 32-deep K steps, 2 buffers -> 36 KiB of LDS, 3
//     workgroups per CU.  These layers run 4-8 K steps per tile; their cost is the per-tile fixed latency
//     (first loads, bias, LDS staging, stores), which only more resident workgroups hide.
//   K  > 256: 64-deep steps, 2 buffers (64 KiB, 2 workgroups per CU): twice the MFMA work per barrier.
//   A 4-deep ring (3 tiles of DMA in flight, counted vmcnt) measured 5-10 % slower than either on every shape, and
//   a persistent-grid variant that prefetched the next tile's first stage under the epilogue 20 % slower: ablation
//   (stores off / K loop off) shows the short-K layers bound by bytes through the CU's vector-memory path --
//   L2->LDS operand re-reads (3.35 GB at ~17 TB/s for the encoder FFN up-projection) + staging + stores -- not by
//   exposed latency.  The lever that remains is a larger tile (fewer operand re-reads); see DESIGN.md section 4.
//   Grids of at most one workgroup per CU (the decoder's 900-query layers: 16 tiles) gain nothing from occupancy;
//     their time is the chain of per-K-step DMA latencies, so they take the 64-deep step (half the steps) as well.
// (The A/B overrides of this choice -- a 4-deep ring, the DMA pieces spread over the MFMAs -- measured slower on every
// shape in rounds 1-3 and are gone from the library; profiles/r02_*, r03_gemm256_ablation.txt keep the numbers.)
int pipeline_cfg(int64_t K, int64_t tiles) { return (K <= 256 && tiles > 256) ? 322 : 642; }

template <class T, int ACT, int BKT, int STAGES>
int launch_cfg(hipStream_t st, const void* X, const void* W, const void* bias, const void* R, void* Y,
               const void* mask, int M, int N, int K, int hm_rows, int hm_hd) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const dim3 grid((unsigned)(tiles_m * tiles_n)), block(kThreads);
  auto x = static_cast<const unsigned short*>(X);
  auto w = static_cast<const unsigned short*>(W);
  auto b = static_cast<const unsigned short*>(bias);
  auto r = static_cast<const unsigned short*>(R);
  auto y = static_cast<unsigned short*>(Y);
  auto mk = static_cast<const unsigned char*>(mask);
  if (bias && R) hipLaunchKernelGGL((linear_kernel<T, ACT, true, true, BKT, STAGES, false>), grid, block, 0, st, x, w, b, r, y, mk, M, N, K, tiles_n, hm_rows, hm_hd, 0);
  else if (bias) hipLaunchKernelGGL((linear_kernel<T, ACT, true, false, BKT, STAGES, false>), grid, block, 0, st, x, w, b, r, y, mk, M, N, K, tiles_n, hm_rows, hm_hd, 0);
  else if (R) hipLaunchKernelGGL((linear_kernel<T, ACT, false, true, BKT, STAGES, false>), grid, block, 0, st, x, w, b, r, y, mk, M, N, K, tiles_n, hm_rows, hm_hd, 0);
  else hipLaunchKernelGGL((linear_kernel<T, ACT, false, false, BKT, STAGES, false>), grid, block, 0, st, x, w, b, r, y, mk, M, N, K, tiles_n, hm_rows, hm_hd, 0);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

// y = act(sum_z part[z] + bias) (masked rows -> 0) + residual, 4 columns per thread
template <class T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part,
                                                            const unsigned short* __restrict__ bias,
                                                            const unsigned short* __restrict__ R,
                                                            const unsigned char* __restrict__ row_mask,
                                                            unsigned short* __restrict__ Y, int M, int N, int splits,
                                                            int act) {
  const int ngroups = (N + 3) / 4;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)M * ngroups) return;
  const int m = (int)(idx / ngroups), n = (int)(idx % ngroups) * 4;
  const size_t off = (size_t)m * N + n;
  const bool vec = (N & 3) == 0;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < splits; ++z) {
    const float* p = part + (size_t)z * M * N + off;
    if (vec) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += q[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) v[r] += p[r];
    }
  }
  const unsigned char mk = row_mask ? row_mask[m] : 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (n + r >= N) break;
    float x = (mk == 2 ? 0.f : v[r]) + (bias ? T::to_f32(bias[n + r]) : 0.f);
    if (act == 1) x = x < 0.f ? 0.f : x;
    if (act == 2) x = gelu_erf(x);
    unsigned short h = (mk != 0 && mk != 2) ? (unsigned short)0 : T::from_f32(x);
    if (R) h = T::from_f32(T::to_f32(h) + T::to_f32(R[off + r]));
    Y[off + r] = h;
  }
}

// Split-K plan: problems whose 128x128 output tiles cannot fill the chip but whose K is long (the neck's extra
// 3x3/s2 level as a GEMM: 600 x 256 x 13824 = 10 tiles, 183 us in one pass) are cut along K into `splits` ranges of
// whole 64-wide K tiles so that ~1.5 blocks per CU are in flight.  Returns 1 when a single pass is the better launch.
int splitk_plan(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN), ktiles = K / 64;
  if (tiles > 128 || K < 2048 || K % 64 != 0) return 1;
  // 65 ... 128 tiles (Swin stage-2 fc2 of a 608x608 image: 72 tiles x 48 k-tiles; stage-3 fc2 of a 1152x768 image: 84 x 96):
  // as many whole passes over K as still fit one round of the chip
  int64_t splits = tiles > 64 ? 256 / tiles : (384 + tiles - 1) / tiles;
  if (splits > ktiles / 4) splits = ktiles / 4;
  if (splits < 2) return 1;
  const int64_t kps = (ktiles + splits - 1) / splits;
  return (int)((ktiles + kps - 1) / kps);
}

template <class T>
int launch_splitk(hipStream_t st, const void* X, const void* W, const void* bias, const void* R, void* Y,
                  const void* mask, int64_t M, int64_t N, int64_t K, int act, int splits, void* ws, int64_t ws_bytes) {
  if (!X || !W || !Y || !ws || M <= 0 || N <= 0 || K <= 0 || splits < 2) return CODETR_E_BADARG;
  if (K % 64 != 0 || act < 0 || act > 2) return CODETR_E_UNSUPPORTED;
  if (M > 0x7fffffffLL || N > 0x7fffffffLL || K > 0x7fffffffLL || splits > 65535) return CODETR_E_TOO_LARGE;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(ws)) & 15)
    return CODETR_E_BADARG;
  if (ws_bytes < (int64_t)splits * M * N * 4) return CODETR_E_BADARG;
  const int ktiles = (int)(K / 64), kps = (ktiles + splits - 1) / splits;
  if ((int64_t)kps * (splits - 1) >= ktiles) return CODETR_E_BADARG;  // an empty last range
  const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
  const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)splits), block(kThreads);
  hipLaunchKernelGGL((linear_kernel<T, 0, false, false, 64, 2, true>), grid, block, 0, st,
                     static_cast<const unsigned short*>(X), static_cast<const unsigned short*>(W), nullptr, nullptr,
                     static_cast<unsigned short*>(ws), nullptr, (int)M, (int)N, (int)K, tiles_n, 0, 0, kps);
  const long groups = (long)M * ((N + 3) / 4);
  hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st,
                     static_cast<const float*>(ws), static_cast<const unsigned short*>(bias),
                     static_cast<const unsigned short*>(R), static_cast<const unsigned char*>(mask),
                     static_cast<unsigned short*>(Y), (int)M, (int)N, splits, act);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

template <class T, int ACT>
int launch_act(hipStream_t st, const void* X, const void* W, const void* bias, const void* R, void* Y,
               const void* mask, int M, int N, int K, int hm_rows, int hm_hd) {
  switch (pipeline_cfg(K, (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN))) {
    case 322: return launch_cfg<T, ACT, 32, 2>(st, X, W, bias, R, Y, mask, M, N, K, hm_rows, hm_hd);
    default: return launch_cfg<T, ACT, 64, 2>(st, X, W, bias, R, Y, mask, M, N, K, hm_rows, hm_hd);
  }
}

// ------------------------------------------------------------------------------------------------------------
// 256 x 256 tile for the large layers (Swin stages 2-3 at 8 images per GPU: M = 77-81 k rows).  A 128 x 128 x 64 tile
// step moves 32 KB of operands through the CU's vector-memory path per 2.1 MFLOP -- 64 B/clk per CU at MFMA peak,
// which IS that path's rate, so the tiled kernel above tops out near 0.9 PF while hipBLASLt's 256-wide macro tiles
// reach 1.1-1.4 PF on these shapes.  Here: 512 threads = 8 waves (2 along m x 4 along n), each wave 128(m) x 64(n) =
// 8 x 4 MFMA tiles (128 accumulator registers); one workgroup per CU (2 x 64 KiB of LDS: W tile 256 x 64 + X tile
// 256 x 64 per stage); 32 B/clk per CU of operand traffic and 12 fragment reads per 32 MFMAs.  Same LDS image,
// swizzle, bias-in-accumulator and LDS-staged epilogue (in two 64-row halves) as linear_kernel.
// Ablation at 8 images (M = 80 640, us): main loop only / full kernel = qkv 287 / 379, proj+res 87 / 161, fc1+GELU
// 314 / 520, fc2+res 355 / 412 -- the main loop runs at 1.0-1.15 PF (hipBLASLt: 1.1-1.2 PF including its store), the
// epilogue costs 25-40 % on top.  Two attempts to hide it, both measured slower and removed: (a) a 256 x 128 / 4-wave /
// BK 32 variant with two workgroups per CU so that one's epilogue overlaps the other's MFMAs (423 vs 384 us on qkv:
// the main loop loses more than the overlap returns); (b) a persistent variant that parks the finished fp16 tile in
// 96 KiB of LDS + 16 registers and writes it back one store per K step of the next tile, with the stores allowed to
// stay in flight across barriers (450 vs 376 us: the write-back costs the same trickled as in bulk, so it is not
// issue latency -- most likely the 372 MB of output passing through the L2 / Infinity Cache evicts the activation
// rows every column tile re-reads).  Next thing to try: non-temporal stores with a column-major tile walk.
#ifdef CODETR_GEMM_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#endif
// diagnostic builds only (tools/micro/gemm256_stamps.hip -DCODETR_GEMM_ABL=mask; WRONG results by construction):
// 1 = no LDS-DMA inside the main loop, 2 = no MFMAs, 4 = no fragment reads, 8 = no wait + barrier per k-tile
#ifdef CODETR_GEMM_ABL
constexpr int kGemmAbl = CODETR_GEMM_ABL;
#else
constexpr int kGemmAbl = 0;
#endif
// XDEEP: the X operand (the one that misses L2 more: each 256-row slice is shared by the N/256 column tiles only, the W
// slices by every row tile) is prefetched TWO k-tiles ahead through a 3-slot ring that takes the last 32 KiB of the CU's
// 160 KiB of LDS; W stays one tile ahead in its 2-slot ring.  In-kernel stamps (tools/micro/gemm256_stamps.hip) show
// the 2-stage loop waiting for its LDS-DMA ~48 % of the time at 8 images (4300 cycles per k-tile against 2050 of MFMA
// work): the operand fetch takes ~2 us under load and only one tile time was there to hide it.
template <class T, int ACT, bool HAS_BIAS, bool HAS_RES, bool XDEEP = false, bool SDMA = false>
__global__ __launch_bounds__(512) void linear_256_kernel(const unsigned short* __restrict__ X,
                                                         const unsigned short* __restrict__ W,
                                                         const unsigned short* __restrict__ bias,
                                                         const unsigned short* __restrict__ R,
                                                         unsigned short* __restrict__ Y,
                                                         const unsigned char* __restrict__ row_mask, int M, int N, int K,
                                                         int tiles_n) {
  constexpr int BKT = 64, NT = 512;
#ifdef CODETR_GEMM_STAMPS   // diagnostic build only (tools/micro/gemm256_stamps.hip): in-kernel timeline of every workgroup
  unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;
  if (threadIdx.x == 0) {
    st0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
#endif
  constexpr int kTileBytes = 256 * BKT * 2;   // 32 KiB: one operand tile
  constexpr int kStageBytes = 2 * kTileBytes;  // W tile + X tile
  // 2-stage ring: [W0 X0][W1 X1] (128 KiB).  XDEEP: [X0 X1 X2][W0 W1] (160 KiB, the whole LDS of the CU)
  __shared__ __attribute__((aligned(16))) unsigned char lds[XDEEP ? 5 * kTileBytes : 2 * kStageBytes];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int frow = lane & 15, fchunk = lane >> 4, ncol = 4 * (lane >> 4);

  f32x4 acc[4][8];  // [n-tile][m-tile]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (HAS_BIAS) {
      const int n = n0 + wn * 64 + i * 16 + ncol;
      if (n < N) {  // N % 8 == 0 on this path
        const s16x4 bb = *reinterpret_cast<const s16x4*>(bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) b4[r] = T::to_f32((unsigned short)bb[r]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = b4;
  }

  // per-thread source pointers of the 4 + 4 LDS-DMA pieces of a stage (piece q covers rows q*64 + tid/8)
  const unsigned short* gw[4];
  const unsigned short* gx[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = q * 64 + (tid >> 3), pos = tid & 7;
    const int chunk = pos ^ sw<BKT>(r);
    int gn = n0 + r, gm = m0 + r;
    gn = gn < N ? gn : N - 1;
    gm = gm < M ? gm : M - 1;
    gw[q] = W + (size_t)gn * K + chunk * 8;
    gx[q] = X + (size_t)gm * K + chunk * 8;
  }
  // SDMA: the same pieces addressed as (wave-uniform 64-bit base of the tile's first row and k-tile, in SGPRs) + (per-thread
  // 32-bit byte offset, fixed for the whole tile): the LDS-DMA is then one inline-assembly instruction per piece -- no
  // 64-bit vector add, no v_readfirstlane for its LDS destination, eight address registers less (the pointer form spilled
  // two registers that were reloaded inside the loop, and a scratch reload counts in vmcnt: the counted wait then asked
  // for half of the tile that was meant to stay in flight)
  unsigned woff[4], xoff[4];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned char* Wt = reinterpret_cast<const unsigned char*>(W) + (size_t)n0 * K * 2;
  const unsigned char* Xt = reinterpret_cast<const unsigned char*>(X) + (size_t)m0 * K * 2;
  if (SDMA) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = q * 64 + (tid >> 3), pos = tid & 7;
      const int chunk = pos ^ sw<BKT>(r);
      const int rn = n0 + r < N ? r : N - 1 - n0, rm = m0 + r < M ? r : M - 1 - m0;   // edge tiles: the last row again
      woff[q] = (unsigned)((rn * K + chunk * 8) * 2);
      xoff[q] = (unsigned)((rm * K + chunk * 8) * 2);
    }
  }
  auto sdmaW = [&](int q, int kt, unsigned char* slot) {
    lds_dma16s(Wt + (size_t)kt * (BKT * 2), woff[q], slot + (q * NT + wave_u * 64) * 16);
  };
  auto sdmaX = [&](int q, int kt, unsigned char* slot) {
    lds_dma16s(Xt + (size_t)kt * (BKT * 2), xoff[q], slot + (q * NT + wave_u * 64) * 16);
  };
  auto issue = [&](int t, int slot) {
    unsigned char* buf = lds + (slot & 1) * kStageBytes;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned char* l = buf + (q * NT + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw[q] + (size_t)t * BKT),
                                       (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned char* l = buf + kTileBytes + (q * NT + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gx[q] + (size_t)t * BKT),
                                       (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
  };
  const int nk = K / BKT;
  // k-tile index of loop step i (steps past the end re-fetch the last one).  A per-tile rotation of this walk (so that
  // tiles sharing an operand slice do not ask for it at the same moment) was measured neutral to -6 % and removed
  // (profiles/r03_gemm256_ablation.txt).
  auto ktile = [&](int i) { return i < nk ? i : nk - 1; };
  auto dma16 = [&](const unsigned short* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  };
  if (XDEEP && SDMA) {
#pragma unroll
    for (int q = 0; q < 4; ++q) sdmaW(q, ktile(0), lds + 3 * kTileBytes);
#pragma unroll
    for (int q = 0; q < 4; ++q) sdmaX(q, ktile(0), lds);
#pragma unroll
    for (int q = 0; q < 4; ++q) sdmaX(q, ktile(1), lds + kTileBytes);
  } else if (XDEEP) {   // issue order W(0), X(0), X(1): the youngest four pieces may stay in flight at the first wait
    const size_t k0 = (size_t)ktile(0) * BKT;
#pragma unroll
    for (int q = 0; q < 4; ++q) dma16(gw[q] + k0, lds + 3 * kTileBytes + (q * NT + wave * 64) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) dma16(gx[q] + k0, lds + (q * NT + wave * 64) * 16);
    const size_t k1 = (size_t)ktile(1) * BKT;
#pragma unroll
    for (int q = 0; q < 4; ++q) dma16(gx[q] + k1, lds + kTileBytes + (q * NT + wave * 64) * 16);
  } else {
    issue(ktile(0), 0);
  }
#ifdef CODETR_GEMM_STAMPS
  unsigned long long wait_dma = 0, wait_bar = 0;
#endif
  int xs = 0;  // XDEEP: ring slot of X(t)
  for (int t = 0; t < nk; ++t) {
#ifdef CODETR_GEMM_STAMPS
    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
    if (XDEEP) wait_vmcnt<4>(); else wait_vmcnt<0>();
    const unsigned long long w1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    const unsigned long long w2 = __builtin_amdgcn_s_memtime();
    wait_dma += w1 - w0;
    wait_bar += w2 - w1;
#else
    if (!(kGemmAbl & 8)) {
      if (XDEEP) wait_vmcnt<4>();   // W(t) and X(t) have landed; the four pieces of X(t+1) may still be in flight
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();  // tile t is in LDS for everyone; everyone is done reading tile t-1
    }
#endif
    const unsigned char* bufW = XDEEP ? lds + (3 + (t & 1)) * kTileBytes : lds + (t & 1) * kStageBytes;
    const unsigned char* bufX = XDEEP ? lds + xs * kTileBytes : bufW + kTileBytes;
    // One workgroup per CU: its 8 waves reach this point together, so nothing else covers a burst of DMA issue or an
    // LDS wait.  The schedule is pinned: fragments of k-step 1 are read behind the first two MFMAs of step 0, and the
    // 8 DMA pieces of tile t+1 go out one per 3 MFMAs of step 0 (early enough to land under step 1).  Past the last
    // tile the pieces re-fetch it into the idle buffer (no branch in the pinned region); drained before the epilogue.
    const size_t koff = (size_t)ktile(t + 1) * BKT;
    unsigned char* nbuf = lds + ((t + 1) & 1) * kStageBytes;
    // XDEEP: W(t+1) -> the W slot tile t-1 used, X(t+2) -> the X slot tile t-1 used (re-fetches of the last tile past
    // the end keep the in-flight count uniform)
    const size_t koffx = (size_t)ktile(t + 2) * BKT;
    unsigned char* nbufW = XDEEP ? lds + (3 + ((t + 1) & 1)) * kTileBytes : nbuf;
    const int xs2 = xs >= 1 ? xs - 1 : 2;   // (xs + 2) % 3
    unsigned char* nbufX = XDEEP ? lds + xs2 * kTileBytes : nbuf + kTileBytes;
    xs = xs == 2 ? 0 : xs + 1;
#ifdef CODETR_GEMM_ABL
    typename T::frag a[2][4] = {}, b[2][8] = {};
#else
    typename T::frag a[2][4], b[2][8];
#endif
    auto read_frags = [&](int ks, int buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[buf][i] = read_frag<T, BKT>(bufW, wn * 64 + i * 16 + frow, ks * 4 + fchunk);
#pragma unroll
      for (int j = 0; j < 8; ++j) b[buf][j] = read_frag<T, BKT>(bufX, wm * 128 + j * 16 + frow, ks * 4 + fchunk);
    };
    if (SDMA) {
      // explicit interleave (the inline-assembly pieces are opaque to sched_group_barrier): step-0 fragments, the first two
      // MFMAs, the step-1 fragments, then eight groups of (3 MFMAs, 1 piece) -- W(t+1) first, it is needed one tile from
      // now, X(t+2) two -- each closed by a scheduling barrier; then the remaining 6 + 32 MFMAs
      read_frags(0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
      read_frags(1, 1);
      auto mf0 = [&](int idx) {
        const int j = idx >> 2, i = idx & 3;
        acc[i][j] = T::mfma(a[0][i], b[0][j], acc[i][j]);
      };
      mf0(0);
      mf0(1);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
      __builtin_amdgcn_sched_barrier(0);
      const int ktw = ktile(t + 1), ktx = ktile(t + 2);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        mf0(2 + 3 * g);
        mf0(3 + 3 * g);
        mf0(4 + 3 * g);
        if (g < 4) sdmaW(g, ktw, nbufW);
        else sdmaX(g - 4, ktx, nbufX);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int idx = 26; idx < 32; ++idx) mf0(idx);
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = T::mfma(a[1][i], b[1][j], acc[i][j]);
      __builtin_amdgcn_sched_group_barrier(0x008, 38, 0);
      continue;
    }
    if (!(kGemmAbl & 4)) read_frags(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
    // ---- k-step 0 ----
    if (!(kGemmAbl & 4)) read_frags(1, 1);
    if (kGemmAbl & 1) {
    } else if (XDEEP) {   // W first: it is needed one tile from now, X two
#pragma unroll
      for (int q = 0; q < 4; ++q) dma16(gw[q] + koff, nbufW + (q * NT + wave * 64) * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) dma16(gx[q] + koffx, nbufX + (q * NT + wave * 64) * 16);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        dma16(gw[q] + koff, nbufW + (q * NT + wave * 64) * 16);
        dma16(gx[q] + koff, nbufX + (q * NT + wave * 64) * 16);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!(kGemmAbl & 2)) acc[i][j] = T::mfma(a[0][i], b[0][j], acc[i][j]);
        else if (!(kGemmAbl & 4)) asm volatile("" ::"v"(a[0][i]), "v"(b[0][j]));
      }
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    // ---- k-step 1 ----
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!(kGemmAbl & 2)) acc[i][j] = T::mfma(a[1][i], b[1][j], acc[i][j]);
        else if (!(kGemmAbl & 4)) asm volatile("" ::"v"(a[1][i]), "v"(b[1][j]));
      }
    __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
  }
  wait_vmcnt<0>();  // the redundant pieces of the last iteration have landed
  __builtin_amdgcn_s_barrier();  // all fragment reads done, no DMA in flight: LDS is free for the epilogue
#ifdef CODETR_GEMM_STAMPS
  if (threadIdx.x == 0) st1 = __builtin_amdgcn_s_memtime();
#endif

  // epilogue in two 64-row halves per wave through its private staging region (64 rows x 144 B)
  constexpr int kPitch = kStagePitch;
  unsigned char* stage = lds + wave * (64 * kPitch);
  const int srow = lane >> 3, schunk = lane & 7;
  const int n = n0 + wn * 64 + schunk * 8;
  unsigned char* ytile = reinterpret_cast<unsigned char*>(Y) + ((size_t)m0 * N + n0) * 2;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[i][h * 4 + j][r];
          if (ACT == 1) x = x < 0.f ? 0.f : x;
          if (ACT == 2) x = gelu_erf(x);
          v[r] = x;
        }
        *reinterpret_cast<s16x4*>(stage + (j * 16 + frow) * kPitch + (i * 16 + ncol) * 2) = T::pack4(v);
      }
    __builtin_amdgcn_wave_barrier();
    // residual rows of this half, all requested before the first use (one exposed latency instead of eight)
    s16x8 rres[8];
    if (HAS_RES) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        int m = m0 + wm * 128 + h * 64 + it * 8 + srow;
        m = m < M ? m : M - 1;
        const int nn = n + 8 <= N ? n : N - 8;
        rres[it] = *reinterpret_cast<const s16x8*>(R + (size_t)m * N + nn);
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int ml = it * 8 + srow;
      const int m = m0 + wm * 128 + h * 64 + ml;
      if (m < M && n < N) {
        s16x8 v = *reinterpret_cast<const s16x8*>(stage + ml * kPitch + schunk * 16);
        if (row_mask) {
          const unsigned char mk = row_mask[m];
          if (mk == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float x = HAS_BIAS ? T::to_f32(bias[n + e]) : 0.f;
              if (ACT == 1) x = x < 0.f ? 0.f : x;
              if (ACT == 2) x = gelu_erf(x);
              v[e] = (short)T::from_f32(x);
            }
          } else if (mk) {
            v = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
          }
        }
        if (HAS_RES) {
          const s16x8 rr = rres[it];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] = (short)T::from_f32(T::to_f32((unsigned short)v[e]) + T::to_f32((unsigned short)rr[e]));
        }
        // uniform 64-bit tile base (SGPRs) + 32-bit lane offset: the store carries half the address bytes
        const unsigned loff = ((unsigned)(wm * 128 + h * 64 + ml) * (unsigned)N + (unsigned)(wn * 64 + schunk * 8)) * 2u;
        *reinterpret_cast<s16x8*>(ytile + loff) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the region is rewritten by the second half
  }
#ifdef CODETR_GEMM_STAMPS
  if (threadIdx.x == 0 && g_stamps) {
    st2 = __builtin_amdgcn_s_memtime();   // all stores of wave 0 issued (not completed)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();   // ... and completed
    unsigned long long* o = g_stamps + 8 * (size_t)blockIdx.x;
    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();
    o[6] = wait_dma; o[7] = wait_bar;
  }
#endif
}

// the 256-tile kernel: K >= 256 (K = 256 layers whose N fills 256-wide tiles -- value / output
// projections, enc_output -- measured 5-15 % faster here than on the X-stationary kernel), see the rule at the end
bool big_applicable(int64_t M, int64_t N, int64_t K, int hm_hd) {
  if (hm_hd != 0 || K < 256 || K % 64 != 0 || N % 8 != 0) return false;
  const int64_t tn = (N + 255) / 256, tm = (M + 255) / 256;
  // at least ~0.8 tiles per CU (200 measured better than 512 for the single-image shapes: Swin stage-2 fc1 78 -> 67 us,
  // stage-3 qkv 58 -> 46 us), and little of the 256-wide tile wasted: >= 87.5 % of the tile columns real,
  // or >= 75 % when one column tile covers N (X is then read exactly once: Swin stage-0 fc2, N = 192, 806 -> 678 us)
  return tm * tn >= 200 && (N * 8 >= tn * 256 * 7 || (tn == 1 && N * 4 >= 256 * 3));
}

template <class T, int ACT>
int launch_big(hipStream_t st, const void* X, const void* W, const void* bias, const void* R, void* Y, const void* mask,
               int M, int N, int K) {
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
  const dim3 grid((unsigned)(tiles_m * tiles_n)), block(512);
  auto x = static_cast<const unsigned short*>(X);
  auto w = static_cast<const unsigned short*>(W);
  auto b = static_cast<const unsigned short*>(bias);
  auto r = static_cast<const unsigned short*>(R);
  auto y = static_cast<unsigned short*>(Y);
  auto mk = static_cast<const unsigned char*>(mask);
  // X two k-tiles ahead in a 3-slot ring (8 images: -3 ... -13 % on every Swin stage 1-3 shape); scalar-base LDS-DMA for
  // problems of more than one column tile (-1 ... -2.5 %; a single column tile -- N = 192 -- measured 6 % slower with it
  // and keeps the pointer form)
  const bool sdma = tiles_n > 1;
#define CODETR_L256S(HB, HR) \
  hipLaunchKernelGGL((linear_256_kernel<T, ACT, HB, HR, true, true>), grid, block, 0, st, x, w, b, r, y, mk, M, N, K, tiles_n)
#define CODETR_L256(HB, HR) \
  hipLaunchKernelGGL((linear_256_kernel<T, ACT, HB, HR, true, false>), grid, block, 0, st, x, w, b, r, y, mk, M, N, K, tiles_n)
  if (sdma) {
    if (bias && R) CODETR_L256S(true, true);
    else if (bias) CODETR_L256S(true, false);
    else if (R) CODETR_L256S(false, true);
    else CODETR_L256S(false, false);
  } else {
    if (bias && R) CODETR_L256(true, true);
    else if (bias) CODETR_L256(true, false);
    else if (R) CODETR_L256(false, true);
    else CODETR_L256(false, false);
  }
#undef CODETR_L256
#undef CODETR_L256S
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

// ------------------------------------------------------------------------------------------------------------
// X-stationary kernel for the short-K layers (K = 192, 256, 384: Swin stages 0-1 and every 256-wide transformer
// layer -- a third of the model's GEMM time).  The tiled kernel above re-reads each 128 x K activation tile from L2
// once per 128-column output tile and pays its fixed per-tile cost (first loads, staging, stores) N/128 times; with
// K this short the whole K extent of 32 rows fits a wave's registers.  So, as in the first product of
// ffn_fused.hip: a 256-thread workgroup owns 128 rows for ALL N; each wave loads its 32 rows of X once, as MFMA B
// fragments straight from global memory (KS k-steps x 2 m-tiles), and keeps them; W streams through a 2-stage LDS
// ring in chunks of 32 output columns (32 x K, LDS-DMA, XOR-swizzled on the source address), D[n][m] = W . X^T, so
// a lane again owns 4 consecutive n of one row.  Per chunk: bias from LDS into the accumulator init, 16*KS/8 MFMAs
// per wave, activation, fp16 image of the wave's 32 x 32 piece in its private LDS region, row-wise read-back,
// residual (requested before the MFMAs) / row mask, 16-B stores.  X is read from HBM exactly once, Y written once,
// W (<= 1.2 MB) comes from L2.
// diagnostic builds only (tools/micro/build_variant.sh ... "-DCODETR_XS_ABL=mask", WRONG results by construction): 1 = no output
// stores, 2 = no MFMAs, 4 = no W chunk staging inside the loop, 8 = no wait / barrier per chunk, 16 = no X / X2 loads
#ifndef CODETR_XS_ABL
#define CODETR_XS_ABL 0
#endif
#ifdef CODETR_XS_STAMPS   // diagnostic build only (tools/micro/xs_stamps.hip): where wave 0 of every workgroup spends its cycles
__device__ unsigned long long* g_xs_stamps = nullptr;
#define XS_STAMP(i) xs_t[i] = __builtin_readcyclecounter()
#else
#define XS_STAMP(i)
#endif
//
// SPLIT (the encoder's two projections of one token row as ONE launch -- reference multi_scale_deformable_attention.py:161-179:
// value_proj(value) and sampling_offsets | attention_weights (query + query_pos), where value IS query): W holds N1 rows
// applied to X and N - N1 rows applied to X + X2; the first N1 columns go to Y (type OT, row mask, optional head-major
// layout), the others to Y2 (type T, row-major, N - N1 columns).  The kept fragments become X + X2 at chunk N1 / 32: X and
// X2 are each read once for both products (two launches read X twice: 419 MB of 2.5 GB at four 1920x1280 images).
// POSGEN (SPLIT only, round 6): X2 is the encoder's sine positional encoding + level embedding -- a pure function of the
// token's (level, y, x) and of the running sums of the padding mask (reference positional_encoding.py:58-93 +
// transformer.py:508-519, csrc/sine_pos.hip) -- and is GENERATED here instead of read: the lane re-derives its row's two
// normalised coordinates from the running sums (8 bytes per token instead of the row's 512) and evaluates its 64 channels
// with the SAME fp32 operations, in the same order, as sine_pos_kernel, so the operand x + pos is bit-identical to the one
// read from the tensor that kernel writes.  Saves the 64 KiB of pos rows per 128-row tile and the exposed wait for them at
// the start of the second product.
struct XsPosGen {
  const float* ycum[5];                 // per level: running sums of the not-mask along y, [B, H_l, W_l] fp32
  const float* xcum[5];                 // ... along x
  const unsigned short* level_embed;    // [L, 256] in the operands' type (nullptr: none)
  int H[5], W[5], start[5];             // level l holds tokens start[l] .. start[l] + H W - 1 of an image's S
  int L, S;
  float log2_temperature, scale, eps, offset;
  int normalize;
};

template <class T, int KS, int ACT, bool HAS_RES, class OT = T, bool SPLIT = false, bool POSGEN = false>   // OT: storage type of Y (bf16 operands -> fp16 output: the
__global__ __launch_bounds__(256) void linear_xs_kernel(const unsigned short* __restrict__ X,   // encoder MSDA's value map)
                                                        const unsigned short* __restrict__ X2,
                                                        const unsigned short* __restrict__ LNG,
                                                        const unsigned short* __restrict__ LNB, float ln_eps,
                                                        const unsigned short* __restrict__ W,
                                                        const unsigned short* __restrict__ bias,
                                                        const unsigned short* __restrict__ R,
                                                        unsigned short* __restrict__ Y,
                                                        const unsigned char* __restrict__ row_mask, int M, int N, int hm_rows,
                                                        int hm_hd, unsigned short* __restrict__ Y2 = nullptr, int N1 = 0,
                                                        const XsPosGen pg = XsPosGen()) {
  constexpr int K = KS * 32, CH = K / 8;          // 16-byte chunks per row
  constexpr int CN = 32;                          // output columns per W chunk
  constexpr int kChunkBytes = CN * K * 2;         // 12 / 16 / 24 KiB
  constexpr int kPieces = kChunkBytes / 4096;     // LDS-DMA instructions per thread per chunk (3 / 4 / 6)
  constexpr int kXsPitch = 2 * CN * 2 + 16;       // staged output row: two chunks side by side, 128 B + 16
  constexpr int kStage = 4 * 32 * kXsPitch;       // 18 KiB: 4 waves x 32 rows
  constexpr int kMaxBias = 1568;
  constexpr int SWZ = (CH % 16 == 0) ? 15 : 7;    // XOR mask that keeps a swizzled chunk inside its row
  constexpr int kRing = 3;                        // W chunks in flight: c (being multiplied), c + 1, c + 2
  constexpr int kLeBytes = POSGEN ? 5 * K * 2 : 0;   // the level embeddings (POSGEN)
  __shared__ __attribute__((aligned(16))) unsigned char lds[kRing * kChunkBytes + kStage + kMaxBias * 2 + kLeBytes];
  unsigned char* stage_base = lds + kRing * kChunkBytes;
  unsigned short* sBias = reinterpret_cast<unsigned short*>(lds + kRing * kChunkBytes + kStage);
  unsigned short* sLe = sBias + kMaxBias;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, grp = lane >> 4;
  const int m0 = (int)xcd_tile(blockIdx.x, gridDim.x) * 128 + wave * 32;
  const int nchunks = (N + CN - 1) / CN;
#ifdef CODETR_XS_STAMPS
  unsigned long long xs_t[8], xs_acc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long xs_start = __builtin_readcyclecounter(), xs_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int Ny = SPLIT ? N1 : N;                  // columns of Y (SPLIT: the others are Y2's)
  const int c1 = SPLIT ? N1 / CN : 0;             // first chunk of the second product

  // LDS-DMA of one W chunk (32 rows x K, kPieces pieces of 4 KiB): piece q covers rows (q * 256 + tid) / CH.  Inline
  // assembly with a wave-uniform base and a per-thread byte offset (see lds_dma16s): no vector arithmetic per piece but
  // the row clamp of a ragged last chunk, and the compiler keeps counting its ds_read waits.
  unsigned piece_row[kPieces], piece_off[kPieces];
#pragma unroll
  for (int q = 0; q < kPieces; ++q) {
    const int u = q * 256 + tid;
    const int r = u / CH, pos = u % CH;
    piece_row[q] = (unsigned)r;
    piece_off[q] = (unsigned)((pos ^ (r & SWZ)) * 16);
  }
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(W);
  auto stage_chunk = [&](int c, unsigned char* dst) {
    const unsigned rmax = (unsigned)(N - 1 - c * CN);  // ragged last chunk: re-read the last row, its columns are never stored
#pragma unroll
    for (int q = 0; q < kPieces; ++q) {
      const unsigned r = piece_row[q] < rmax ? piece_row[q] : rmax;
      lds_dma16s(Wb + (size_t)c * kChunkBytes, r * (unsigned)(K * 2) + piece_off[q], dst + (q * 256 + wave * 64) * 16);
    }
  };
  stage_chunk(0, lds);
  stage_chunk(nchunks > 1 ? 1 : 0, lds + kChunkBytes);   // (a single chunk: a second fetch nobody reads keeps the counts uniform)

  // this wave's 32 rows of X, B-operand fragments: lane (j = l15, g = grp) holds X[m][32*ks + 8g .. +7]
  typename T::frag xf[2][KS];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int m = m0 + mt * 16 + l15;
    m = m < M ? m : M - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (CODETR_XS_ABL & 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[mt][ks][e] = (typename T::elem)(float)(m + ks);
        continue;
      }
      xf[mt][ks] = *reinterpret_cast<const typename T::frag*>(X + (size_t)m * K + ks * 32 + grp * 8);
    }
    // optional second input, added element-wise on the way in (x + x2 rounded to T, the fp16 / bf16 add the host
    // would otherwise run as its own kernel: `query + query_pos` in front of the offsets | logits projection)
    if (X2 && !SPLIT && !(CODETR_XS_ABL & 16)) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const typename T::frag p = *reinterpret_cast<const typename T::frag*>(X2 + (size_t)m * K + ks * 32 + grp * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[mt][ks][e] = (typename T::elem)((float)xf[mt][ks][e] + (float)p[e]);
      }
    }
    // optional LayerNorm of the input rows (the whole row is in this wave: K values in the four lanes l15 + 16 g), fp32
    // two-pass statistics, result rounded to T = the operand a separate LayerNorm kernel would have written; used
    // where nothing else reads that normalised tensor (Swin's pre-norm blocks: norm1 -> qkv, norm2 -> fc1)
    if (LNG) {
      float sm = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += (float)xf[mt][ks][e];
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      const float mean = sm * (1.0f / K);
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)xf[mt][ks][e] - mean;
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const float rstd = rsqrtf(q * (1.0f / K) + ln_eps);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const typename T::frag gw = *reinterpret_cast<const typename T::frag*>(LNG + ks * 32 + grp * 8);
        const typename T::frag gb = *reinterpret_cast<const typename T::frag*>(LNB + ks * 32 + grp * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          xf[mt][ks][e] = (typename T::elem)fmaf(((float)xf[mt][ks][e] - mean) * rstd, (float)gw[e], (float)gb[e]);
      }
    }
  }
  // POSGEN: this lane's two rows -> level, normalised coordinates (the running sums are requested here, next to the rows)
  float pg_ey[2] = {0.f, 0.f}, pg_ex[2] = {0.f, 0.f};
  int pg_lv[2] = {0, 0};
  if constexpr (POSGEN) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      int m = m0 + mt * 16 + l15;
      m = m < M ? m : M - 1;
      const int b = m / pg.S, tok = m - b * pg.S;
      int l = 0;
#pragma unroll
      for (int k = 1; k < 5; ++k) l += (k < pg.L && tok >= pg.start[k]) ? 1 : 0;
      int Hl = pg.H[0], Wl = pg.W[0], st = pg.start[0];
      const float* yc = pg.ycum[0];
      const float* xc = pg.xcum[0];
#pragma unroll
      for (int k = 1; k < 5; ++k) {
        const bool me = l == k;
        Hl = me ? pg.H[k] : Hl;
        Wl = me ? pg.W[k] : Wl;
        st = me ? pg.start[k] : st;
        yc = me ? pg.ycum[k] : yc;
        xc = me ? pg.xcum[k] : xc;
      }
      const int r = tok - st, y = r / Wl, x = r - y * Wl;
      const int64_t base = (int64_t)b * Hl * Wl;
      float ey = yc[base + r], ex = xc[base + r];
      if (pg.normalize) {   // exactly sine_pos_kernel's expression
        const float ly = yc[base + (int64_t)(Hl - 1) * Wl + x], lx = xc[base + (int64_t)y * Wl + (Wl - 1)];
        ey = (ey + pg.offset) / (ly + pg.eps) * pg.scale;
        ex = (ex + pg.offset) / (lx + pg.eps) * pg.scale;
      }
      pg_ey[mt] = ey;
      pg_ex[mt] = ex;
      pg_lv[mt] = l;
    }
    if (pg.level_embed)
      for (int i = tid; i < pg.L * K / 8; i += 256)
        *reinterpret_cast<s16x8*>(sLe + i * 8) = *reinterpret_cast<const s16x8*>(pg.level_embed + i * 8);
  }
  if (bias) {
    for (int i = tid; i < N / 8; i += 256)
      *reinterpret_cast<s16x8*>(sBias + i * 8) = *reinterpret_cast<const s16x8*>(bias + i * 8);
  }
  // the row states of the lane's four output rows (m0 + 8 it + (lane >> 3)), fetched ONCE here: read inside the chunk loop
  // they sat behind the LDS-DMA pieces in flight (vector memory returns in order) -- + 16-19 % on the decoder's value
  // projection (924 -> 1 100 us at 4 images)
  unsigned row_states = 0;
  if (row_mask) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int m = m0 + it * 8 + (lane >> 3);
      row_states |= (m < M ? (unsigned)row_mask[m] : 0u) << (8 * it);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  unsigned char* my_stage = stage_base + wave * (32 * kXsPitch);
  // Outputs leave two chunks at a time: 64 columns = 128 contiguous bytes per row, i.e. whole cache lines (a single
  // 32-column chunk would write half lines and leave the merge to the L2).  Read-back of a pair: 8 rows x 8 chunks of
  // 16 B per step, 4 steps.
  const int srow = lane >> 3, schunk = lane & 7;
  s16x8 rr[4];
  // residual rows of the pair that starts at column np (requested one chunk before they are added)
  auto load_residual = [&](int np) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int m = m0 + it * 8 + srow;
      m = m < M ? m : M - 1;
      int n = np + schunk * 8;
      n = n + 8 <= N ? n : N - 8;
      rr[it] = *reinterpret_cast<const s16x8*>(R + (size_t)m * N + n);
    }
  };
  // the staged pair that starts at column np: + residual / row mask, 16-byte stores of whole 128-byte row segments
  auto flush_pair = [&](int np) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int ml = it * 8 + srow;
      const int m = m0 + ml;
      const int n = np + schunk * 8;
      s16x8 v = *reinterpret_cast<const s16x8*>(my_stage + ml * kXsPitch + schunk * 16);
      if (SPLIT && np >= N1) {   // (N1 % 64 == 0: a pair belongs to one of the two outputs)
        if (m < M && n < N) *reinterpret_cast<s16x8*>(Y2 + (size_t)m * (N - N1) + (n - N1)) = v;
        continue;
      }
      if (m < M && n < Ny) {  // N % 8 == 0: a chunk of 8 columns is inside or outside as a whole
        if (row_mask) {
          const unsigned char mk = (unsigned char)((row_states >> (8 * it)) & 0xffu);
          if (mk == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float x = bias ? T::to_f32(sBias[n + e]) : 0.f;
              if (ACT == 1) x = x < 0.f ? 0.f : x;
              if (ACT == 2) x = gelu_erf(x);
              v[e] = (short)OT::from_f32(x);
            }
          } else if (mk) {
            v = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
          }
        }
        if (HAS_RES) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] = (short)T::from_f32(T::to_f32((unsigned short)v[e]) + T::to_f32((unsigned short)rr[it][e]));
        }
        size_t off = (size_t)m * Ny + n;
        if (hm_hd > 0) {  // column-block-major destination y[b][n / hm_hd][position][n % hm_hd] (hm_hd % 8 == 0: a lane's 8-column chunk lies inside one block)
          const int bb = m / hm_rows, pos = m - bb * hm_rows;
          const int head = n / hm_hd, ch = n - head * hm_hd;
          off = (((size_t)bb * (Ny / hm_hd) + head) * hm_rows + pos) * hm_hd + ch;
        }
        if (!(CODETR_XS_ABL & 1) || v[0] == 0x1234) *reinterpret_cast<s16x8*>(Y + off) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the staging region is rewritten by the next pair
  };
  // Schedule (vmcnt counts LDS-DMA pieces, loads and stores in issue order).  Chunk c:
  //   wait until W[c] landed: everything but the kPieces pieces of W[c+1], the youngest operations, is complete --
  //     including the stores of the pair flushed a whole chunk earlier, so nobody ever waits for a store just issued
  //     (the 2-stage form of this loop waited `vmcnt(0)` right behind its own stores: 2.8 us per chunk for 0.25 us of
  //     MFMAs); barrier: W[c] is visible, everyone is done with W[c-1]
  //   even c >= 2: flush the pair (c-2, c-1) staged by the previous two chunks; odd c: request that pair's residual rows
  //   issue W[c+2] into the stage W[c-1] used
  //   32 x 32 outputs per wave: MFMAs, activation, fp16 image into the wave's staging region
  int slot = 0;  // ring slot of W[c]
#ifdef CODETR_XS_STAMPS
  const unsigned long long xs_loop = __builtin_readcyclecounter();
#endif
  for (int c = 0; c < nchunks; ++c) {
    XS_STAMP(0);
    if (!(CODETR_XS_ABL & 8)) wait_vmcnt_n<kPieces>();
    XS_STAMP(1);
    if (!(CODETR_XS_ABL & 8)) __builtin_amdgcn_s_barrier();
    XS_STAMP(2);
    const int n0 = c * CN;
    if (!(c & 1)) {
      if (c >= 2) flush_pair((c - 2) * CN);
    } else if (HAS_RES) {
      load_residual((c - 1) * CN);
    }
    XS_STAMP(3);
    {
      const int c2 = c + 2 < nchunks ? c + 2 : nchunks - 1;  // (past the end: a fetch nobody reads, same counts)
      const int s2 = slot == 0 ? 2 : slot - 1;               // the slot W[c-1] used
      if (!(CODETR_XS_ABL & 4)) stage_chunk(c2, lds + s2 * kChunkBytes);
    }
    XS_STAMP(4);
    if (SPLIT && c == c1) {
      // the second product's operand: x + x2 rounded to T (exactly the separate add); its loads are the youngest vector
      // memory operations and are waited for here, so the counted wait of the next chunk sees the W pieces only
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        int m = m0 + mt * 16 + l15;
        m = m < M ? m : M - 1;
        if constexpr (POSGEN) {
          // lane (row l15, group g) holds channels 32 ks + 8 g .. + 7 = 16-byte chunk c = 4 ks + g of the row: ks < KS / 2 the y
          // half, the others the x half; chunk (c mod 16) covers frequencies 4 (c mod 16) .. + 3 (channel 2f: sin, 2f + 1: cos).
          // One fragment at a time, added as it is made (eight live fragments cost the second workgroup of the CU its registers).
          static_assert(!POSGEN || KS == 8, "256 channels");
          constexpr int num_feats = K / 2;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            typename T::frag pfk;
            const float e_ = ks >= KS / 2 ? pg_ex[mt] : pg_ey[mt];
            const int ch0 = (4 * (ks & 3) + grp) * 8;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const int f = (ch0 >> 1) + p;
              const float inv = __builtin_amdgcn_exp2f(-pg.log2_temperature * (2.0f * (float)f / (float)num_feats));
              const float rev = e_ * inv * 0.15915494309189535f;
              pfk[2 * p] = (typename T::elem)__builtin_amdgcn_sinf(rev);
              pfk[2 * p + 1] = (typename T::elem)__builtin_amdgcn_cosf(rev);
            }
            if (pg.level_embed) {
              const typename T::frag le = *reinterpret_cast<const typename T::frag*>(sLe + pg_lv[mt] * K + (4 * ks + grp) * 8);
#pragma unroll
              for (int k = 0; k < 8; ++k) pfk[k] = (typename T::elem)((float)pfk[k] + (float)le[k]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[mt][ks][e] = (typename T::elem)((float)xf[mt][ks][e] + (float)pfk[e]);
            asm volatile("" : "+v"(pg_ey[mt]), "+v"(pg_ex[mt]));   // (keeps the frequency factors from being hoisted out of the loops)
          }
        } else {
          typename T::frag pf[KS];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            pf[ks] = *reinterpret_cast<const typename T::frag*>(X2 + (size_t)m * K + ks * 32 + grp * 8);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[mt][ks][e] = (typename T::elem)((float)xf[mt][ks][e] + (float)pf[ks][e]);
        }
      }
    }
    const unsigned char* sW = lds + slot * kChunkBytes;
    slot = slot == 2 ? 0 : slot + 1;

    f32x4 acc[2][2];  // [n-tile][m-tile]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if (bias) {
        const s16x4 bv = *reinterpret_cast<const s16x4*>(sBias + n0 + nt * 16 + grp * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) b4[r] = T::to_f32((unsigned short)bv[r]);
      }
      acc[nt][0] = b4;
      acc[nt][1] = b4;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      typename T::frag a[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int row = nt * 16 + l15;
        const int chunk = (ks * 4 + grp) ^ (row & SWZ);
        a[nt] = *reinterpret_cast<const typename T::frag*>(sW + row * (K * 2) + chunk * 16);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (CODETR_XS_ABL & 2) acc[nt][mt][0] += (float)a[nt][0] * (float)xf[mt][ks][0];
          else acc[nt][mt] = T::mfma(a[nt], xf[mt][ks], acc[nt][mt]);
        }
    }
    // ---- chunk epilogue: this wave's 32 rows x 32 columns into its half of the staged pair ----
#ifdef CODETR_XS_STAMPS
    asm volatile("" ::"v"(acc[1][1][3]) : "memory");
#endif
    XS_STAMP(5);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[nt][mt][r];
          if (ACT == 1) x = x < 0.f ? 0.f : x;
          if (ACT == 2) x = gelu_erf(x);
          v[r] = x;
        }
        *reinterpret_cast<s16x4*>(my_stage + (mt * 16 + l15) * kXsPitch + ((c & 1) * CN + nt * 16 + grp * 4) * 2) =
            (SPLIT && c >= c1) ? T::pack4(v) : OT::pack4(v);
      }
#ifdef CODETR_XS_STAMPS
    asm volatile("" ::: "memory");
    XS_STAMP(6);
#pragma unroll
    for (int i = 0; i < 6; ++i) xs_acc[i] += xs_t[i + 1] - xs_t[i];
#endif
  }
  // the last pair (one chunk if nchunks is odd)
  {
    const int np = ((nchunks - 1) & ~1) * CN;
    if (HAS_RES && (nchunks & 1)) load_residual(np);   // (an even count requested it in the last chunk)
    flush_pair(np);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant fetches of the last two iterations
#ifdef CODETR_XS_STAMPS
  if (threadIdx.x == 0 && g_xs_stamps) {
    unsigned long long* o = g_xs_stamps + 12 * (size_t)blockIdx.x;
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = xs_acc[i];
    o[6] = xs_loop - xs_start;                       // prologue
    o[7] = __builtin_readcyclecounter() - xs_start;  // whole workgroup
    o[8] = xs_rt0;
    o[9] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// The short-K kernel serves f16 / bf16, K in {192, 256}, N % 8 == 0 (16-byte row chunks), 128 <= N <= 1536 (bias in
// LDS; narrower outputs measured equal or slower), at least one workgroup per CU (below that the tiled kernel's
// N-parallelism wins); column-block-major output only for blocks of a multiple of 64 columns (a stored chunk pair
// must not straddle blocks: the six decoder value projections as one N = 1536 GEMM, not the 32-wide MSDA heads).  K = 384 (Swin stage 1) fits the registers but measured 20-30 % slower
// than the tiled kernel (2 workgroups per CU, 48 MFMAs per barrier) and stays there.
bool xs_applicable(int64_t M, int64_t N, int64_t K, int hm_hd) {
  // (K = 64: the patch-embedding GEMM of the Swin stem, act 0 / no residual only)
  return (K == 192 || K == 256 || K == 64) && N % 8 == 0 && N >= 128 && N <= 1536 && M >= 128 * 256 && hm_hd % 8 == 0;
}

template <class T, int KS, int ACT, class OT = T>
int launch_xs_res(hipStream_t st, const void* X, const void* W, const void* bias, const void* R, void* Y,
                  const void* mask, int M, int N, int hm_rows, int hm_hd, const void* X2 = nullptr,
                  const void* LNG = nullptr, const void* LNB = nullptr, float ln_eps = 0.f) {
  const dim3 grid((unsigned)((M + 127) / 128)), block(256);
  auto x = static_cast<const unsigned short*>(X);
  auto x2 = static_cast<const unsigned short*>(X2);
  auto lg = static_cast<const unsigned short*>(LNG);
  auto lb = static_cast<const unsigned short*>(LNB);
  auto w = static_cast<const unsigned short*>(W);
  auto b = static_cast<const unsigned short*>(bias);
  auto r = static_cast<const unsigned short*>(R);
  auto y = static_cast<unsigned short*>(Y);
  auto mk = static_cast<const unsigned char*>(mask);
  if (R && std::is_same<T, OT>::value)
    hipLaunchKernelGGL((linear_xs_kernel<T, KS, ACT, true, T>), grid, block, 0, st, x, x2, lg, lb, ln_eps, w, b, r, y, mk, M, N, hm_rows, hm_hd);
  else if (R) return CODETR_E_UNSUPPORTED;   // (a residual is added in the operands' type)
  else hipLaunchKernelGGL((linear_xs_kernel<T, KS, ACT, false, OT>), grid, block, 0, st, x, x2, lg, lb, ln_eps, w, b, r, y, mk, M, N, hm_rows, hm_hd);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

template <class T, int KS>
int launch_xs_act(hipStream_t st, const void* X, const void* W, const void* bias, const void* R, void* Y,
                  const void* mask, int M, int N, int act, int hm_rows, int hm_hd) {
  switch (act) {
    case 0: return launch_xs_res<T, KS, 0>(st, X, W, bias, R, Y, mask, M, N, hm_rows, hm_hd);
    case 1: return launch_xs_res<T, KS, 1>(st, X, W, bias, R, Y, mask, M, N, hm_rows, hm_hd);
    default: return launch_xs_res<T, KS, 2>(st, X, W, bias, R, Y, mask, M, N, hm_rows, hm_hd);
  }
}

template <class T>
int launch_xs(hipStream_t st, const void* X, const void* W, const void* bias, const void* R, void* Y, const void* mask,
              int M, int N, int K, int act, int hm_rows, int hm_hd) {
  switch (K) {
    case 64: return launch_xs_res<T, 2, 0>(st, X, W, bias, nullptr, Y, mask, M, N, hm_rows, hm_hd);
    case 192: return launch_xs_act<T, 6>(st, X, W, bias, R, Y, mask, M, N, act, hm_rows, hm_hd);
    default: return launch_xs_act<T, 8>(st, X, W, bias, R, Y, mask, M, N, act, hm_rows, hm_hd);
  }
}

// which of the three kernels behind codetr_linear_* serves a problem (the ONE place this is decided; exported as
// codetr_linear_variant so hosts and tests can tell which kernel ran).  io16: Y (and R) 16-byte aligned.
enum { kKernel128 = 0, kKernel256 = 1, kKernelXS = 2 };
int pick_kernel(int64_t M, int64_t N, int64_t K, int act, bool has_res, int hm_hd, bool io16) {
  if (io16 && big_applicable(M, N, K, hm_hd)) return kKernel256;
  if (io16 && xs_applicable(M, N, K, hm_hd) && (K != 64 || (act == 0 && !has_res))) return kKernelXS;
  return kKernel128;
}

template <class T>
int launch(hipStream_t st, const void* X, const void* W, const void* bias, const void* R, void* Y, const void* mask,
           int64_t M, int64_t N, int64_t K, int act, int64_t hm_rows, int hm_hd) {
  if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return CODETR_E_BADARG;
  if (K % 64 != 0 || act < 0 || act > 2) return CODETR_E_UNSUPPORTED;
  if (M > 0x7fffffffLL || N > 0x7fffffffLL || K > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  if (((M + BM - 1) / BM) * ((N + BN - 1) / BN) > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) & 15) return CODETR_E_BADARG;
  if (hm_hd != 0 || hm_rows != 0) {
    if (hm_hd <= 0 || hm_rows <= 0 || hm_hd % 8 != 0 || N % hm_hd != 0 || N % 8 != 0 || M % hm_rows != 0 || R)
      return CODETR_E_UNSUPPORTED;
  }
  const bool io16 = (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && (!R || (reinterpret_cast<uintptr_t>(R) & 15) == 0);
  const int kind = pick_kernel(M, N, K, act, R != nullptr, hm_hd, io16);
  if (kind == kKernel256) {
    switch (act) {
      case 0: return launch_big<T, 0>(st, X, W, bias, R, Y, mask, (int)M, (int)N, (int)K);
      case 1: return launch_big<T, 1>(st, X, W, bias, R, Y, mask, (int)M, (int)N, (int)K);
      default: return launch_big<T, 2>(st, X, W, bias, R, Y, mask, (int)M, (int)N, (int)K);
    }
  }
  if (kind == kKernelXS)
    return launch_xs<T>(st, X, W, bias, R, Y, mask, (int)M, (int)N, (int)K, act, (int)hm_rows, hm_hd);
  switch (act) {
    case 0: return launch_act<T, 0>(st, X, W, bias, R, Y, mask, (int)M, (int)N, (int)K, (int)hm_rows, hm_hd);
    case 1: return launch_act<T, 1>(st, X, W, bias, R, Y, mask, (int)M, (int)N, (int)K, (int)hm_rows, hm_hd);
    default: return launch_act<T, 2>(st, X, W, bias, R, Y, mask, (int)M, (int)N, (int)K, (int)hm_rows, hm_hd);
  }
}

// x + x2 folded into the X-stationary kernel's operand load (act 0, no residual / mask): only where that kernel applies
template <class T>
int launch_xadd(hipStream_t st, const void* X, const void* X2, const void* W, const void* bias, void* Y, int64_t M,
                int64_t N, int64_t K) {
  if (!X || !X2 || !W || !Y || M <= 0 || N <= 0 || K <= 0) return CODETR_E_BADARG;
  if (M > 0x7fffffffLL || N > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  if (!xs_applicable(M, N, K, 0) || K == 64 || (reinterpret_cast<uintptr_t>(Y) & 15) ||
      ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(X2) | reinterpret_cast<uintptr_t>(W)) & 15))
    return CODETR_E_UNSUPPORTED;
  if (K == 192) return launch_xs_res<T, 6, 0>(st, X, W, bias, nullptr, Y, nullptr, (int)M, (int)N, 0, 0, X2);
  return launch_xs_res<T, 8, 0>(st, X, W, bias, nullptr, Y, nullptr, (int)M, (int)N, 0, 0, X2);
}

// LayerNorm of the input rows folded into the X-stationary kernel (no residual / mask): only where that kernel applies
template <class T>
int launch_ln(hipStream_t st, const void* X, const void* G, const void* Bt, float eps, const void* W, const void* bias,
              void* Y, int64_t M, int64_t N, int64_t K, int act) {
  if (!X || !G || !Bt || !W || !Y || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return CODETR_E_BADARG;
  if (M > 0x7fffffffLL || N > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  if (!xs_applicable(M, N, K, 0) || K == 64 || (reinterpret_cast<uintptr_t>(Y) & 15) ||
      ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Bt) |
        reinterpret_cast<uintptr_t>(W)) & 15))
    return CODETR_E_UNSUPPORTED;
#define CODETR_LN_CASE(KS, ACT) \
  return launch_xs_res<T, KS, ACT>(st, X, W, bias, nullptr, Y, nullptr, (int)M, (int)N, 0, 0, nullptr, G, Bt, eps)
  if (K == 192) {
    if (act == 0) CODETR_LN_CASE(6, 0);
    if (act == 1) CODETR_LN_CASE(6, 1);
    CODETR_LN_CASE(6, 2);
  }
  if (act == 0) CODETR_LN_CASE(8, 0);
  if (act == 1) CODETR_LN_CASE(8, 1);
  CODETR_LN_CASE(8, 2);
#undef CODETR_LN_CASE
}

// the encoder's two projections as one launch of the X-stationary kernel (SPLIT form): K = 256 only
template <class T, class OT>
int launch_split(hipStream_t st, const void* X, const void* X2, const void* W, const void* bias, const void* mask, void* Y,
                 void* Y2, int64_t M, int64_t N1, int64_t N2, int64_t K, int64_t hm_rows, int hm_hd) {
  if (!X || !X2 || !W || !Y || !Y2 || M <= 0 || N1 <= 0 || N2 <= 0 || K <= 0) return CODETR_E_BADARG;
  if (M > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(X2) | reinterpret_cast<uintptr_t>(W) |
       reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(Y2) | reinterpret_cast<uintptr_t>(bias)) & 15)
    return CODETR_E_BADARG;
  if (hm_hd != 0 || hm_rows != 0) {
    if (hm_hd <= 0 || hm_rows <= 0 || hm_hd % 8 != 0 || N1 % hm_hd != 0 || M % hm_rows != 0) return CODETR_E_UNSUPPORTED;
  }
  if (K != 256 || N1 % 64 != 0 || N2 % 8 != 0 || N1 + N2 > 1536 || M < 128 * 256) return CODETR_E_UNSUPPORTED;
  const dim3 grid((unsigned)((M + 127) / 128)), block(256);
  hipLaunchKernelGGL((linear_xs_kernel<T, 8, 0, false, OT, true>), grid, block, 0, st, static_cast<const unsigned short*>(X),
                     static_cast<const unsigned short*>(X2), nullptr, nullptr, 0.f, static_cast<const unsigned short*>(W),
                     static_cast<const unsigned short*>(bias), nullptr, static_cast<unsigned short*>(Y),
                     static_cast<const unsigned char*>(mask), (int)M, (int)(N1 + N2), (int)hm_rows, hm_hd,
                     static_cast<unsigned short*>(Y2), (int)N1);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

// ... with the positional operand generated in the kernel (POSGEN): same checks, plus the pyramid
template <class T, class OT>
int launch_split_pos(hipStream_t st, const void* X, const float* const* ycum, const float* const* xcum,
                     const int64_t* shapes, int L, const void* level_embed, float temperature, float scale, float eps,
                     float offset, int normalize, const void* W, const void* bias, const void* mask, void* Y, void* Y2,
                     int64_t M, int64_t S, int64_t N1, int64_t N2, int64_t K, int64_t hm_rows, int hm_hd) {
  if (!X || !ycum || !xcum || !shapes || !W || !Y || !Y2 || M <= 0 || S <= 0 || N1 <= 0 || N2 <= 0 || K <= 0 || L <= 0 ||
      !(temperature > 0.f))
    return CODETR_E_BADARG;
  if (L > 5) return CODETR_E_UNSUPPORTED;
  if (M > 0x7fffffffLL || M % S != 0) return M % S != 0 ? CODETR_E_BADARG : CODETR_E_TOO_LARGE;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Y) |
       reinterpret_cast<uintptr_t>(Y2) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(level_embed)) & 15)
    return CODETR_E_BADARG;
  if (hm_hd != 0 || hm_rows != 0) {
    if (hm_hd <= 0 || hm_rows <= 0 || hm_hd % 8 != 0 || N1 % hm_hd != 0 || M % hm_rows != 0) return CODETR_E_UNSUPPORTED;
  }
  if (K != 256 || N1 % 64 != 0 || N2 % 8 != 0 || N1 + N2 > 1536 || M < 128 * 256) return CODETR_E_UNSUPPORTED;
  XsPosGen pg;
  int64_t sum = 0;
  for (int l = 0; l < 5; ++l) {
    pg.ycum[l] = l < L ? ycum[l] : nullptr;
    pg.xcum[l] = l < L ? xcum[l] : nullptr;
    const int64_t h = l < L ? shapes[2 * l] : 1, w = l < L ? shapes[2 * l + 1] : 1;
    if (l < L && (!ycum[l] || !xcum[l] || h <= 0 || w <= 0 || h > 32767 || w > 32767)) return CODETR_E_BADARG;
    pg.H[l] = (int)h;
    pg.W[l] = (int)w;
    pg.start[l] = l < L ? (int)sum : 0x7fffffff;
    if (l < L) sum += h * w;
  }
  if (sum != S) return CODETR_E_BADARG;
  pg.level_embed = static_cast<const unsigned short*>(level_embed);
  pg.L = L;
  pg.S = (int)S;
  pg.log2_temperature = log2f(temperature);
  pg.scale = scale;
  pg.eps = eps;
  pg.offset = offset;
  pg.normalize = normalize;
  const dim3 grid((unsigned)((M + 127) / 128)), block(256);
  hipLaunchKernelGGL((linear_xs_kernel<T, 8, 0, false, OT, true, true>), grid, block, 0, st, static_cast<const unsigned short*>(X),
                     nullptr, nullptr, nullptr, 0.f, static_cast<const unsigned short*>(W),
                     static_cast<const unsigned short*>(bias), nullptr, static_cast<unsigned short*>(Y),
                     static_cast<const unsigned char*>(mask), (int)M, (int)(N1 + N2), (int)hm_rows, hm_hd,
                     static_cast<unsigned short*>(Y2), (int)N1, pg);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int codetr_encoder_projections_posgen_f16(void* stream, const void* x_dev, const float* ycum0, const float* ycum1,
                                          const float* ycum2, const float* ycum3, const float* ycum4, const float* xcum0,
                                          const float* xcum1, const float* xcum2, const float* xcum3, const float* xcum4,
                                          const int64_t* level_shapes_host, int num_levels, const void* level_embed_dev,
                                          float temperature, float scale, float eps, float offset, int normalize,
                                          const void* w_dev, const void* bias_dev, const void* row_mask_dev, void* value_dev,
                                          void* packed_dev, int64_t M, int64_t S, int64_t N_value, int64_t N_packed, int64_t K,
                                          int64_t hm_rows, int hm_head_dim) {
  const float* yc[5] = {ycum0, ycum1, ycum2, ycum3, ycum4};
  const float* xc[5] = {xcum0, xcum1, xcum2, xcum3, xcum4};
  return launch_split_pos<HalfT, HalfT>(static_cast<hipStream_t>(stream), x_dev, yc, xc, level_shapes_host, num_levels,
                                        level_embed_dev, temperature, scale, eps, offset, normalize, w_dev, bias_dev, row_mask_dev,
                                        value_dev, packed_dev, M, S, N_value, N_packed, K, hm_rows, hm_head_dim);
}

int codetr_encoder_projections_posgen_bf16(void* stream, const void* x_dev, const float* ycum0, const float* ycum1,
                                          const float* ycum2, const float* ycum3, const float* ycum4, const float* xcum0,
                                          const float* xcum1, const float* xcum2, const float* xcum3, const float* xcum4,
                                          const int64_t* level_shapes_host, int num_levels, const void* level_embed_dev,
                                          float temperature, float scale, float eps, float offset, int normalize,
                                          const void* w_dev, const void* bias_dev, const void* row_mask_dev, void* value_f16_dev,
                                          void* packed_dev, int64_t M, int64_t S, int64_t N_value, int64_t N_packed, int64_t K,
                                          int64_t hm_rows, int hm_head_dim) {
  const float* yc[5] = {ycum0, ycum1, ycum2, ycum3, ycum4};
  const float* xc[5] = {xcum0, xcum1, xcum2, xcum3, xcum4};
  return launch_split_pos<BFloatT, HalfT>(static_cast<hipStream_t>(stream), x_dev, yc, xc, level_shapes_host, num_levels,
                                        level_embed_dev, temperature, scale, eps, offset, normalize, w_dev, bias_dev, row_mask_dev,
                                        value_f16_dev, packed_dev, M, S, N_value, N_packed, K, hm_rows, hm_head_dim);
}

int codetr_encoder_projections_f16(void* stream, const void* x_dev, const void* pos_dev, const void* w_dev,
                                   const void* bias_dev, const void* row_mask_dev, void* value_dev, void* packed_dev,
                                   int64_t M, int64_t N_value, int64_t N_packed, int64_t K, int64_t hm_rows, int hm_head_dim) {
  return launch_split<HalfT, HalfT>(static_cast<hipStream_t>(stream), x_dev, pos_dev, w_dev, bias_dev, row_mask_dev,
                                    value_dev, packed_dev, M, N_value, N_packed, K, hm_rows, hm_head_dim);
}

int codetr_encoder_projections_bf16(void* stream, const void* x_dev, const void* pos_dev, const void* w_dev,
                                    const void* bias_dev, const void* row_mask_dev, void* value_f16_dev, void* packed_dev,
                                    int64_t M, int64_t N_value, int64_t N_packed, int64_t K, int64_t hm_rows,
                                    int hm_head_dim) {
  return launch_split<BFloatT, HalfT>(static_cast<hipStream_t>(stream), x_dev, pos_dev, w_dev, bias_dev, row_mask_dev,
                                      value_f16_dev, packed_dev, M, N_value, N_packed, K, hm_rows, hm_head_dim);
}

int codetr_linear_ln_f16(void* stream, const void* x_dev, const void* ln_gamma_dev, const void* ln_beta_dev, float ln_eps,
                         const void* w_dev, const void* bias_dev, void* y_dev, int64_t M, int64_t N, int64_t K, int act) {
  return launch_ln<HalfT>(static_cast<hipStream_t>(stream), x_dev, ln_gamma_dev, ln_beta_dev, ln_eps, w_dev, bias_dev,
                          y_dev, M, N, K, act);
}

int codetr_linear_ln_bf16(void* stream, const void* x_dev, const void* ln_gamma_dev, const void* ln_beta_dev, float ln_eps,
                          const void* w_dev, const void* bias_dev, void* y_dev, int64_t M, int64_t N, int64_t K, int act) {
  return launch_ln<BFloatT>(static_cast<hipStream_t>(stream), x_dev, ln_gamma_dev, ln_beta_dev, ln_eps, w_dev, bias_dev,
                            y_dev, M, N, K, act);
}

int codetr_linear_xadd_f16(void* stream, const void* x_dev, const void* x_add_dev, const void* w_dev,
                           const void* bias_dev, void* y_dev, int64_t M, int64_t N, int64_t K) {
  return launch_xadd<HalfT>(static_cast<hipStream_t>(stream), x_dev, x_add_dev, w_dev, bias_dev, y_dev, M, N, K);
}

int codetr_linear_xadd_bf16(void* stream, const void* x_dev, const void* x_add_dev, const void* w_dev,
                            const void* bias_dev, void* y_dev, int64_t M, int64_t N, int64_t K) {
  return launch_xadd<BFloatT>(static_cast<hipStream_t>(stream), x_dev, x_add_dev, w_dev, bias_dev, y_dev, M, N, K);
}

int codetr_linear_f16(void* stream, const void* x_dev, const void* w_dev, const void* bias_dev, const void* residual_dev,
                      const void* row_mask_dev, void* y_dev, int64_t M, int64_t N, int64_t K, int act, int64_t hm_rows,
                      int hm_head_dim) {
  return launch<HalfT>(static_cast<hipStream_t>(stream), x_dev, w_dev, bias_dev, residual_dev, y_dev, row_mask_dev, M,
                       N, K, act, hm_rows, hm_head_dim);
}

int codetr_linear_bf16(void* stream, const void* x_dev, const void* w_dev, const void* bias_dev,
                       const void* residual_dev, const void* row_mask_dev, void* y_dev, int64_t M, int64_t N,
                       int64_t K, int act, int64_t hm_rows, int hm_head_dim) {
  return launch<BFloatT>(static_cast<hipStream_t>(stream), x_dev, w_dev, bias_dev, residual_dev, y_dev, row_mask_dev,
                         M, N, K, act, hm_rows, hm_head_dim);
}

int codetr_linear_bf16_f16out(void* stream, const void* x_dev, const void* w_dev, const void* bias_dev,
                              const void* row_mask_dev, void* y_dev, int64_t M, int64_t N, int64_t K, int64_t hm_rows,
                              int hm_head_dim) {
  if (!x_dev || !w_dev || !y_dev || M <= 0 || N <= 0 || K <= 0) return CODETR_E_BADARG;
  if (M > 0x7fffffffLL || N > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  if ((reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(w_dev) | reinterpret_cast<uintptr_t>(y_dev)) & 15) return CODETR_E_BADARG;
  if (hm_head_dim != 0 || hm_rows != 0) {
    if (hm_head_dim <= 0 || hm_rows <= 0 || hm_head_dim % 8 != 0 || N % hm_head_dim != 0 || M % hm_rows != 0) return CODETR_E_UNSUPPORTED;
  }
  if (!xs_applicable(M, N, K, hm_head_dim) || K == 64) return CODETR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (K == 192) return launch_xs_res<BFloatT, 6, 0, HalfT>(st, x_dev, w_dev, bias_dev, nullptr, y_dev, row_mask_dev, (int)M, (int)N, (int)hm_rows, hm_head_dim);
  return launch_xs_res<BFloatT, 8, 0, HalfT>(st, x_dev, w_dev, bias_dev, nullptr, y_dev, row_mask_dev, (int)M, (int)N, (int)hm_rows, hm_head_dim);
}

const char* codetr_linear_variant(int64_t M, int64_t N, int64_t K, int act, int has_residual, int hm_head_dim) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 64 != 0) return "unsupported";
  switch (pick_kernel(M, N, K, act, has_residual != 0, hm_head_dim, true)) {
    case kKernel256: return "tile256";
    case kKernelXS: return "xs";
    default: return "tile128";
  }
}

int codetr_linear_splitk_plan(int64_t M, int64_t N, int64_t K, int64_t* workspace_bytes) {
  const int splits = splitk_plan(M, N, K);
  if (workspace_bytes) *workspace_bytes = splits > 1 ? (int64_t)splits * M * N * 4 : 0;
  return splits;
}

int codetr_linear_splitk_f16(void* stream, const void* x_dev, const void* w_dev, const void* bias_dev,
                             const void* residual_dev, const void* row_mask_dev, void* y_dev, int64_t M, int64_t N,
                             int64_t K, int act, int splits, void* workspace_dev, int64_t workspace_bytes) {
  return launch_splitk<HalfT>(static_cast<hipStream_t>(stream), x_dev, w_dev, bias_dev, residual_dev, y_dev,
                              row_mask_dev, M, N, K, act, splits, workspace_dev, workspace_bytes);
}

int codetr_linear_splitk_bf16(void* stream, const void* x_dev, const void* w_dev, const void* bias_dev,
                              const void* residual_dev, const void* row_mask_dev, void* y_dev, int64_t M, int64_t N,
                              int64_t K, int act, int splits, void* workspace_dev, int64_t workspace_bytes) {
  return launch_splitk<BFloatT>(static_cast<hipStream_t>(stream), x_dev, w_dev, bias_dev, residual_dev, y_dev,
                                row_mask_dev, M, N, K, act, splits, workspace_dev, workspace_bytes);
}

}  // extern "C"
