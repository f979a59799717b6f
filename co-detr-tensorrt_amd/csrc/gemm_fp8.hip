// FP8 (OCP e4m3) linear layer for MI355X (gfx950) -- BASELINE config 5, "fp8 weights + activations":
//   Y[M,N] = act( (X8[M,K] . W8[N,K]^T) * x_scale * w_scale[n] + bias[n] ) (+ R[M,N])
// X8 / W8 are e4m3 bytes (K contiguous), x_scale one fp32 per tensor (static, from a calibration forward), w_scale one
// fp32 per output channel, accumulation in fp32 on the matrix cores with the K = 128 block-scaled instruction
// v_mfma_scale_f32_16x16x128_f8f6f4 at unit block scales (the real scales are applied once, in the epilogue): on
// gfx950 that instruction takes twice the cycles of v_mfma_f32_16x16x32_f16 at 4x the K, i.e. twice the fp16 rate,
// while the plain fp8 MFMA (16x16x32_fp8_fp8) only runs at the fp16 rate (MI355X_MICROARCH.md, matrix cores).
// Output: fp16 (default) or e4m3 with its own static scale (an fc1 whose only reader is the fp8 fc2).
//
// There is no reference counterpart (the reference's dtypes stop at half: codetr/csrc/ms_deform_attn.cu:946,
// export.py:39-44); the layers it serves are the reference's nn.Linears of codetr/swin.py:91-115, 345-355.
//
// Structure = linear_256_kernel of gemm_f16.hip with bytes in place of halves: a 128-byte k-tile row is 128 fp8
// values instead of 64 halves, so the LDS image (256 rows x 8 16-byte chunks per operand and stage, XOR swizzle on the
// DMA source address), the 2-stage LDS-DMA ring and the LDS-staged epilogue are the same; one k-tile is ONE MFMA deep
// (K = 128) per 16x16 output tile: lane l holds bytes 32 (l >> 4) .. + 31 of row l & 15 of each operand (two
// ds_read_b128; any consistent k permutation of A and B gives the same dot product).  512 threads = 8 waves (2 x 4),
// each 128 (m) x 64 (n) = 8 x 4 MFMA tiles; fragment reads of m-tile j + 1 are issued under the MFMAs of m-tile j.
//
// Requirements: K % 128 == 0, N % 8 == 0, 16-byte aligned bases; M arbitrary.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "codetr_hip.h"
#include "mx_scale.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int kStagePitch = 64 * 2 + 16;  // epilogue staging: bytes per staged row of a wave's 64-column slice
constexpr float kFp8Max = 448.0f;         // largest finite e4m3 (OCP e4m3fn)

__device__ __forceinline__ unsigned xcd_tile(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

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

// XOR swizzle of the 16-byte chunk index of a 128-byte LDS row.  A lane's fragment is two ds_read_b128 (chunks 2g, 2g+1,
// g = lane >> 4), and ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): a group
// holds 8 rows at chunk c and the other 8 at chunk c ^ 2.  With q = (row >> 1) & 7, f(q) = q ^ ((q & 2) << 1) maps the
// row pairs {2..5} (rows 4-11) onto {4..7}, a set closed under ^ 2, so the 16 lanes of a group hit 16 different 16-byte
// bank groups (plain f(q) = q is 2-way conflicted: tests/test_lds_bank_model.py).
__device__ __forceinline__ int sw(int row) {
  const int q = (row >> 1) & 7;
  return q ^ ((q & 2) << 1);
}

__device__ __forceinline__ i32x4 read_piece(const unsigned char* tile, int row, int chunk) {
  return *reinterpret_cast<const i32x4*>(tile + row * 128 + ((chunk ^ sw(row)) * 16));
}
// the lane's 32 bytes of one operand row: chunks 2g and 2g + 1 (g = lane >> 4)
__device__ __forceinline__ i32x8 read_frag(const unsigned char* tile, int row, int g) {
  const i32x4 lo = read_piece(tile, row, 2 * g), hi = read_piece(tile, row, 2 * g + 1);
  return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// MX form: the hardware's k order.  Lane group g = lane >> 4 holds k [16 g, 16 g + 16) in registers 0-3 and
// k [64 + 16 g, 64 + 16 g + 16) in registers 4-7 -- 16-byte chunks g and g + 4 -- and the scale byte of lane row + 16 b covers
// MX block b = k [32 b, 32 b + 32) = chunks 2 b, 2 b + 1 (probed: tools/micro/mx_scale_probe2.hip).  With unit scales any
// consistent k permutation works (read_frag above); with block scales the operands must sit where the instruction
// expects them.  Bank check as for read_frag: a ds_read_b128 service group then reads chunks {c, c ^ 1} of 16 different
// rows whose swizzled images are distinct (tests/test_lds_bank_model.py).
__device__ __forceinline__ i32x8 read_frag_mx(const unsigned char* tile, int row, int g) {
  const i32x4 lo = read_piece(tile, row, g), hi = read_piece(tile, row, g + 4);
  return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ float h2f(unsigned short bits) {
  _Float16 h;
  __builtin_memcpy(&h, &bits, 2);
  return (float)h;
}
__device__ __forceinline__ unsigned short f2h(float v) {
  _Float16 h = (_Float16)v;
  unsigned short bits;
  __builtin_memcpy(&bits, &h, 2);
  return bits;
}
// two floats -> two e4m3 bytes (saturating: clamped to +-448 first; NaN stays NaN)
__device__ __forceinline__ unsigned pk_fp8(float a, float b) {
  a = __builtin_amdgcn_fmed3f(a, -kFp8Max, kFp8Max);
  b = __builtin_amdgcn_fmed3f(b, -kFp8Max, kFp8Max);
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}

// one lane's 8 scale bytes of a k-tile (mx_scale.h layout): an ORDINARY load, waited for by the compiler.  hipcc drains
// the whole vector-memory queue (vmcnt(0)) at the first use of such a load while LDS-DMA pieces are in flight, i.e. at the
// top of the next k-tile -- so in the MX form X(t+2)'s pieces have to land one tile early and the three-slot X ring
// degenerates to a two-slot one in time (measured cost: profiles/r03_fp8_mx_gemm.txt).  The alternative -- a raw
// inline-assembly load hidden from the compiler's bookkeeping, covered by the loop's own counted wait -- was built and
// REMOVED: the compiler is free to copy / recycle the asm's destination register before the data lands (it placed the
// loop-carried copy ahead of the wait and reused the register for address arithmetic: wrong scales now and then, and a
// memory fault when the late write hit a recycled address register; tools/micro notes in DESIGN.md section 4).
__device__ __forceinline__ unsigned long long ld_scales(const unsigned char* tile_base, unsigned lane_off) {
  return *reinterpret_cast<const unsigned long long*>(tile_base + lane_off);
}
template <int OPSEL>
__device__ __forceinline__ f32x4 mfma_mx(const i32x8& a, const i32x8& b, const f32x4& c, int scale_b) {
  // A = the weight rows at unit block scales (their real scale is per output channel, applied in the epilogue),
  // B = the activation rows with their block scales
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, OPSEL, scale_b);
}

// ACT: 0 none, 1 relu, 2 gelu(erf).  OUT8: Y is e4m3 (y / out_scale), else fp16.
// MX: X carries block scales (sx, mx_scale.h) instead of one static scale; with OUT8 the output gets block scales of its
// own along N (sy), laid out for a consumer GEMM whose K is this N.
template <int ACT, bool HAS_BIAS, bool HAS_RES, bool OUT8, bool MX = false>
__global__ __launch_bounds__(512) void linear_256_fp8_kernel(const unsigned char* __restrict__ X,
                                                             const unsigned char* __restrict__ W,
                                                             const float* __restrict__ w_scale, const float x_scale,
                                                             const unsigned short* __restrict__ bias,
                                                             const unsigned short* __restrict__ R, void* __restrict__ Yv,
                                                             const float out_inv_scale, int M, int N, int K,
                                                             int tiles_n, const unsigned char* __restrict__ sx,
                                                             unsigned char* __restrict__ sy) {
  constexpr int NT = 512;
  constexpr int kTileBytes = 256 * 128;        // 32 KiB: one operand tile (256 rows x 128 bytes of K)
  // LDS = [X0 X1 X2][W0 W1], 160 KiB (all of the CU's): X is prefetched two k-tiles ahead, W one (the XDEEP form of
  // linear_256_kernel -- with k-tiles half as long in time as fp16's, hiding the ~2 us operand fetch matters more here)
  __shared__ __attribute__((aligned(16))) unsigned char lds[5 * kTileBytes];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  const int frow = lane & 15, fg = lane >> 4, ncol = 4 * (lane >> 4);

  f32x4 acc[4][8];  // [n-tile][m-tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-thread source pointers of the 4 + 4 LDS-DMA pieces of a stage (piece q covers rows q*64 + tid/8)
  const unsigned char* gw[4];
  const unsigned char* gx[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = q * 64 + (tid >> 3), pos = tid & 7;
    const int chunk = pos ^ sw(r);
    int gn = n0 + r, gm = m0 + r;
    gn = gn < N ? gn : N - 1;
    gm = gm < M ? gm : M - 1;
    gw[q] = W + (size_t)gn * K + chunk * 16;
    gx[q] = X + (size_t)gm * K + chunk * 16;
  }
  auto dma = [&](const unsigned char* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  };
  const int nk = K / 128;
  const int64_t MB = mx_blocks128(M);
  const unsigned char* sx_blk = MX ? sx + (size_t)(tm * 2 + wm < MB ? tm * 2 + wm : MB - 1) * 512 : nullptr;
  unsigned long long sxc = 0, sxn = 0;   // scale bytes of the current / next k-tile
  if (MX) sxn = ld_scales(sx_blk, (unsigned)lane * 8);
  // issue order W(0), X(0), X(1): the youngest four pieces may stay in flight at the first wait
#pragma unroll
  for (int q = 0; q < 4; ++q) dma(gw[q], lds + 3 * kTileBytes + (q * NT + wave * 64) * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) dma(gx[q], lds + (q * NT + wave * 64) * 16);
  {
    const size_t k1 = (size_t)(nk > 1 ? 1 : 0) * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) dma(gx[q] + k1, lds + kTileBytes + (q * NT + wave * 64) * 16);
  }
  int xs = 0;  // ring slot of X(t)
  // one k-tile.  MX: `s_use` = the tile's scale bytes, `s_load` receives the next tile's
  auto k_tile = [&](const int t, unsigned long long& s_use, unsigned long long& s_load) {
    // W(t), X(t) (and the scales of tile t) landed; the four pieces of X(t+1) may be in flight
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile t is in LDS for everyone; everyone is done reading tile t-1
    const unsigned char* bufW = lds + (3 + (t & 1)) * kTileBytes;
    const unsigned char* bufX = lds + xs * kTileBytes;
    // W(t+1) -> the W slot tile t-1 used, X(t+2) -> the X slot tile t-1 used, one piece per m-tile below, W first (past
    // the last tile: re-fetch it, no branch; drained before the epilogue)
    const size_t koffw = (size_t)(t + 1 < nk ? t + 1 : nk - 1) * 128;
    const size_t koffx = (size_t)(t + 2 < nk ? t + 2 : nk - 1) * 128;
    unsigned char* nbufW = lds + (3 + ((t + 1) & 1)) * kTileBytes;
    unsigned char* nbufX = lds + (xs >= 1 ? xs - 1 : 2) * kTileBytes;
    xs = xs == 2 ? 0 : xs + 1;
    if (MX) s_load = ld_scales(sx_blk + (size_t)(t + 1 < nk ? t + 1 : nk - 1) * (size_t)MB * 512, (unsigned)lane * 8);
    i32x8 a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      a[i] = MX ? read_frag_mx(bufW, wn * 64 + i * 16 + frow, fg) : read_frag(bufW, wn * 64 + i * 16 + frow, fg);
    b[0] = MX ? read_frag_mx(bufX, wm * 128 + frow, fg) : read_frag(bufX, wm * 128 + frow, fg);
    __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < 7)
        b[(j + 1) & 1] = MX ? read_frag_mx(bufX, wm * 128 + (j + 1) * 16 + frow, fg)
                            : read_frag(bufX, wm * 128 + (j + 1) * 16 + frow, fg);
      if (j < 4) dma(gw[j] + koffw, nbufW + (j * NT + wave * 64) * 16);
      else dma(gx[j - 4] + koffx, nbufX + ((j - 4) * NT + wave * 64) * 16);
      const int sb = (int)(j < 4 ? (unsigned)s_use : (unsigned)(s_use >> 32));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!MX) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j & 1], acc[i][j], 0, 0, 0, 0, 0, 0);
        else if ((j & 3) == 0) acc[i][j] = mfma_mx<0>(a[i], b[j & 1], acc[i][j], sb);
        else if ((j & 3) == 1) acc[i][j] = mfma_mx<1>(a[i], b[j & 1], acc[i][j], sb);
        else if ((j & 3) == 2) acc[i][j] = mfma_mx<2>(a[i], b[j & 1], acc[i][j], sb);
        else acc[i][j] = mfma_mx<3>(a[i], b[j & 1], acc[i][j], sb);
      }
      if (j < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  };
  if (MX) {
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      k_tile(t, sxn, sxc);
      k_tile(t + 1, sxc, sxn);
    }
    if (t < nk) k_tile(t, sxn, sxc);
  } else {
    for (int t = 0; t < nk; ++t) k_tile(t, sxc, sxn);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // all fragment reads done, no DMA in flight: LDS is free for the epilogue

  // per-lane column constants: 4 consecutive n per n-tile
  float sc[4][4], bs[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + wn * 64 + i * 16 + ncol;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = n + r < N;
      sc[i][r] = ok ? w_scale[n + r] * (MX ? 1.0f : x_scale) : 0.f;
      bs[i][r] = (HAS_BIAS && ok) ? h2f(bias[n + r]) : 0.f;
    }
  }

  // epilogue in two 64-row halves per wave through its private staging region (64 rows x 144 B), fp16 image
  constexpr int kPitch = kStagePitch;
  unsigned char* stage = lds + wave * (64 * kPitch);
  const int srow = lane >> 3, schunk = lane & 7;
  const int n = n0 + wn * 64 + schunk * 8;
  unsigned char* ytile = reinterpret_cast<unsigned char*>(Yv) + ((size_t)m0 * N + n0) * (OUT8 ? 1 : 2);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f16x4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = fmaf(acc[i][h * 4 + j][r], sc[i][r], bs[i][r]);
          if (ACT == 1) x = x < 0.f ? 0.f : x;
          if (ACT == 2) x = gelu_erf(x);
          hv[r] = (_Float16)x;
        }
        s16x4 o;
        __builtin_memcpy(&o, &hv, 8);
        *reinterpret_cast<s16x4*>(stage + (j * 16 + frow) * kPitch + (i * 16 + ncol) * 2) = o;
      }
    __builtin_amdgcn_wave_barrier();
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
        if (HAS_RES) {
          const s16x8 rr = rres[it];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (short)f2h(h2f((unsigned short)v[e]) + h2f((unsigned short)rr[e]));
        }
        const unsigned eoff = (unsigned)(wm * 128 + h * 64 + ml) * (unsigned)N + (unsigned)(wn * 64 + schunk * 8);
        if (OUT8 && MX) {
          // block scales along N: the 32 columns of a block are the 8 values of 4 neighbouring lanes
          float y[8], amax = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            y[e] = h2f((unsigned short)v[e]);
            amax = fmaxf(amax, fabsf(y[e]));
          }
          amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
          amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
          const unsigned sbyte = mx_e8m0(amax);
          const float inv = mx_inv_scale(sbyte);
          const unsigned lo = pk_fp8(y[0] * inv, y[1] * inv) | (pk_fp8(y[2] * inv, y[3] * inv) << 16);
          const unsigned hi = pk_fp8(y[4] * inv, y[5] * inv) | (pk_fp8(y[6] * inv, y[7] * inv) << 16);
          *reinterpret_cast<uint2*>(ytile + eoff) = uint2{lo, hi};
          if ((schunk & 3) == 0) sy[mx_index(m, n >> 5, MB)] = (unsigned char)sbyte;
        } else if (OUT8) {
          unsigned lo = 0, hi = 0;
          lo = pk_fp8(h2f((unsigned short)v[0]) * out_inv_scale, h2f((unsigned short)v[1]) * out_inv_scale) |
               (pk_fp8(h2f((unsigned short)v[2]) * out_inv_scale, h2f((unsigned short)v[3]) * out_inv_scale) << 16);
          hi = pk_fp8(h2f((unsigned short)v[4]) * out_inv_scale, h2f((unsigned short)v[5]) * out_inv_scale) |
               (pk_fp8(h2f((unsigned short)v[6]) * out_inv_scale, h2f((unsigned short)v[7]) * out_inv_scale) << 16);
          *reinterpret_cast<uint2*>(ytile + eoff) = uint2{lo, hi};
        } else {
          *reinterpret_cast<s16x8*>(ytile + eoff * 2u) = v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the region is rewritten by the second half
  }
}

template <int ACT, bool OUT8, bool MX = false>
int launch_bias_res(hipStream_t st, const void* X, const void* W, const float* ws, float xs, const void* bias,
                    const void* R, void* Y, float out_inv, int M, int N, int K, const unsigned char* sx = nullptr,
                    unsigned char* sy = nullptr) {
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
  const dim3 grid((unsigned)(tiles_m * tiles_n)), block(512);
  auto x = static_cast<const unsigned char*>(X);
  auto w = static_cast<const unsigned char*>(W);
  auto b = static_cast<const unsigned short*>(bias);
  auto r = static_cast<const unsigned short*>(R);
#define CODETR_FP8_LAUNCH(HB, HR) \
  hipLaunchKernelGGL((linear_256_fp8_kernel<ACT, HB, HR, OUT8, MX>), grid, block, 0, st, x, w, ws, xs, b, r, Y, out_inv, M, N, K, tiles_n, sx, sy)
  if (bias && R) CODETR_FP8_LAUNCH(true, true);
  else if (bias) CODETR_FP8_LAUNCH(true, false);
  else if (R) CODETR_FP8_LAUNCH(false, true);
  else CODETR_FP8_LAUNCH(false, false);
#undef CODETR_FP8_LAUNCH
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

// ---- producers of e4m3 activations -------------------------------------------------------------------------

// y8 = sat(x * inv_scale): n16 = number of 16-byte (8-half) groups
__global__ __launch_bounds__(256) void cast_fp8_kernel(const s16x8* __restrict__ x, uint2* __restrict__ y, float inv_scale,
                                                       int64_t n16) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  const s16x8 v = x[i];
  const unsigned lo = pk_fp8(h2f((unsigned short)v[0]) * inv_scale, h2f((unsigned short)v[1]) * inv_scale) |
                      (pk_fp8(h2f((unsigned short)v[2]) * inv_scale, h2f((unsigned short)v[3]) * inv_scale) << 16);
  const unsigned hi = pk_fp8(h2f((unsigned short)v[4]) * inv_scale, h2f((unsigned short)v[5]) * inv_scale) |
                      (pk_fp8(h2f((unsigned short)v[6]) * inv_scale, h2f((unsigned short)v[7]) * inv_scale) << 16);
  y[i] = uint2{lo, hi};
}

// LayerNorm over the last dimension (fp16 in, fp32 statistics, two passes over registers) -> e4m3: one wave per row,
// C % 8 == 0, C <= 4096 (8 chunks of 8 halves per lane)
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const unsigned short* __restrict__ x,
                                                            const unsigned short* __restrict__ gamma,
                                                            const unsigned short* __restrict__ beta,
                                                            unsigned char* __restrict__ y, int64_t rows, int C, float eps,
                                                            float inv_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = C >> 3;  // 16-byte chunks per row
  float v[8][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nch) {
      const s16x8 r = *reinterpret_cast<const s16x8*>(x + row * C + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[c][e] = h2f((unsigned short)r[e]);
        sum += v[c][e];
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (c * 64 + lane < nch)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[c][e] - mean;
        var += d * d;
      }
  for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
  const float rstd = rsqrtf(var / (float)C + eps);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nch) {
      const s16x8 g = *reinterpret_cast<const s16x8*>(gamma + ch * 8);
      const s16x8 b = *reinterpret_cast<const s16x8*>(beta + ch * 8);
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // the fp16 LayerNorm output the fp16 model would have produced, then quantised
        const float ln = h2f(f2h((v[c][e] - mean) * rstd * h2f((unsigned short)g[e]) + h2f((unsigned short)b[e])));
        o[e] = ln * inv_scale;
      }
      const unsigned lo = pk_fp8(o[0], o[1]) | (pk_fp8(o[2], o[3]) << 16);
      const unsigned hi = pk_fp8(o[4], o[5]) | (pk_fp8(o[6], o[7]) << 16);
      *reinterpret_cast<uint2*>(y + row * C + ch * 8) = uint2{lo, hi};
    }
  }
}

// ---- MX producers: e4m3 + one e8m0 byte per (row, 32 channels), mx_scale.h layout for a consumer GEMM over these rows ----

// plain cast of a 16-bit [rows, C] tensor (C % 32 == 0): a lane owns 8 consecutive channels, 4 neighbouring lanes a block
__global__ __launch_bounds__(256) void cast_fp8mx_kernel(const s16x8* __restrict__ x, uint2* __restrict__ y,
                                                         unsigned char* __restrict__ sy, int64_t n16, int chunks_per_row,
                                                         int64_t MB) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool ok = i < n16;
  const s16x8 v = ok ? x[i] : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
  float f[8], amax = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    f[e] = h2f((unsigned short)v[e]);
    amax = fmaxf(amax, fabsf(f[e]));
  }
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  if (!ok) return;
  const unsigned sb = mx_e8m0(amax);
  const float inv = mx_inv_scale(sb);
  y[i] = uint2{pk_fp8(f[0] * inv, f[1] * inv) | (pk_fp8(f[2] * inv, f[3] * inv) << 16),
               pk_fp8(f[4] * inv, f[5] * inv) | (pk_fp8(f[6] * inv, f[7] * inv) << 16)};
  if ((i & 3) == 0) {
    const int64_t row = i / chunks_per_row;
    const int ch = (int)(i - row * chunks_per_row);
    sy[mx_index(row, ch >> 2, MB)] = (unsigned char)sb;
  }
}

// layernorm_fp8_kernel with block scales: the fp16 LayerNorm output the fp16 model would have produced, then MX-quantised
__global__ __launch_bounds__(256) void layernorm_fp8mx_kernel(const unsigned short* __restrict__ x,
                                                              const unsigned short* __restrict__ gamma,
                                                              const unsigned short* __restrict__ beta,
                                                              unsigned char* __restrict__ y, unsigned char* __restrict__ sy,
                                                              int64_t rows, int C, float eps, int64_t MB) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;   // (whole wave)
  const int nch = C >> 3;  // 16-byte chunks per row (a multiple of 4: C % 32 == 0)
  float v[8][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nch) {
      const s16x8 r = *reinterpret_cast<const s16x8*>(x + row * C + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[c][e] = h2f((unsigned short)r[e]);
        sum += v[c][e];
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (c * 64 + lane < nch)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[c][e] - mean;
        var += d * d;
      }
  for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
  const float rstd = rsqrtf(var / (float)C + eps);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ch = c * 64 + lane;
    const bool ok = ch < nch;   // uniform over a 4-lane block group (nch % 4 == 0)
    float o[8], amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    if (ok) {
      const s16x8 g = *reinterpret_cast<const s16x8*>(gamma + ch * 8);
      const s16x8 b = *reinterpret_cast<const s16x8*>(beta + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = h2f(f2h((v[c][e] - mean) * rstd * h2f((unsigned short)g[e]) + h2f((unsigned short)b[e])));
        amax = fmaxf(amax, fabsf(o[e]));
      }
    }
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    if (ok) {
      const unsigned sb = mx_e8m0(amax);
      const float inv = mx_inv_scale(sb);
      const unsigned lo = pk_fp8(o[0] * inv, o[1] * inv) | (pk_fp8(o[2] * inv, o[3] * inv) << 16);
      const unsigned hi = pk_fp8(o[4] * inv, o[5] * inv) | (pk_fp8(o[6] * inv, o[7] * inv) << 16);
      *reinterpret_cast<uint2*>(y + row * C + ch * 8) = uint2{lo, hi};
      if ((lane & 3) == 0) sy[mx_index(row, ch >> 2, MB)] = (unsigned char)sb;
    }
  }
}

}  // namespace

extern "C" {

int64_t codetr_mx_scale_bytes(int64_t M, int64_t K) { return M > 0 && K > 0 && K % 128 == 0 ? (int64_t)mx_bytes(M, K) : CODETR_E_BADARG; }

int codetr_linear_fp8mx(void* stream, const void* x8_dev, const void* x_scales_dev, const void* w8_dev,
                        const float* w_scale_dev, const void* bias_f16_dev, const void* residual_f16_dev, void* y_dev,
                        void* y_scales_dev, int64_t M, int64_t N, int64_t K, int act) {
  if (!x8_dev || !x_scales_dev || !w8_dev || !w_scale_dev || !y_dev || M <= 0 || N <= 0 || K <= 0) return CODETR_E_BADARG;
  if (K % 128 != 0 || N % 8 != 0 || act < 0 || act > 2) return CODETR_E_UNSUPPORTED;
  const bool out8 = y_scales_dev != nullptr;
  if (out8 && (residual_f16_dev || N % 128 != 0)) return CODETR_E_BADARG;
  if (M > 0x7fffffffLL || N > 0x7fffffffLL || K > 0x7fffffffLL || M * N > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  if ((reinterpret_cast<uintptr_t>(x8_dev) | reinterpret_cast<uintptr_t>(w8_dev) | reinterpret_cast<uintptr_t>(y_dev) |
       reinterpret_cast<uintptr_t>(residual_f16_dev)) & 15 || (reinterpret_cast<uintptr_t>(x_scales_dev) & 7))
    return CODETR_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto sx = static_cast<const unsigned char*>(x_scales_dev);
  auto sy = static_cast<unsigned char*>(y_scales_dev);
#define CODETR_FP8MX_ACT(A)                                                                                               \
  return out8 ? launch_bias_res<A, true, true>(st, x8_dev, w8_dev, w_scale_dev, 1.0f, bias_f16_dev, residual_f16_dev,     \
                                               y_dev, 1.0f, (int)M, (int)N, (int)K, sx, sy)                               \
              : launch_bias_res<A, false, true>(st, x8_dev, w8_dev, w_scale_dev, 1.0f, bias_f16_dev, residual_f16_dev,    \
                                                y_dev, 1.0f, (int)M, (int)N, (int)K, sx, sy)
  if (act == 0) CODETR_FP8MX_ACT(0);
  if (act == 1) CODETR_FP8MX_ACT(1);
  CODETR_FP8MX_ACT(2);
#undef CODETR_FP8MX_ACT
}

int codetr_cast_fp8mx_f16(void* stream, const void* x_f16_dev, void* y8_dev, void* y_scales_dev, int64_t rows, int64_t C) {
  if (!x_f16_dev || !y8_dev || !y_scales_dev || rows < 0 || C <= 0) return CODETR_E_BADARG;
  if (rows == 0) return 0;
  if (C % 128 != 0 || (reinterpret_cast<uintptr_t>(x_f16_dev) & 15) || (reinterpret_cast<uintptr_t>(y8_dev) & 7))
    return CODETR_E_UNSUPPORTED;
  const int64_t n16 = rows * C / 8, blocks = (n16 + 255) / 256;
  if (blocks > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(cast_fp8mx_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const s16x8*>(x_f16_dev), static_cast<uint2*>(y8_dev),
                     static_cast<unsigned char*>(y_scales_dev), n16, (int)(C / 8), mx_blocks128(rows));
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

int codetr_layernorm_fp8mx_f16(void* stream, const void* x_f16_dev, const void* gamma_f16_dev, const void* beta_f16_dev,
                               void* y8_dev, void* y_scales_dev, int64_t rows, int64_t C, float eps) {
  if (!x_f16_dev || !gamma_f16_dev || !beta_f16_dev || !y8_dev || !y_scales_dev || rows < 0 || C <= 0) return CODETR_E_BADARG;
  if (rows == 0) return 0;
  if (C % 128 != 0 || C > 4096) return CODETR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x_f16_dev) | reinterpret_cast<uintptr_t>(gamma_f16_dev) |
       reinterpret_cast<uintptr_t>(beta_f16_dev)) & 15 || (reinterpret_cast<uintptr_t>(y8_dev) & 7))
    return CODETR_E_BADARG;
  const int64_t blocks = (rows + 3) / 4;
  if (blocks > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(layernorm_fp8mx_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(x_f16_dev), static_cast<const unsigned short*>(gamma_f16_dev),
                     static_cast<const unsigned short*>(beta_f16_dev), static_cast<unsigned char*>(y8_dev),
                     static_cast<unsigned char*>(y_scales_dev), rows, (int)C, eps, mx_blocks128(rows));
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

int codetr_linear_fp8(void* stream, const void* x8_dev, const void* w8_dev, const float* w_scale_dev, float x_scale,
                      const void* bias_f16_dev, const void* residual_f16_dev, void* y_dev, int out_is_fp8, float out_scale,
                      int64_t M, int64_t N, int64_t K, int act) {
  if (!x8_dev || !w8_dev || !w_scale_dev || !y_dev || M <= 0 || N <= 0 || K <= 0) return CODETR_E_BADARG;
  if (K % 128 != 0 || N % 8 != 0 || act < 0 || act > 2) return CODETR_E_UNSUPPORTED;
  if (out_is_fp8 && (!(out_scale > 0.f) || residual_f16_dev)) return CODETR_E_BADARG;
  if (M > 0x7fffffffLL || N > 0x7fffffffLL || K > 0x7fffffffLL || M * N > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  if ((reinterpret_cast<uintptr_t>(x8_dev) | reinterpret_cast<uintptr_t>(w8_dev) | reinterpret_cast<uintptr_t>(y_dev) |
       reinterpret_cast<uintptr_t>(residual_f16_dev)) & 15)
    return CODETR_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float inv = out_is_fp8 ? 1.0f / out_scale : 1.0f;
#define CODETR_FP8_ACT(A)                                                                                                  \
  return out_is_fp8 ? launch_bias_res<A, true>(st, x8_dev, w8_dev, w_scale_dev, x_scale, bias_f16_dev, residual_f16_dev,   \
                                               y_dev, inv, (int)M, (int)N, (int)K)                                         \
                    : launch_bias_res<A, false>(st, x8_dev, w8_dev, w_scale_dev, x_scale, bias_f16_dev, residual_f16_dev,  \
                                                y_dev, inv, (int)M, (int)N, (int)K)
  if (act == 0) CODETR_FP8_ACT(0);
  if (act == 1) CODETR_FP8_ACT(1);
  CODETR_FP8_ACT(2);
#undef CODETR_FP8_ACT
}

int codetr_cast_fp8_f16(void* stream, const void* x_f16_dev, void* y8_dev, int64_t n, float scale) {
  if (!x_f16_dev || !y8_dev || n < 0 || !(scale > 0.f)) return CODETR_E_BADARG;
  if (n == 0) return 0;
  if (n % 8 != 0 || (reinterpret_cast<uintptr_t>(x_f16_dev) & 15) || (reinterpret_cast<uintptr_t>(y8_dev) & 7))
    return CODETR_E_UNSUPPORTED;
  const int64_t n16 = n / 8, blocks = (n16 + 255) / 256;
  if (blocks > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(cast_fp8_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const s16x8*>(x_f16_dev), static_cast<uint2*>(y8_dev), 1.0f / scale, n16);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

int codetr_layernorm_fp8_f16(void* stream, const void* x_f16_dev, const void* gamma_f16_dev, const void* beta_f16_dev,
                             void* y8_dev, int64_t rows, int64_t C, float eps, float scale) {
  if (!x_f16_dev || !gamma_f16_dev || !beta_f16_dev || !y8_dev || rows < 0 || C <= 0 || !(scale > 0.f))
    return CODETR_E_BADARG;
  if (rows == 0) return 0;
  if (C % 8 != 0 || C > 4096) return CODETR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x_f16_dev) | reinterpret_cast<uintptr_t>(gamma_f16_dev) |
       reinterpret_cast<uintptr_t>(beta_f16_dev)) & 15 || (reinterpret_cast<uintptr_t>(y8_dev) & 7))
    return CODETR_E_BADARG;
  const int64_t blocks = (rows + 3) / 4;
  if (blocks > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(layernorm_fp8_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(x_f16_dev), static_cast<const unsigned short*>(gamma_f16_dev),
                     static_cast<const unsigned short*>(beta_f16_dev), static_cast<unsigned char*>(y8_dev), rows, (int)C, eps,
                     1.0f / scale);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // extern "C"
