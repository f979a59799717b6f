// Fused transformer FFN, two waves per SIMD (round 4):  Y = LN(X' + relu(X' . W1^T + b1) . W2^T + b2),  X' = X or LN_in(X)
// (reference codetr/transformer_mmcv.py:484-500 and the post-norm layer around it, transformer.py:50-90).
//
// ffn_fused.hip runs this with ONE wave per SIMD (4 waves x 32 rows, 428 registers): its bare MFMA stream reaches 78 % of
// what a two-waves-per-SIMD stream reaches (profiles/r03_ffn_ablation.txt) because every non-MFMA instruction of a lone
// wave -- accumulator init, fragment reads, the 100+ cycles an LDS-DMA piece takes to issue, ReLU / pack -- lands in the
// matrix pipe's issue stream with nobody to cover it (the same finding as profiles/r04_gemm_sk.txt, step 1).  The row
// tile of a wave cannot shrink (one W fragment read per two MFMAs is already the LDS budget), so here TWO waves share a
// wave's 32 rows and split the work along the other axes:
//   * 512 threads = 8 waves per 128-row tile; waves w and w + 4 (the two waves of a SIMD) own rows 32 w .. 32 w + 31;
//   * product 1 (H^T = W1c . X^T, 64 hidden units per chunk = 4 tiles of 16): member h of the pair computes hidden tiles
//     2h and 2h + 1 -- exactly the two tiles whose packed ReLU'ed accumulators form k-step h of the second product's B
//     operand ("accumulator tile as the next MFMA's operand", same W2 pre-packing as ffn_fused.hip);
//   * the members exchange that packed fragment through LDS (2 KiB per wave and chunk, written before the barrier that
//     already separates the products, read behind it while the k-step a member owns itself is being multiplied);
//   * product 2 (Y^T += W2c . relu(H)^T): member h accumulates output columns 128 h .. 128 h + 127 (8 tiles of 16): 64
//     accumulator registers instead of 128 -- the whole wave fits 256 registers, two waves per SIMD;
//   * per chunk and wave: 32 + 32 MFMAs, 16 + 16 fragment reads, 4 + 4 LDS-DMA pieces (512 threads move 8 KiB per piece);
//     the same two barriers and counted waits per chunk as ffn_fused.hip;
//   * epilogue out of the accumulators as before (lanes 16 apart swap halves: 16-byte row accesses), on the member's 128
//     columns; the LayerNorm statistics of a row are combined across the pair with the pairwise-variance formula
//     (mean and centred sum of squares of each half: the two-pass result with one 8-byte exchange per row).
// Same entry points (codetr_ffn_relu_ln2_*): this kernel serves hidden <= 4096, ffn_fused.hip the rest.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

constexpr int C = 256;         // model width (K of the first product, N of the second)
constexpr int BH = 64;         // hidden units per chunk
constexpr int kThreads = 512;
constexpr int kW1Bytes = BH * C * 2;   // 32 KiB: [64 h][256 k]
constexpr int kW2Bytes = C * BH * 2;   // 32 KiB: [256 n][64 h]
constexpr int kStageBytes = kW1Bytes + kW2Bytes;
constexpr int kMaxHidden8 = 4096;      // b1 lives in LDS behind the two stages (8 KiB)
constexpr int kXchBytes = 8 * 2048;    // packed hidden fragments of the 8 waves
constexpr int kStatBytes = 8 * 32 * 8; // (mean, M2) of a wave's 32 rows

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct F16E {
  using e = _Float16;
  using v8 = f16x8;
  using v4 = f16x4;
  __device__ static f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  __device__ static v8 relu(v8 x) {
    const v8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_elementwise_max(x, z);
  }
};
struct BF16E {
  using e = __bf16;
  using v8 = bf16x8;
  using v4 = bf16x4;
  __device__ static f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  __device__ static v8 relu(v8 x) {
    s16x8 i;
    __builtin_memcpy(&i, &x, 16);
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    i = __builtin_elementwise_max(i, z);
    __builtin_memcpy(&x, &i, 16);
    return x;
  }
};

// one LDS-DMA piece: 512 threads x 16 B = 8 KiB (see ffn_fused.hip dma16)
__device__ __forceinline__ void dma16(const unsigned char* src, unsigned voff, unsigned char* dst) {
  const unsigned lds_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)dst);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds_addr)
               : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the whole persistent loop of one wave; HALF = member of the pair (a template parameter: the two members pick different
// registers as MFMA operands, which a run-time choice would turn into select instructions)
template <class ET, int HALF>
__device__ __forceinline__ void ffn8_body(
    unsigned char* lds, const unsigned short* __restrict__ X, const unsigned short* __restrict__ W1,
    const unsigned short* __restrict__ b1, const unsigned short* __restrict__ W2, const unsigned short* __restrict__ b2,
    unsigned short* __restrict__ Y, int M, int Hd, const unsigned short* __restrict__ ln_g,
    const unsigned short* __restrict__ ln_b, float ln_eps, const unsigned short* __restrict__ pos,
    unsigned short* __restrict__ Y2, const unsigned short* __restrict__ lnin_g, const unsigned short* __restrict__ lnin_b,
    float lnin_eps, int ntiles) {
  using E = typename ET::e;
  using V8 = typename ET::v8;
  using V4 = typename ET::v4;
  constexpr int MT = 2;   // 16-row tiles per wave (32 rows per pair of waves, 128 per workgroup)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3;   // row group of the pair
  constexpr int half = HALF;
  const int l15 = lane & 15, grp = lane >> 4;
  const int nchunks = Hd / BH;
  unsigned char* ringA = lds;                   // W1 chunks [64 h][256 k]
  unsigned char* ringB = lds + 2 * kW1Bytes;    // packed W2 chunks [256 n][64 h]
  unsigned short* sB1 = reinterpret_cast<unsigned short*>(lds + 2 * kStageBytes);
  unsigned short* sLn = reinterpret_cast<unsigned short*>(lds + 2 * kStageBytes + kMaxHidden8 * 2);  // gamma[256], beta[256]
  unsigned short* sB2 = sLn + 512;
  unsigned char* xch = lds + 2 * kStageBytes + kMaxHidden8 * 2 + 1536;
  float* stat = reinterpret_cast<float*>(xch + kXchBytes);
  unsigned char* my_xch = xch + wave * 2048;
  const unsigned char* partner_xch = xch + (wave ^ 4) * 2048;

  // LDS-DMA geometry, 8 KiB pieces.  W1 chunk: 64 rows x 32 chunks of 16 B; piece p (0..3) = rows 16 p + (tid >> 5),
  // position tid & 31 of row r holds source chunk (tid & 31) ^ (r & 15) (= (tid >> 5): piece-independent).  W2 chunk: 256
  // rows x 8 chunks; piece q (0..3) = rows 64 q + (tid >> 3), position tid & 7 holds chunk (tid & 7) ^ ((row >> 1) & 7).
  const unsigned w1_voff = (unsigned)((tid >> 5) * (C * 2) + (((tid & 31) ^ ((tid >> 5) & 15)) * 16));
  const unsigned w2_voff = (unsigned)((tid >> 3) * Hd * 2 + (((tid & 7) ^ ((tid >> 4) & 7)) * 16));
  const unsigned char* W1b = reinterpret_cast<const unsigned char*>(W1);
  const unsigned char* W2b = reinterpret_cast<const unsigned char*>(W2);
  auto stage_w1 = [&](int p, int c, unsigned char* dst) {
    dma16(W1b + (size_t)c * kW1Bytes + p * 8192, w1_voff, dst + (p * kThreads + wave * 64) * 16);
  };
  auto stage_w2 = [&](int q, int c, unsigned char* dst) {
    dma16(W2b + (size_t)q * 128 * Hd + c * (BH * 2), w2_voff, dst + (q * kThreads + wave * 64) * 16);
  };
#pragma unroll
  for (int p = 0; p < 4; ++p) stage_w1(p, 0, ringA);
#pragma unroll
  for (int p = 0; p < 4; ++p) stage_w2(p, 0, ringB);

  // the pair's input rows (both members hold them: B-operand layout, lane (j = l15, g) holds X[m][32 ks + 8 g .. + 7])
  V8 x[MT][8];
  auto load_x = [&](int tile, int mt) {
    int m = tile * 128 + rg * 32 + mt * 16 + l15;
    m = m < M ? m : M - 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) x[mt][ks] = *reinterpret_cast<const V8*>(X + (size_t)m * C + ks * 32 + grp * 8);
  };
  load_x(blockIdx.x, 0);
  load_x(blockIdx.x, 1);
  for (int i = tid; i < Hd / 8; i += kThreads)
    *reinterpret_cast<s16x8*>(sB1 + i * 8) = *reinterpret_cast<const s16x8*>(b1 + i * 8);
  if (ln_g && tid < 64) {
    const int i = tid & 31;
    *reinterpret_cast<s16x8*>(sLn + (tid >> 5) * 256 + i * 8) = *reinterpret_cast<const s16x8*>((tid >> 5 ? ln_b : ln_g) + i * 8);
  }
  if (tid >= 64 && tid < 96) *reinterpret_cast<s16x8*>(sB2 + (tid - 64) * 8) = *reinterpret_cast<const s16x8*>(b2 + (tid - 64) * 8);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // tables written; the first barrier below publishes them

  int gc = 0;  // chunks consumed so far: ring stage = gc & 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * 128 + rg * 32;
    // ---- the MFMA operand of the first product and the identity: the rows, or their LayerNorm (in place) ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (lnin_g) {
        float sm = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) sm += (float)x[mt][ks][e];
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        const float mean = sm * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = (float)x[mt][ks][e] - mean;
            q = fmaf(d, d, q);
          }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.0f / C) + lnin_eps);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const V8 gw = *reinterpret_cast<const V8*>(lnin_g + ks * 32 + grp * 8);
          const V8 gb = *reinterpret_cast<const V8*>(lnin_b + ks * 32 + grp * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            x[mt][ks][e] = (E)fmaf(((float)x[mt][ks][e] - mean) * rstd, (float)gw[e], (float)gb[e]);
        }
      }
      // chunk 0 of a tile takes no counted wait: it relies on the tile's input rows -- issued AFTER the LDS-DMA pieces of
      // W1[0] / W2[0] -- having landed (vmcnt retires in order); make the dependence explicit
      asm volatile("" ::"v"(x[mt][7]) : "memory");
    }
    f32x4 yacc[8][MT];   // [n-tile of this member's 128 columns][m-tile]
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) yacc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Schedule of one chunk c (ring stage gc & 1; vmcnt counts LDS-DMA pieces, loads and stores in issue order):
    //   T: wait until W1[c] landed (the 4 younger pieces are W2[c]'s), barrier (not in a tile's chunk 0, see above)
    //   product 1 on the member's two hidden tiles, one piece of W1[c+1] every other k-step
    //   ReLU + pack -> the member's k-step of the second product's B operand; written to the exchange buffer
    //   M: wait until W2[c] landed (the 4 younger pieces are W1[c+1]'s), barrier (also publishes the exchange buffer)
    //   product 2 on the member's 8 output tiles: own k-step first, the partner's (read from LDS meanwhile) second;
    //   one piece of W2[c+1] per pair of output tiles
    for (int c = 0; c < nchunks; ++c, ++gc) {
      const int cn = c + 1 < nchunks ? c + 1 : 0;
      const unsigned char* sW1 = ringA + (gc & 1) * kW1Bytes;
      const unsigned char* sW2 = ringB + (gc & 1) * kW2Bytes;
      unsigned char* nW1 = ringA + ((gc + 1) & 1) * kW1Bytes;
      unsigned char* nW2 = ringB + ((gc + 1) & 1) * kW2Bytes;
      if (c > 0) wait_vmcnt<4>();
      __builtin_amdgcn_s_barrier();  // T

      // ---- H^T = W1c . X^T on hidden tiles 2 half, 2 half + 1 : D[i = h][j = m] ----
      f32x4 hacc[2][MT];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const V4 bv = *reinterpret_cast<const V4*>(sB1 + c * BH + (2 * half + t) * 16 + grp * 4);
        const f32x4 b4 = {(float)bv[0], (float)bv[1], (float)bv[2], (float)bv[3]};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hacc[t][mt] = b4;
      }
      auto read_w1 = [&](int ks, V8 (&a)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row = (2 * half + t) * 16 + l15;
          const int chunk = (ks * 4 + grp) ^ (row & 15);
          a[t] = *reinterpret_cast<const V8*>(sW1 + row * (C * 2) + chunk * 16);
        }
      };
      V8 aw[2][2];
      read_w1(0, aw[0]);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) read_w1(ks + 1, aw[(ks + 1) & 1]);
        if (ks & 1) stage_w1(ks >> 1, cn, nW1);
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) hacc[i / MT][i % MT] = ET::mfma(aw[ks & 1][i / MT], x[i % MT][ks], hacc[i / MT][i % MT]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * MT - 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- ReLU + pack: k-step `half` of the second product's B operand (k-slot 8g+j = rows 4g..4g+3 of the two tiles) ----
      V8 pown[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) pown[mt][h * 4 + r] = (E)hacc[h][mt][r];
        pown[mt] = ET::relu(pown[mt]);
        *reinterpret_cast<V8*>(my_xch + (mt * 64 + lane) * 16) = pown[mt];
      }
      if (c > 0) wait_vmcnt<4>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the exchange writes have left this wave
      __builtin_amdgcn_s_barrier();  // M
      V8 pother[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) pother[mt] = *reinterpret_cast<const V8*>(partner_xch + (mt * 64 + lane) * 16);

      // ---- Y^T += W2c . relu(H)^T on output tiles 8 half .. 8 half + 7 : D[i = n][j = m] ----
      // W2 fragments (pre-packed k-slots), one pair of n-tiles ahead; index [t * 2 + s], s = k-step
      auto read_w2 = [&](int ntp, V8 (&a)[4]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int n = (8 * half + 2 * ntp + t) * 16 + l15;
          const unsigned char* rowp = sW2 + n * (BH * 2);
          const int sw = (n >> 1) & 7;
#pragma unroll
          for (int s = 0; s < 2; ++s) a[t * 2 + s] = *reinterpret_cast<const V8*>(rowp + ((4 * s + grp) ^ sw) * 16);
        }
      };
      V8 a2[2][4];
      read_w2(0, a2[0]);
#pragma unroll
      for (int ntp = 0; ntp < 4; ++ntp) {
        if (ntp + 1 < 4) read_w2(ntp + 1, a2[(ntp + 1) & 1]);
        stage_w2(ntp, cn, nW2);
        // own k-step first: the partner's fragment is still on its way from LDS for the first pair of tiles
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            yacc[2 * ntp + t][mt] = ET::mfma(half ? a2[ntp & 1][t * 2 + 1] : a2[ntp & 1][t * 2], pown[mt], yacc[2 * ntp + t][mt]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            yacc[2 * ntp + t][mt] = ET::mfma(half ? a2[ntp & 1][t * 2] : a2[ntp & 1][t * 2 + 1], pother[mt], yacc[2 * ntp + t][mt]);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT - 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue out of the accumulators: lane (column l15 = row m of the tile, group g) holds
    // Y^T[n = 128 half + 16 nt + 4 g + r][m].  Lanes g and g ^ 1 swap halves so that every lane owns 8 CONSECUTIVE channels
    // of 4 of the member's 8 tiles: channels 128 half + 32 j + cbase .. + 7, j < 4 (see ffn_fused.hip).  The identity
    // X'[m][32 (4 half + j) + 8 v ..] (v = cbase / 8) is register x[mt][4 half + j] of lane group v.
    const int next_tile = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;
    const int odd = grp & 1;
    const int cbase = 16 * odd + 8 * (grp >> 1);
    const int src_lane4 = (l15 + 16 * (2 * odd + (grp >> 1))) * 4;   // byte address of the lane that holds the identity
    float o[MT][4][8];
    float mean_own[MT], m2_own[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      unsigned yp[8][2];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const V4 bb = *reinterpret_cast<const V4*>(sB2 + (8 * half + nt) * 16 + grp * 4);
        const V4 y = {(E)(yacc[nt][mt][0] + (float)bb[0]), (E)(yacc[nt][mt][1] + (float)bb[1]),
                      (E)(yacc[nt][mt][2] + (float)bb[2]), (E)(yacc[nt][mt][3] + (float)bb[3])};
        __builtin_memcpy(yp[nt], &y, 8);
      }
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned s0 = odd ? yp[2 * j][0] : yp[2 * j + 1][0], s1 = odd ? yp[2 * j][1] : yp[2 * j + 1][1];
        const unsigned k0 = odd ? yp[2 * j + 1][0] : yp[2 * j][0], k1 = odd ? yp[2 * j + 1][1] : yp[2 * j][1];
        const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16, 64), r1 = (unsigned)__shfl_xor((int)s1, 16, 64);
        const unsigned z[4] = {odd ? r0 : k0, odd ? r1 : k1, odd ? k0 : r0, odd ? k1 : r1};   // channels cbase .. + 7
        V8 yv, xid;
        __builtin_memcpy(&yv, z, 16);
        // the identity: register x[mt][4 half + j] (a wave-uniform choice between two registers) of lane group v
        const V8 xsel = half ? x[mt][4 + j] : x[mt][j];
        int xw[4];
        __builtin_memcpy(xw, &xsel, 16);
#pragma unroll
        for (int d = 0; d < 4; ++d) xw[d] = __builtin_amdgcn_ds_bpermute(src_lane4, xw[d]);
        __builtin_memcpy(&xid, xw, 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[mt][j][e] = (float)(E)((float)yv[e] + (float)xid[e]);   // identity + ffn(x): E + E -> E
          sm += o[mt][j][e];
        }
      }
      // statistics of the member's 128 channels of the row: mean and centred sum of squares (32 in-lane values + the four
      // lanes of a row)
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      mean_own[mt] = sm * (1.0f / 128.f);
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = o[mt][j][e] - mean_own[mt];
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      m2_own[mt] = q;
      if (ln_g && grp == 0) *reinterpret_cast<float2*>(stat + (wave * 32 + mt * 16 + l15) * 2) = float2{mean_own[mt], q};
    }
    // the rows of the next tile: x is dead from here on (both m-tiles' identities are in o)
    load_x(next_tile, 0);
    load_x(next_tile, 1);
    if (ln_g) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // the pair's statistics are visible
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + l15;
      V8 pr[4];
      if (Y2) {
        const unsigned short* prow = pos + (size_t)(m < M ? m : M - 1) * C + 128 * half + cbase;
#pragma unroll
        for (int j = 0; j < 4; ++j) pr[j] = *reinterpret_cast<const V8*>(prow + 32 * j);
      }
      if (ln_g) {
        // pairwise combination of the two halves (Chan et al.): n_a = n_b = 128
        const float2 ot = *reinterpret_cast<const float2*>(stat + ((wave ^ 4) * 32 + mt * 16 + l15) * 2);
        const float dm = ot.x - mean_own[mt];
        const float mean = mean_own[mt] + 0.5f * dm;
        const float m2 = m2_own[mt] + ot.y + dm * dm * 64.f;
        const float rstd = rsqrtf(m2 * (1.0f / C) + ln_eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const V8 gw = *reinterpret_cast<const V8*>(sLn + 128 * half + 32 * j + cbase);
          const V8 gb = *reinterpret_cast<const V8*>(sLn + 256 + 128 * half + 32 * j + cbase);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[mt][j][e] = (float)(E)fmaf((o[mt][j][e] - mean) * rstd, (float)gw[e], (float)gb[e]);
        }
      }
      if (m < M) {
        unsigned short* yrow = Y + (size_t)m * C + 128 * half + cbase;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          V8 ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = (E)o[mt][j][e];
          *reinterpret_cast<V8*>(yrow + 32 * j) = ov;
        }
        if (Y2) {  // the next layer's attention input: this row + its positional encoding (E + E -> E)
          unsigned short* y2row = Y2 + (size_t)m * C + 128 * half + cbase;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            V8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (E)(o[mt][j][e] + (float)pr[j][e]);
            *reinterpret_cast<V8*>(y2row + 32 * j) = ov;
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <class ET>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void ffn8_fused_kernel(
    const unsigned short* __restrict__ X, const unsigned short* __restrict__ W1, const unsigned short* __restrict__ b1,
    const unsigned short* __restrict__ W2, const unsigned short* __restrict__ b2, unsigned short* __restrict__ Y, int M,
    int Hd, const unsigned short* __restrict__ ln_g, const unsigned short* __restrict__ ln_b, float ln_eps,
    const unsigned short* __restrict__ pos, unsigned short* __restrict__ Y2,
    const unsigned short* __restrict__ lnin_g, const unsigned short* __restrict__ lnin_b, float lnin_eps, int ntiles) {
  // [W1 stage 0 | W1 stage 1 | W2 stage 0 | W2 stage 1 | b1 | LayerNorm gamma, beta | b2 | hidden exchange | row statistics]
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kStageBytes + kMaxHidden8 * 2 + 1536 + kXchBytes + kStatBytes];
  // both members run the same number of barriers; only their operand registers differ
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8))
    ffn8_body<ET, 1>(lds, X, W1, b1, W2, b2, Y, M, Hd, ln_g, ln_b, ln_eps, pos, Y2, lnin_g, lnin_b, lnin_eps, ntiles);
  else
    ffn8_body<ET, 0>(lds, X, W1, b1, W2, b2, Y, M, Hd, ln_g, ln_b, ln_eps, pos, Y2, lnin_g, lnin_b, lnin_eps, ntiles);
}

template <class ET>
int launch8(hipStream_t st, const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* y, int64_t M,
            int64_t hidden, const void* lnin_g, const void* lnin_b, float lnin_eps, const void* ln_g, const void* ln_b,
            float ln_eps, const void* pos, void* y2) {
  const int ntiles = (int)((M + 127) / 128);
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const unsigned blocks = (unsigned)(ntiles < cus ? ntiles : cus);
  hipLaunchKernelGGL((ffn8_fused_kernel<ET>), dim3(blocks), dim3(kThreads), 0, st, static_cast<const unsigned short*>(x),
                     static_cast<const unsigned short*>(w1), static_cast<const unsigned short*>(b1),
                     static_cast<const unsigned short*>(w2), static_cast<const unsigned short*>(b2),
                     static_cast<unsigned short*>(y), (int)M, (int)hidden, static_cast<const unsigned short*>(ln_g),
                     static_cast<const unsigned short*>(ln_b), ln_eps, static_cast<const unsigned short*>(pos),
                     static_cast<unsigned short*>(y2), static_cast<const unsigned short*>(lnin_g),
                     static_cast<const unsigned short*>(lnin_b), lnin_eps, ntiles);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

// called by ffn_fused.hip's entry points after they have validated the arguments (hidden % 64 == 0, 16-byte alignment):
// returns CODETR_E_UNSUPPORTED when this kernel does not take the shape (the one-wave-per-SIMD kernel then runs)
int codetr_ffn8_launch(int bf16, void* stream, const void* x, const void* w1, const void* b1, const void* w2, const void* b2,
                       void* y, int64_t M, int64_t hidden, const void* lnin_g, const void* lnin_b, float lnin_eps,
                       const void* ln_g, const void* ln_b, float ln_eps, const void* pos, void* y2) {
  if (hidden > kMaxHidden8) return CODETR_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return bf16 ? launch8<BF16E>(st, x, w1, b1, w2, b2, y, M, hidden, lnin_g, lnin_b, lnin_eps, ln_g, ln_b, ln_eps, pos, y2)
              : launch8<F16E>(st, x, w1, b1, w2, b2, y, M, hidden, lnin_g, lnin_b, lnin_eps, ln_g, ln_b, ln_eps, pos, y2);
}
