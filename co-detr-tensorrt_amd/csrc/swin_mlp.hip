// Fused Swin MLP for MI355X (gfx950), round 6:   Y = X + fc2(GELU(fc1(LayerNorm(X))))
// (reference codetr/swin.py:331-352, the second half of a SwinBlock: x = x + ffn(norm2(x)), FFN = Linear(C, 4C) -> GELU ->
// Linear(4C, C); stages 0 and 1 of Swin-L: C = 192 / 384 at 153 600 / 38 400 tokens per 1920x1280 image).
//
// As two GEMMs + a LayerNorm the hidden activation [M, 4C] is written to HBM and read back (944 / 472 MB per launch at four
// images: these layers run at 480-670 TF/s, bound by bytes -- VERDICT r05 weak 6) and norm2's output takes another round
// trip.  Here neither leaves the CU.  The structure is csrc/ffn_fused.hip's, re-derived for C that is not a power of two:
//   * persistent 256-thread workgroups (one per CU) walk 128-row tiles; a wave keeps its 32 rows of LayerNorm(X) as MFMA B
//     fragments (KS = C / 32 k-steps x 2 m-tiles) and its 32 x C slice of Y in accumulators (C / 16 n-tiles x 2).
//   * the hidden dimension is walked in chunks of BH (64 for C = 192, 32 for C = 384: 24 KiB of W1 and of W2 either way):
//       H^T[h][m] = W1c . LN(X)^T + b1 (K = C),  GELU (erf form, fp32),  fp16 pack -- the packed accumulator IS the B operand of
//       Y^T[n][m] += W2c . H^T (K = BH): the k-slot order that makes this work is baked into W2 once (codetr_ffn_pack_w2_f16,
//       the same re-layout as the encoder FFN's).
//   * W1 / W2 chunks stream through two 2-stage LDS rings by LDS-DMA (6 pieces of 4 KiB per chunk and operand, one per MFMA
//     group), one barrier + counted wait in front of each product, fragments read one step ahead of their MFMAs.
//   * LDS images.  A W1 chunk row is 2 C bytes = 24 / 48 chunks of 16 B -- not a power of two, so the bank swizzle is a
//     ROTATION: position (c + s(row)) mod (C / 8) holds source chunk c, s = (row >> 1) & 7 for 384-byte rows (rows alternate
//     between the two halves of the 256-byte bank row: 8 rotations x 2 halves = 16 distinct 16-byte slots for the 16 rows of a
//     ds_read_b128 lane group), s = row & 15 for 768-byte rows (every row starts at bank 0).  W2 chunk rows are 128 / 64 bytes:
//     the XOR keys of the GEMM kernels.
//   * W2's ROWS (output channels) are staged permuted -- LDS row 16 nt + i holds channel (i >> 2) (C / 4) + 4 nt + (i & 3) --
//     so that the lane that owns column m of the accumulator tiles (lane group g = i >> 2, register r = i & 3) holds the
//     C / 4 CONSECUTIVE channels g C / 4 ... of row m: the epilogue adds b2 and the residual (the raw rows of X, read again in
//     this layout) and stores 16 bytes per lane, a row's four lanes one contiguous run.
// Algorithmic HBM traffic: X twice in (operand layout + residual layout; the second read hits L2), Y once out; the weights
// (1.2 / 2.4 MB) come from L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "codetr_hip.h"

namespace {

constexpr int kThreads = 256;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct F16E {
  using e = _Float16;
  using v8 = f16x8;
  using v4 = f16x4;
  __device__ static f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
struct BF16E {
  using e = __bf16;
  using v8 = bf16x8;
  using v4 = bf16x4;
  __device__ static f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

// nn.GELU (erf form) on two values: gemm_elem.h's gelu_erf2 (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7)
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
  const f32x2 u = {fabsf(x.x), fabsf(x.y)};
  const f32x2 d = __builtin_elementwise_fma(f32x2{0.3275911f * 0.70710678118654752f, 0.3275911f * 0.70710678118654752f}, u, f32x2{1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2 p = __builtin_elementwise_fma(f32x2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, t, f32x2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
  const f32x2 e = (u * u) * f32x2{-0.5f * 1.4426950408889634f, -0.5f * 1.4426950408889634f};
  const f32x2 ez = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
  const f32x2 h = __builtin_elementwise_fma(-(p * t), ez, f32x2{0.5f, 0.5f});
  return __builtin_elementwise_fma(u, h, x * f32x2{0.5f, 0.5f});
}

__device__ __forceinline__ void dma16(const unsigned char* src, unsigned voff, unsigned char* dst) {
  const unsigned lds_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)dst);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// C = 192: BH = 64, C = 384: BH = 32, one workgroup per CU (WPE = 1); C = 192 also as BH = 32 with TWO workgroups per CU
// (WPE = 2: 12 KiB chunks, 256 registers per wave -- the second wave of a SIMD runs its MFMAs under this one's GELU)
template <class ET, int C, int BH, int WPE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void swin_mlp_kernel(
    const unsigned short* __restrict__ X, const unsigned short* __restrict__ ln_g, const unsigned short* __restrict__ ln_b,
    const float ln_eps, const unsigned short* __restrict__ W1, const unsigned short* __restrict__ b1,
    const unsigned short* __restrict__ W2p, const unsigned short* __restrict__ b2, unsigned short* __restrict__ Y, const int M,
    const int ntiles) {
  using E = typename ET::e;
  using V8 = typename ET::v8;
  using V4 = typename ET::v4;
  constexpr int Hd = 4 * C, NCHUNK = Hd / BH; // 12 / 48 chunks
  constexpr int KS = C / 32;                  // k-steps of the first product: 6 / 12
  constexpr int HT = BH / 16;                 // h-tiles of a chunk: 4 / 2
  constexpr int KS2 = BH / 32;                // k-steps of the second product: 2 / 1
  constexpr int NT = C / 16;                  // n-tiles of Y: 12 / 24
  constexpr int NCH = C / 8;                  // 16-byte chunks of a W1 row: 24 / 48
  constexpr int RB1 = 2 * C;                  // bytes of a W1 row
  constexpr int RB2 = 2 * BH;                 // bytes of a staged W2 row: 128 / 64
  constexpr int MT = 2, WR = 32, TR = 128;    // m-tiles per wave, rows per wave / workgroup
  constexpr int kChunkBytes = BH * RB1;       // one chunk of W1 = one chunk of W2: 24 KiB (12 KiB in the WPE = 2 form)
  constexpr int NP = kChunkBytes / (kThreads * 16);   // LDS-DMA pieces of a chunk and operand: 6 / 3
  static_assert(C * RB2 == kChunkBytes && NP * kThreads * 16 == kChunkBytes && KS % NP == 0 && (NT / 2) % NP == 0, "chunk geometry");
  // [W1 stage 0 | W1 stage 1 | W2 stage 0 | W2 stage 1 | b1 | b2]: 96 KiB + 3.75 KiB
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kChunkBytes + Hd * 2 + C * 2];
  unsigned char* const ringA = lds;
  unsigned char* const ringB = lds + 2 * kChunkBytes;
  unsigned short* const sB1 = reinterpret_cast<unsigned short*>(lds + 4 * kChunkBytes);
  unsigned short* const sB2 = sB1 + Hd;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, grp = lane >> 4;

  // ---- LDS-DMA geometry: 6 pieces of 256 x 16 B per chunk and operand ----
  // W1 chunk [BH rows][NCH positions]: linear 16-byte index L = 256 p + tid -> row L / NCH, position L % NCH, which holds
  // source chunk (position - s(row)) mod NCH
  unsigned w1_voff[NP], w2_voff[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int Lx = 256 * p + tid;
    const int r = Lx / NCH, pos = Lx - r * NCH;
    const int s = C == 192 ? (r >> 1) & 7 : r & 15;
    int c = pos - s;
    c = c < 0 ? c + NCH : c;
    w1_voff[p] = (unsigned)(r * RB1 + c * 16);
    // W2 chunk [C rows][RB2 / 16 positions]: piece p covers rows (RB2 == 128 ? 32 : 64) p + ...; LDS row rho holds channel
    // (i >> 2) (C / 4) + 4 nt + (i & 3), nt = rho / 16, i = rho % 16; position q holds source chunk q ^ key(rho)
    constexpr int PPR = RB2 / 16;   // positions per row: 8 / 4
    const int rho = (256 / PPR) * p + tid / PPR, q = tid % PPR;
    const int i = rho & 15, nt = rho >> 4;
    const int ch = (i >> 2) * (C / 4) + 4 * nt + (i & 3);
    int key;
    if (RB2 == 128) key = (rho >> 1) & 7;
    else {
      const int qq = (rho >> 2) & 3;
      key = qq ^ ((qq & 1) << 1);
    }
    w2_voff[p] = (unsigned)(ch * (Hd * 2) + ((q ^ key) * 16));
  }
  const unsigned char* const W1b = reinterpret_cast<const unsigned char*>(W1);
  const unsigned char* const W2b = reinterpret_cast<const unsigned char*>(W2p);
  auto stage_w1 = [&](int p, int c, unsigned char* dst) {   // piece p of chunk c
    dma16(W1b + (size_t)c * kChunkBytes, w1_voff[p], dst + (p * kThreads + wave * 64) * 16);
  };
  auto stage_w2 = [&](int p, int c, unsigned char* dst) {   // (chunk c = packed columns c BH .. of every row)
    dma16(W2b + (size_t)c * RB2, w2_voff[p], dst + (p * kThreads + wave * 64) * 16);
  };
#pragma unroll
  for (int p = 0; p < NP; ++p) stage_w1(p, 0, ringA);
#pragma unroll
  for (int p = 0; p < NP; ++p) stage_w2(p, 0, ringB);

  // biases -> LDS once per workgroup
  for (int i = tid; i < Hd / 8; i += kThreads) *reinterpret_cast<u32x4*>(sB1 + i * 8) = *reinterpret_cast<const u32x4*>(b1 + i * 8);
  if (tid < C / 8) *reinterpret_cast<u32x4*>(sB2 + tid * 8) = *reinterpret_cast<const u32x4*>(b2 + tid * 8);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the first barrier below publishes them

  // fragment read offsets (per lane): W1 row 16 ht + l15, chunk 4 ks + grp at position (chunk + s) mod NCH
  const int s1 = C == 192 ? (l15 >> 1) & 7 : l15;
  const int gs1 = grp + s1;                                   // position = 4 ks + gs1 (- NCH if it wraps)
  const unsigned rowoff1 = (unsigned)(l15 * RB1);
  int key2;
  if (RB2 == 128) key2 = (l15 >> 1) & 7;
  else {
    const int qq = (l15 >> 2) & 3;
    key2 = qq ^ ((qq & 1) << 1);
  }
  const unsigned rowoff2 = (unsigned)(l15 * RB2);

  int gc = 0;   // chunks consumed so far: ring stage = gc & 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * TR + wave * WR;
    // ---- the tile's rows as MFMA B fragments (lane (l15 = row, grp) holds X[m][32 ks + 8 grp .. + 7]), LayerNorm in registers
    // (fp32 two-pass over the row's C values: KS x 8 in-lane + the four lanes of the row; result rounded to E = the tensor
    // norm2 would have written)
    V8 xf[MT][KS];
    // (opaque per tile: left visible, the compiler hoists the loop-invariant gamma / beta loads out of the tile loop and keeps
    // their 2 x C / 4 values per lane alive across it -- 160 spilled registers at C = 384)
    const unsigned short* lg = ln_g;
    const unsigned short* lb = ln_b;
    asm volatile("" : "+s"(lg), "+s"(lb));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int m = m0 + mt * 16 + l15;
      m = m < M ? m : M - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[mt][ks] = *reinterpret_cast<const V8*>(X + (size_t)m * C + ks * 32 + grp * 8);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float sm = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += (float)xf[mt][ks][e];
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      const float mean = sm * (1.0f / C);
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
      const float rstd = rsqrtf(q * (1.0f / C) + ln_eps);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const V8 gw = *reinterpret_cast<const V8*>(lg + ks * 32 + grp * 8);
        const V8 gb = *reinterpret_cast<const V8*>(lb + ks * 32 + grp * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[mt][ks][e] = (E)fmaf(((float)xf[mt][ks][e] - mean) * rstd, (float)gw[e], (float)gb[e]);
      }
    }
    // (the rows were requested AFTER the pieces of the tile's first chunks -- issued before the loop or during the previous
    // tile's last chunk -- and the LayerNorm above has consumed them: vmcnt retires in order, so those pieces have landed)
    f32x4 yacc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) yacc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline over the chunks (ring stage of chunk c = (gc + c) & 1; vmcnt counts the LDS-DMA pieces in issue
    // order): the GELU of chunk c -- 32 / 16 evaluations per lane, as many issue cycles as the chunk's 96 MFMAs for C = 192
    // -- runs INSIDE the second product of chunk c - 1, one or two evaluations per MFMA group, instead of between the two
    // products of its own chunk where nothing overlaps it (one wave per SIMD):
    //   iteration c:  T: wait until W1[c] landed (the 6 younger pieces are W2[c - 1]'s), barrier
    //                 product 1 of chunk c, the 6 pieces of W1[c + 1] spread over its MFMA groups
    //                 M: wait until W2[c - 1] landed (the 6 younger pieces are W1[c + 1]'s), barrier
    //                 product 2 of chunk c - 1 with the GELU + pack of chunk c inside, the 6 pieces of W2[c] spread over it
    //   (iteration 0 has no second product: GELU alone, W2[0]'s pieces issued in one burst; behind the loop the second
    //   product of the last chunk alone).  W1[c + 1] goes to the stage product 1 of iteration c - 1 read, W2[c] to the one
    //   product 2 of iteration c - 1 read: every wave passed T of iteration c since.  Chunk NCHUNK wraps to chunk 0 of the
    //   next tile.
    auto product1 = [&](int c, f32x4 (&hacc)[HT][MT]) {
      const int cn = c + 1 < NCHUNK ? c + 1 : 0;
      const unsigned char* sW1 = ringA + ((gc + c) & 1) * kChunkBytes;
      unsigned char* nW1 = ringA + ((gc + c + 1) & 1) * kChunkBytes;
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
        const V4 bv = *reinterpret_cast<const V4*>(sB1 + c * BH + ht * 16 + grp * 4);
        const f32x4 b4 = {(float)bv[0], (float)bv[1], (float)bv[2], (float)bv[3]};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hacc[ht][mt] = b4;
      }
      auto read_w1 = [&](int ks, V8 (&a)[HT]) {
        int pos = 4 * ks + gs1;
        pos = pos >= NCH ? pos - NCH : pos;
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
          a[ht] = *reinterpret_cast<const V8*>(sW1 + ht * 16 * RB1 + rowoff1 + pos * 16);
      };
      V8 aw[2][HT];
      read_w1(0, aw[0]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) read_w1(ks + 1, aw[(ks + 1) & 1]);
        if ((ks * NP) % KS == 0) stage_w1(ks * NP / KS, cn, nW1);
#pragma unroll
        for (int i = 0; i < HT * MT; ++i) hacc[i / MT][i % MT] = ET::mfma(aw[ks & 1][i / MT], xf[i % MT][ks], hacc[i / MT][i % MT]);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, HT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, HT * MT - 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // GELU + pack of accumulator quad q (0 .. HT * MT - 1: h-tile q / MT, m-tile q % MT), half hh (0: registers 0, 1; 1:
    // registers 2, 3): B operand of the second product, k-slot 8 g + j = rows 4 g .. 4 g + 3 of tiles 2 s and 2 s + 1
    auto gelu_half = [&](const f32x4 (&hacc)[HT][MT], V8 (&pf)[KS2][MT], int q, int hh) {
      const int ht = q / MT, mt = q % MT;
      const f32x4 v = hacc[ht][mt];
      const f32x2 g = gelu2(hh ? f32x2{v[2], v[3]} : f32x2{v[0], v[1]});
      pf[ht >> 1][mt][(ht & 1) * 4 + 2 * hh] = (E)g.x;
      pf[ht >> 1][mt][(ht & 1) * 4 + 2 * hh + 1] = (E)g.y;
    };
    constexpr int NG = NT / 2;               // MFMA groups of the second product: 6 / 12
    constexpr int NGELU = 2 * HT * MT;       // GELU half-quads of a chunk: 16 / 8
    // second product of chunk c (operand pf), the 6 pieces of W2[cs] spread over its groups when `stage`; when `next` is
    // given, the GELU of the NEXT chunk's accumulators runs inside
    auto product2 = [&](int c, const V8 (&pf)[KS2][MT], auto overlap_c, int cs, const f32x4 (&hnext)[HT][MT], V8 (&pfn)[KS2][MT]) {
      constexpr bool overlap = decltype(overlap_c)::value;   // false: the tile's last second product (nothing staged, no GELU)
      const unsigned char* sW2 = ringB + ((gc + c) & 1) * kChunkBytes;
      unsigned char* nW2 = ringB + ((gc + c + 1) & 1) * kChunkBytes;
      auto read_w2 = [&](int ntp, V8 (&a)[2 * KS2]) {   // n-tiles 2 ntp, 2 ntp + 1; index [t * KS2 + s]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int s_ = 0; s_ < KS2; ++s_)
            a[t * KS2 + s_] = *reinterpret_cast<const V8*>(sW2 + (2 * ntp + t) * 16 * RB2 + rowoff2 + (((4 * s_ + grp) ^ key2) * 16));
      };
      V8 a2[2][2 * KS2];
      read_w2(0, a2[0]);
#pragma unroll
      for (int ntp = 0; ntp < NG; ++ntp) {
        if (ntp + 1 < NG) read_w2(ntp + 1, a2[(ntp + 1) & 1]);
        if (overlap && (ntp * NP) % NG == 0) stage_w2(ntp * NP / NG, cs, nW2);
#pragma unroll
        for (int i = 0; i < 2 * KS2 * MT; ++i) {   // MFMA index i -> (t, s, mt)
          const int ts = i / MT, mt = i % MT, nt = 2 * ntp + ts / KS2;
          yacc[nt][mt] = ET::mfma(a2[ntp & 1][ts], pf[ts % KS2][mt], yacc[nt][mt]);
        }
        if (overlap) {   // this group's share of the next chunk's GELU: half-quads [ntp * NGELU / NG, (ntp + 1) * NGELU / NG)
#pragma unroll
          for (int k = ntp * NGELU / NG; k < (ntp + 1) * NGELU / NG; ++k) gelu_half(hnext, pfn, k >> 1, k & 1);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * KS2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * KS2 * MT - 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    f32x4 hacc[HT][MT];
    V8 pfa[KS2][MT], pfb[KS2][MT];
    // ---- iteration 0 ----
    __builtin_amdgcn_s_barrier();   // T (no counted wait: see above)
    product1(0, hacc);
#pragma unroll
    for (int k = 0; k < NGELU; ++k) gelu_half(hacc, pfa, k >> 1, k & 1);
#pragma unroll
    for (int p = 0; p < NP; ++p) stage_w2(p, 0, ringB + (gc & 1) * kChunkBytes);
    // ---- iterations 1 .. NCHUNK - 1, two per trip (the operand registers alternate statically) ----
    for (int c = 1; c < NCHUNK; c += 2) {
      wait_vmcnt<NP>();
      __builtin_amdgcn_s_barrier();   // T
      product1(c, hacc);
      wait_vmcnt<NP>();
      __builtin_amdgcn_s_barrier();   // M
      product2(c - 1, pfa, std::true_type{}, c, hacc, pfb);
      if (c + 1 < NCHUNK) {
        wait_vmcnt<NP>();
        __builtin_amdgcn_s_barrier();   // T
        product1(c + 1, hacc);
        wait_vmcnt<NP>();
        __builtin_amdgcn_s_barrier();   // M
        product2(c, pfb, std::true_type{}, c + 1, hacc, pfa);
      }
    }
    // ---- the last chunk's second product (NCHUNK is even: its operand is pfb) ----
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // M'
    product2(NCHUNK - 1, pfb, std::false_type{}, 0, hacc, pfa);
    gc += NCHUNK;

    // ---- epilogue out of the accumulators: lane (l15 = row m of the m-tile, group g) holds, for nt < NT and r < 4,
    // Y[m][g C / 4 + 4 nt + r] (the W2 row permutation): C / 4 consecutive channels.  + b2 -> E, + the raw row of X -> E,
    // 16 bytes per lane and store
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + l15;
      const int mc = m < M ? m : M - 1;
      const unsigned short* xr = X + (size_t)mc * C + grp * (C / 4);
      unsigned short* yr = Y + (size_t)mc * C + grp * (C / 4);
      V8 res[NT / 2];
#pragma unroll
      for (int k = 0; k < NT / 2; ++k) res[k] = *reinterpret_cast<const V8*>(xr + 8 * k);
#pragma unroll
      for (int k = 0; k < NT / 2; ++k) {
        const V8 bb = *reinterpret_cast<const V8*>(sB2 + grp * (C / 4) + 8 * k);
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float y = yacc[2 * k + (e >> 2)][mt][e & 3] + (float)bb[e];
          o[e] = (E)((float)(E)y + (float)res[k][e]);   // E(fc2) + identity -> E: the reference's two roundings
        }
        if (m < M) *reinterpret_cast<V8*>(yr + 8 * k) = o;
      }
    }
  }
  wait_vmcnt<0>();   // the redundant pieces fetched during the last chunk
}

int device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

template <class ET>
int launch_swin_mlp(void* stream, const void* x, const void* g, const void* b, float eps, const void* w1, const void* b1,
                    const void* w2p, const void* b2, void* y, int64_t M, int64_t C) {
  if (!x || !g || !b || !w1 || !b1 || !w2p || !b2 || !y || M <= 0) return CODETR_E_BADARG;
  if (C != 192 && C != 384) return CODETR_E_UNSUPPORTED;
  if (M > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(w1) |
       reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(w2p) | reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(y)) & 15)
    return CODETR_E_BADARG;
  const int ntiles = (int)((M + 127) / 128);
#ifndef SWIN_MLP_192_WPE
#define SWIN_MLP_192_WPE 2
#endif
  constexpr int kWpe192 = SWIN_MLP_192_WPE;   // workgroups per CU of the C = 192 form (2: 32-wide hidden chunks)
  const int slots = device_cus() * (C == 192 ? kWpe192 : 1);
  const int grid = ntiles < slots ? ntiles : slots;
  auto X = static_cast<const unsigned short*>(x);
#define CODETR_SWIN_MLP(CC, BH, WPE)                                                                                      \
  hipLaunchKernelGGL((swin_mlp_kernel<ET, CC, BH, WPE>), dim3((unsigned)grid), dim3(kThreads), 0, static_cast<hipStream_t>(stream), X,  \
                     static_cast<const unsigned short*>(g), static_cast<const unsigned short*>(b), eps,                   \
                     static_cast<const unsigned short*>(w1), static_cast<const unsigned short*>(b1),                       \
                     static_cast<const unsigned short*>(w2p), static_cast<const unsigned short*>(b2),                      \
                     static_cast<unsigned short*>(y), (int)M, ntiles)
  if (C == 192) CODETR_SWIN_MLP(192, (kWpe192 == 2 ? 32 : 64), kWpe192);
  else CODETR_SWIN_MLP(384, 32, 1);
#undef CODETR_SWIN_MLP
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int codetr_swin_mlp_supported(int64_t M, int64_t C, int64_t hidden) {
  return (C == 192 || C == 384) && hidden == 4 * C && M > 0 && M <= 0x7fffffffLL ? 1 : 0;
}

int codetr_swin_mlp_f16(void* stream, const void* x_dev, const void* ln_gamma_dev, const void* ln_beta_dev, float ln_eps,
                        const void* w1_dev, const void* b1_dev, const void* w2_packed_dev, const void* b2_dev, void* y_dev,
                        int64_t M, int64_t C) {
  return launch_swin_mlp<F16E>(stream, x_dev, ln_gamma_dev, ln_beta_dev, ln_eps, w1_dev, b1_dev, w2_packed_dev, b2_dev, y_dev, M, C);
}

int codetr_swin_mlp_bf16(void* stream, const void* x_dev, const void* ln_gamma_dev, const void* ln_beta_dev, float ln_eps,
                         const void* w1_dev, const void* b1_dev, const void* w2_packed_dev, const void* b2_dev, void* y_dev,
                         int64_t M, int64_t C) {
  return launch_swin_mlp<BF16E>(stream, x_dev, ln_gamma_dev, ln_beta_dev, ln_eps, w1_dev, b1_dev, w2_packed_dev, b2_dev, y_dev, M, C);
}

}  // extern "C"
