// Fused (shifted-)window multi-head self-attention for the Swin backbone on MI355X (gfx950).
//
// One kernel replaces what the reference runs, per Swin block, as ~15 ATen kernels
// (reference codetr/swin.py:191-252 and :92-112): zero-pad to a multiple of the window, cyclic
// roll, window partition, q*scale, q@k^T, + relative-position bias, + shift mask, softmax,
// attn@v, head merge, window reverse, reverse roll, crop.  Here none of the intermediate tensors
// exist: the kernel reads q/k/v straight from the qkv GEMM output in SPATIAL token order
// [B, H*W, 3C] and writes the attention output in spatial order [B, H*W, C]; padding, roll and
// window (un)partitioning are index arithmetic on the loads and stores.
//
// Pad tokens.  The reference pads AFTER norm1 with zeros, so a pad token's qkv row equals the qkv
// bias, and pad tokens take part in the softmax as ordinary keys (only the shift mask exists).
// The qkv GEMM therefore runs on real tokens only and this kernel substitutes the bias vector for
// the q/k/v of pad tokens.
//
// Mapping: one wave per (image, window, head); 4 waves per workgroup, no inter-wave traffic.
//   * K [N x 32] and V [N x 32] (N = ws*ws = 144 tokens, head_dim 32 = 64-byte rows) are staged
//     into the wave's LDS region (16-byte chunks, XOR-swizzled by (row>>2)&3);
//   * per 16-query tile: S^T = K . Q^T with v_mfma_f32_16x16x32_f16 (K = head_dim, one MFMA per
//     16x16 score tile, Q fragment straight from global memory), scores kept TRANSPOSED so that a
//     lane owns 4 consecutive keys of ONE query: scale, + bias (8-byte loads of the gathered
//     [nH,N,N] bias), + shift mask (region ids from LDS), softmax with two cross-lane reductions;
//   * the fp16 probabilities are already laid out as the B operand of the next MFMA
//     (cdna_hip_programming.md section 3, "accumulator tile as the next MFMA's operand"): O^T = V^T . P^T,
//     V^T fragments come from the row-major V image through ds_read_b64_tr_b16 (hardware
//     transpose), the k-slot permutation being the same on both operands;
//   * O^T leaves as 8-byte stores (4 consecutive channels of one token), normalised by 1/rowsum.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "codetr_hip.h"
#include "mx_scale.h"

namespace {

constexpr int HD = 32;  // head_dim of every Swin variant
constexpr int kWaves = 4;
constexpr int kThreads = 64 * kWaves;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// element types: fp16 / bf16 storage, fp32 scores and accumulation either way
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

struct Geometry {
  int B, H, W, Hp, Wp, shift, nH, nWx, nWin;  // nWin = (Hp/ws)*(Wp/ws)
};

template <int WS>
struct Tok {
  int token;    // y*W + x in the unpadded map (valid only if `valid`)
  bool valid;   // false: pad token -> q/k/v = qkv bias, output dropped
  int region;   // shift-mask region id 0..8
};

template <int WS>
__device__ __forceinline__ Tok<WS> map_token(int i, int wy, int wx, const Geometry& g) {
  const int iy = i / WS, ix = i - iy * WS;
  const int ys = wy * WS + iy, xs = wx * WS + ix;  // coordinates in the shifted, padded map
  int y = ys + g.shift, x = xs + g.shift;          // roll(-shift): shifted[ys] = padded[(ys + shift) mod Hp]
  y = y >= g.Hp ? y - g.Hp : y;
  x = x >= g.Wp ? x - g.Wp : x;
  Tok<WS> t;
  t.valid = (y < g.H) && (x < g.W);
  t.token = y * g.W + x;
  const int rh = ys < g.Hp - WS ? 0 : (ys < g.Hp - g.shift ? 1 : 2);
  const int rw = xs < g.Wp - WS ? 0 : (xs < g.Wp - g.shift ? 1 : 2);
  t.region = g.shift > 0 ? rh * 3 + rw : 0;
  return t;
}

// XOR swizzle of the 16-byte chunk of a 64-byte (head_dim 32) LDS row.  With j = (row >> 2) & 3 the key f(j) = j ^ ((j & 1) << 1)
// = {0, 3, 2, 1} keeps both read shapes conflict-free under the hardware's lane groups (tests/test_lds_bank_model.py):
// the K fragments (ds_read_b128: a group of 16 lanes holds rows {0-3, 12-15} at chunk g and rows {4-11} at chunk g ^ 1)
// and the transposing V reads (ds_read_b64_tr_b16: 32 lanes = 8 rows x two chunks; rows r and r + 4 alias mod 256 B and
// need keys that differ in bit 1).  The plain key f(j) = j is 2-way conflicted on both.
__device__ __forceinline__ int swz(int row, int chunk) {
  const int j = (row >> 2) & 3;
  return chunk ^ j ^ ((j & 1) << 1);
}

// OUT8: the output is e4m3 = sat(f16(o) * out_inv_scale) (the operand of the fp8 proj GEMM, BASELINE config 5) instead of
// 16-bit -- what codetr_cast_fp8_f16 would make of the 16-bit output, without that tensor's round trip
// OUTMX (with OUT8): block-scaled e4m3 instead -- one e8m0 byte per (token, head) = per 32 channels, written to `out_scales`
// in the consumer GEMM's layout (mx_scale.h); out_inv_scale is unused
// PB: the relative-position bias arrives in the LANE ORDER of the score tiles (codetr_window_attention_bias_index): row q of
// head h holds, for lane group g, the keys 16 kt + 4 g + r of every key tile kt next to each other -- a lane's values of a
// query tile are 8 NT contiguous bytes (72 for the 12 x 12 window: five loads, every line fetched once) instead of NT 8-byte
// pieces 32 bytes apart (nine loads of sixteen 32-byte row segments each, and the L1 does not hold the rows between them:
// -4 ... -12 % per launch, profiles/r06_window_attention.txt).  N % 16 == 0 only.
template <class ET, int WS, bool OUT8 = false, bool OUTMX = false, bool PB = false>
__global__ __launch_bounds__(kThreads) void window_attention_kernel(
    const typename ET::e* __restrict__ qkv,       // [B, H*W, 3C]
    const typename ET::e* __restrict__ qkv_bias,  // [3C] (zeros if the layer has no bias)
    const typename ET::e* __restrict__ rel_bias,  // [nH, N, N]
    void* __restrict__ out_v,                     // [B, H*W, C] 16-bit, or e4m3 bytes (OUT8)
    Geometry g, int n_problems, float out_inv_scale, unsigned char* __restrict__ out_scales) {
  using E = typename ET::e;
  using V8 = typename ET::v8;
  using V4 = typename ET::v4;
  constexpr int N = WS * WS;
  constexpr int NT = (N + 15) / 16;   // 16-token tiles
  constexpr int NP = NT * 16;         // padded token count of K
  constexpr int KS = (NT + 1) / 2;    // 32-key steps of the P.V product
  constexpr int VR = KS * 32;         // padded rows of V (zero-filled beyond N)
  constexpr int kWaveLds = NP * 64 + VR * 64 + NP;  // K image, V image, region ids
  constexpr int kWaveLdsAligned = (kWaveLds + 15) & ~15;
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[kWaves * kWaveLdsAligned];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int prob = blockIdx.x * kWaves + wave;
  if (prob >= n_problems) return;  // whole wave exits; waves never synchronise with each other
  const int head = prob % g.nH;
  const int win = (prob / g.nH) % g.nWin;
  const int b = prob / (g.nH * g.nWin);
  const int wy = win / g.nWx, wx = win - wy * g.nWx;
  const int C = g.nH * HD;
  const size_t row_elems = (size_t)3 * C;
  const E* qkv_b = qkv + (size_t)b * g.H * g.W * row_elems;
  const int hoff = head * HD;

  unsigned char* ldsK = lds_all + wave * kWaveLdsAligned;
  unsigned char* ldsV = ldsK + NP * 64;
  unsigned char* ldsR = ldsV + VR * 64;

  // ---------------- stage K and V (+ region ids) ----------------
  // all 2 * ITER row loads are requested before the first LDS write (one exposed latency instead of ITER: a rolled
  // load -> write loop cost ~10 us of a ~27 us problem)
  constexpr int ITER = (VR * 4 + 63) / 64;
  s16x8 kv[ITER], vv[ITER];
  unsigned char regv[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int c = lane + it * 64;
    const int row = c >> 2, chunk = c & 3;
    kv[it] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    vv[it] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    regv[it] = 0;
    if (c < VR * 4 && row < N) {
      const Tok<WS> t = map_token<WS>(row, wy, wx, g);
      const E* src = t.valid ? qkv_b + (size_t)t.token * row_elems : qkv_bias;
      kv[it] = *reinterpret_cast<const s16x8*>(src + C + hoff + chunk * 8);
      vv[it] = *reinterpret_cast<const s16x8*>(src + 2 * C + hoff + chunk * 8);
      regv[it] = (unsigned char)t.region;
    }
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int c = lane + it * 64;
    const int row = c >> 2, chunk = c & 3;
    if (c < VR * 4) {
      if (chunk == 0 && row < NP) ldsR[row] = regv[it];
      const int pos = swz(row, chunk) * 16;
      if (row < NP) *reinterpret_cast<s16x8*>(ldsK + row * 64 + pos) = kv[it];
      *reinterpret_cast<s16x8*>(ldsV + row * 64 + pos) = vv[it];
    }
  }
  __builtin_amdgcn_wave_barrier();  // LDS ops of one wave retire in order; keep the compiler from reordering

  const int l15 = lane & 15, grp = lane >> 4;
  const float log2e = 1.4426950408889634f;
  const float scale = 0.17677669529663687f;   // head_dim^-0.5
  const E* bias_h = rel_bias + (size_t)head * N * N;

  // per-lane addresses that do not depend on the query tile
  // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4 x 16 block
  const int tr_q = l15 >> 2, tr_p = l15 & 3;

  // The query fragment and the 9 bias groups of a query tile are requested one tile ahead: with one wave per problem
  // and two waves per SIMD nothing else covers a global-load latency, and unprefetched they cost two of them per
  // tile (18 per problem: ~17 of the ~20 us a problem took).
  struct QTile {
    V8 qf;
    V4 bv[NT];
    Tok<WS> tq;
    bool q_in;
  };
  auto load_qtile = [&](int qt, QTile& T) {
    const int qi = qt * 16 + l15;
    T.q_in = (N % 16 == 0) || qi < N;
    T.tq = map_token<WS>(T.q_in ? qi : 0, wy, wx, g);
    const E* qsrc = (T.tq.valid ? qkv_b + (size_t)T.tq.token * row_elems : qkv_bias) + hoff + grp * 8;
    T.qf = *reinterpret_cast<const V8*>(qsrc);
    if constexpr (PB) {
      static_assert(!PB || N % 16 == 0, "lane-order bias: whole key tiles only");
      typedef V8 __attribute__((aligned(4))) V8u;
      typedef V4 __attribute__((aligned(4))) V4u;
      const E* pb = bias_h + (size_t)qi * N + grp * (4 * NT);
#pragma unroll
      for (int j = 0; j < NT / 2; ++j) {
        const V8 w = *reinterpret_cast<const V8u*>(pb + 8 * j);
        T.bv[2 * j] = V4{w[0], w[1], w[2], w[3]};
        T.bv[2 * j + 1] = V4{w[4], w[5], w[6], w[7]};
      }
      if constexpr (NT & 1) T.bv[NT - 1] = *reinterpret_cast<const V4u*>(pb + 4 * (NT - 1));
      return;
    }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const int key0 = kt * 16 + grp * 4;
      V4 bv = {(E)0.f, (E)0.f, (E)0.f, (E)0.f};
      if constexpr (N % 4 == 0) {  // rows of the bias are 8-byte aligned and a 4-key group is all in or all out
        if ((N % 16 == 0) || (T.q_in && key0 < N))
          bv = *reinterpret_cast<const V4*>(bias_h + (size_t)(T.q_in ? qi : 0) * N + key0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (T.q_in && key0 + r < N) bv[r] = bias_h[(size_t)qi * N + key0 + r];
      }
      T.bv[kt] = bv;
    }
  };
  const bool multi_region = g.shift > 0 && (wy == g.nWin / g.nWx - 1 || wx == g.nWx - 1);
  // Software pipeline over the query tiles: the score half of tile t + 1 (S^T MFMAs, scale / bias / mask, row max) and the
  // probability half of tile t (exp, row sums, P.V, store) are two independent instruction streams in ONE basic block, so
  // that with two waves per SIMD the MFMA / LDS / cross-lane latencies of one are covered by the vector work of the other
  // (one tile per iteration: a wave waits to issue for 48 % of its life and its vector instructions are active 30 %,
  // profiles/r06_window_attention.txt).  The shift-mask branch is hoisted out of the
  // loop (wave-uniform), the prefetch index is clamped instead of guarded: no control flow inside an iteration.
  auto run = [&](auto mask_c) {
    auto stage1 = [&](const QTile& cur, f32x4 (&s)[NT], float& mx) {
      const Tok<WS> tq = cur.tq;
      const V8 qf = cur.qf;
      // ---- S^T tiles: D[i = key][j = query] ----
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        const int row = kt * 16 + l15;
        const V8 kf = *reinterpret_cast<const V8*>(ldsK + row * 64 + swz(row, grp) * 16);
        s[kt] = ET::mfma(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f});
      }
      // ---- scale, bias, mask, row max ----
      // (Pre-multiplying the bias table by log2 e on the host -- one fma per score instead of mul + fma -- measured -2 %
      // and loses the large-bias cases to f16 rounding of the scaled table: not done.)
      // The shift mask costs ~3 of the ~12 VALU instructions per score, and this kernel is VALU-bound (two waves per
      // SIMD, ~6 k instructions per problem): only windows on the last window row / column of a shifted block hold more
      // than one region, everything else takes the mask-free instantiation (wave-uniform branch).
      mx = -INFINITY;
      auto scores = [&](auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          const int key0 = kt * 16 + grp * 4;
          const V4 bv = cur.bv[kt];
          unsigned regk = 0;
          if (MASK) regk = *reinterpret_cast<const unsigned*>(ldsR + key0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // (bias * log2e + score * scale: the 16-bit bias goes into the fused multiply-add directly -- v_fma_mix_f32 --
            // instead of a conversion, a multiply and an fma)
            // natural-log domain here: ONE v_fma_mix_f32 per score (fp32 score x scale + the 16-bit bias); the log2 e factor
            // rides in the fma in front of the exponential below (the separate loop multiplies first: one instruction more)
            float v = fmaf(s[kt][r], scale, (float)bv[r]);
            if (MASK && (int)((regk >> (8 * r)) & 0xff) != tq.region) v -= 100.0f;
            if ((N % 16 != 0) && key0 + r >= N) v = -INFINITY;
            s[kt][r] = v;
            mx = fmaxf(mx, v);
          }
        }
      };
      scores(mask_c);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    };
    auto stage2 = [&](const QTile& cur, f32x4 (&s)[NT], const float mx) {
      const bool q_in = cur.q_in;
      const Tok<WS> tq = cur.tq;
      // ---- exp, row sum, pack P^T as MFMA B fragments ----
      // The row sums come from the matrix pipe (8 % busy in this kernel, the vector pipe is the bound): one more MFMA per
      // 32-key step with an all-ones A operand gives D[i][query] = sum over the keys of P^T[key][query] for every i -- the sum
      // of exactly the rounded probabilities that multiply V, in this lane's own query column.  Replaces an add per score
      // and two cross-lane reductions per query tile.
      const float nmx = -mx * log2e;
      V8 pf[KS];
      V8 ones;
#pragma unroll
      for (int e = 0; e < 8; ++e) ones[e] = (E)1.0f;
      f32x4 rs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int kt = 2 * ks + h;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = 0.f;
            if (kt < NT) p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], log2e, nmx));  // v_exp_f32; argument <= 0
            pf[ks][h * 4 + r] = (E)p;
          }
        }
        rs = ET::mfma(ones, pf[ks], rs);
      }
      const float sum = rs[0];
      // ---- O^T = V^T . P^T : D[i = channel][j = query] ----
      f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int chunk = dt * 2 + (tr_p >> 1);
          const int row0 = (2 * ks) * 16 + grp * 4 + tr_q;
          const int row1 = row0 + 16;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(ldsV + row0 * 64 + swz(row0, chunk) * 16 + (tr_p & 1) * 8));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(ldsV + row1 * 64 + swz(row1, chunk) * 16 + (tr_p & 1) * 8));
          s16x8 vf8 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          V8 vf;
          __builtin_memcpy(&vf, &vf8, 16);
          o[dt] = ET::mfma(vf, pf[ks], o[dt]);
        }
      }
      // ---- normalise and store: lane holds channels 16*dt + 4*grp + r of query l15 ----
      if constexpr (OUT8 && OUTMX) {
        // the 32 channels of (token, head) sit in the four lanes l15 + 16 grp: block maximum by two xor-shuffles (every
        // lane takes part; the shuffles stay outside the validity branch)
        const float inv = 1.0f / sum;
        float f[2][4], amax = 0.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            f[dt][r] = (float)(E)(o[dt][r] * inv);
            amax = fmaxf(amax, fabsf(f[dt][r]));
          }
        amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        if (q_in && tq.valid) {
          const unsigned sb = mx_e8m0(amax);
          const float qs = mx_inv_scale(sb);
          const size_t trow = (size_t)b * g.H * g.W + tq.token;
          const size_t doff = trow * C + hoff + grp * 4;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            float q4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) q4[r] = __builtin_amdgcn_fmed3f(f[dt][r] * qs, -448.0f, 448.0f);
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(q4[0], q4[1], 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(q4[2], q4[3], w, true);
            *reinterpret_cast<int*>(static_cast<unsigned char*>(out_v) + doff + dt * 16) = w;
          }
          if (grp == 0) out_scales[mx_index((int64_t)trow, head, mx_blocks128((int64_t)g.B * g.H * g.W))] = (unsigned char)sb;
        }
      } else if (q_in && tq.valid) {
        const float inv = 1.0f / sum;
        const size_t doff = ((size_t)b * g.H * g.W + tq.token) * C + hoff + grp * 4;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          V4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (E)(o[dt][r] * inv);
          if constexpr (OUT8) {
            float f[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) f[r] = __builtin_amdgcn_fmed3f((float)ov[r] * out_inv_scale, -448.0f, 448.0f);
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
            *reinterpret_cast<int*>(static_cast<unsigned char*>(out_v) + doff + dt * 16) = w;
          } else {
            *reinterpret_cast<V4*>(static_cast<E*>(out_v) + doff + dt * 16) = ov;
          }
        }
      }
    };
    QTile tA, tB;
    load_qtile(0, tA);
    load_qtile(NT > 1 ? 1 : 0, tB);
    f32x4 sA[NT];
    float mxA;
    stage1(tA, sA, mxA);
#pragma unroll 2
    for (int qt = 0; qt < NT - 1; ++qt) {
      QTile tC;
      load_qtile(qt + 2 < NT ? qt + 2 : NT - 1, tC);
      f32x4 sB[NT];
      float mxB;
      stage1(tB, sB, mxB);
      stage2(tA, sA, mxA);
      tA = tB;
      tB = tC;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) sA[kt] = sB[kt];
      mxA = mxB;
    }
    stage2(tA, sA, mxA);
  };
  if (multi_region) run(std::true_type{});
  else run(std::false_type{});
}

template <class ET, int WS, bool OUT8, bool OUTMX = false, bool PB = false>
int launch_ws(hipStream_t st, const void* qkv, const void* qkv_bias, const void* rel_bias, void* out, Geometry g,
              float out_inv_scale, unsigned char* out_scales = nullptr) {
  const int64_t n = (int64_t)g.B * g.nWin * g.nH;
  if (n > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  const unsigned blocks = (unsigned)((n + kWaves - 1) / kWaves);
  hipLaunchKernelGGL((window_attention_kernel<ET, WS, OUT8, OUTMX, PB>), dim3(blocks), dim3(kThreads), 0, st,
                     static_cast<const typename ET::e*>(qkv), static_cast<const typename ET::e*>(qkv_bias),
                     static_cast<const typename ET::e*>(rel_bias), out, g, (int)n, out_inv_scale, out_scales);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

template <class ET, bool OUT8 = false, bool OUTMX = false>
int window_attention_entry(void* stream, const void* qkv_dev, const void* qkv_bias_dev, const void* rel_bias_dev,
                                void* out_dev, int64_t B, int64_t H, int64_t W, int num_heads, int head_dim,
                                int window_size, int shift, float out_inv_scale = 1.0f,
                                unsigned char* out_scales = nullptr, int bias_layout = 0) {
  if (!qkv_dev || !qkv_bias_dev || !rel_bias_dev || !out_dev || B <= 0 || H <= 0 || W <= 0 || num_heads <= 0)
    return CODETR_E_BADARG;
  if (head_dim != HD || shift < 0 || shift >= window_size) return CODETR_E_UNSUPPORTED;
  if (B * H * W > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  Geometry g;
  g.B = (int)B;
  g.H = (int)H;
  g.W = (int)W;
  g.Hp = (int)((H + window_size - 1) / window_size * window_size);
  g.Wp = (int)((W + window_size - 1) / window_size * window_size);
  g.shift = shift;
  g.nH = num_heads;
  g.nWx = g.Wp / window_size;
  g.nWin = (g.Hp / window_size) * g.nWx;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (bias_layout == 1) {   // lane order: whole key tiles only
    switch (window_size) {
      case 12: return launch_ws<ET, 12, OUT8, OUTMX, true>(st, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, g, out_inv_scale, out_scales);
      case 8: return launch_ws<ET, 8, OUT8, OUTMX, true>(st, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, g, out_inv_scale, out_scales);
      case 4: return launch_ws<ET, 4, OUT8, OUTMX, true>(st, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, g, out_inv_scale, out_scales);
    }
    return CODETR_E_UNSUPPORTED;
  }
  if (bias_layout != 0) return CODETR_E_BADARG;
  switch (window_size) {
    case 12: return launch_ws<ET, 12, OUT8, OUTMX>(st, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, g, out_inv_scale, out_scales);
    case 8: return launch_ws<ET, 8, OUT8, OUTMX>(st, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, g, out_inv_scale, out_scales);
    case 7: return launch_ws<ET, 7, OUT8, OUTMX>(st, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, g, out_inv_scale, out_scales);
    case 4: return launch_ws<ET, 4, OUT8, OUTMX>(st, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, g, out_inv_scale, out_scales);
  }
  return CODETR_E_UNSUPPORTED;
}


}  // namespace

extern "C" {

int codetr_window_attention_f16(void* stream, const void* qkv_dev, const void* qkv_bias_dev, const void* rel_bias_dev,
                                void* out_dev, int64_t B, int64_t H, int64_t W, int num_heads, int head_dim,
                                int window_size, int shift) {
  return window_attention_entry<F16E>(stream, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, B, H, W, num_heads, head_dim,
                                      window_size, shift);
}

int codetr_window_attention_bf16(void* stream, const void* qkv_dev, const void* qkv_bias_dev, const void* rel_bias_dev,
                                 void* out_dev, int64_t B, int64_t H, int64_t W, int num_heads, int head_dim,
                                 int window_size, int shift) {
  return window_attention_entry<BF16E>(stream, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, B, H, W, num_heads,
                                       head_dim, window_size, shift);
}

int codetr_window_attention_fp8mx_f16(void* stream, const void* qkv_dev, const void* qkv_bias_dev,
                                      const void* rel_bias_dev, void* out8_dev, void* out_scales_dev, int64_t B, int64_t H,
                                      int64_t W, int num_heads, int head_dim, int window_size, int shift) {
  if (!out_scales_dev) return CODETR_E_BADARG;
  if ((num_heads * (int64_t)head_dim) % 128 != 0 || (reinterpret_cast<uintptr_t>(out8_dev) & 3)) return CODETR_E_UNSUPPORTED;
  return window_attention_entry<F16E, true, true>(stream, qkv_dev, qkv_bias_dev, rel_bias_dev, out8_dev, B, H, W, num_heads,
                                                  head_dim, window_size, shift, 1.0f,
                                                  static_cast<unsigned char*>(out_scales_dev));
}

int codetr_window_attention_fp8out_f16(void* stream, const void* qkv_dev, const void* qkv_bias_dev,
                                       const void* rel_bias_dev, void* out8_dev, float out_scale, int64_t B, int64_t H,
                                       int64_t W, int num_heads, int head_dim, int window_size, int shift) {
  if (!(out_scale > 0.f)) return CODETR_E_BADARG;
  if ((num_heads * (int64_t)head_dim) % 4 != 0 || (reinterpret_cast<uintptr_t>(out8_dev) & 3)) return CODETR_E_BADARG;
  return window_attention_entry<F16E, true>(stream, qkv_dev, qkv_bias_dev, rel_bias_dev, out8_dev, B, H, W, num_heads,
                                            head_dim, window_size, shift, 1.0f / out_scale);
}

// One entry for every form above plus the bias layout (see include/codetr_hip.h).
int codetr_window_attention_ex(void* stream, const void* qkv_dev, const void* qkv_bias_dev, const void* rel_bias_dev,
                               void* out_dev, void* out_scales_dev, float out_scale, int64_t B, int64_t H, int64_t W,
                               int num_heads, int head_dim, int window_size, int shift, int elem, int out_mode,
                               int bias_layout) {
  if (elem != 0 && elem != 1) return CODETR_E_BADARG;
  if (out_mode == 0) {
    if (elem == 1)
      return window_attention_entry<BF16E>(stream, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, B, H, W, num_heads, head_dim,
                                           window_size, shift, 1.0f, nullptr, bias_layout);
    return window_attention_entry<F16E>(stream, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, B, H, W, num_heads, head_dim,
                                        window_size, shift, 1.0f, nullptr, bias_layout);
  }
  if (elem != 0) return CODETR_E_UNSUPPORTED;   // the e4m3 outputs take fp16 qkv
  if (out_mode == 1) {
    if (!(out_scale > 0.f)) return CODETR_E_BADARG;
    if ((num_heads * (int64_t)head_dim) % 4 != 0 || (reinterpret_cast<uintptr_t>(out_dev) & 3)) return CODETR_E_BADARG;
    return window_attention_entry<F16E, true>(stream, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, B, H, W, num_heads,
                                              head_dim, window_size, shift, 1.0f / out_scale, nullptr, bias_layout);
  }
  if (out_mode == 2) {
    if (!out_scales_dev) return CODETR_E_BADARG;
    if ((num_heads * (int64_t)head_dim) % 128 != 0 || (reinterpret_cast<uintptr_t>(out_dev) & 3)) return CODETR_E_UNSUPPORTED;
    return window_attention_entry<F16E, true, true>(stream, qkv_dev, qkv_bias_dev, rel_bias_dev, out_dev, B, H, W, num_heads,
                                                    head_dim, window_size, shift, 1.0f,
                                                    static_cast<unsigned char*>(out_scales_dev), bias_layout);
  }
  return CODETR_E_BADARG;
}

// idx_host[j] (j < window_size^2) = the key whose bias sits at position j of a lane-order row: position 4 NT g + 4 kt + r
// holds key 16 kt + 4 g + r (NT = window_size^2 / 16 key tiles).  Window sizes whose token count is not a multiple of 16 have
// no lane-order form.
int codetr_window_attention_bias_index(int window_size, int32_t* idx_host) {
  if (!idx_host || window_size <= 0) return CODETR_E_BADARG;
  const int N = window_size * window_size;
  if (N % 16 != 0 || (window_size != 12 && window_size != 8 && window_size != 4)) return CODETR_E_UNSUPPORTED;
  const int NT = N / 16;
  for (int g = 0; g < 4; ++g)
    for (int kt = 0; kt < NT; ++kt)
      for (int r = 0; r < 4; ++r) idx_host[4 * NT * g + 4 * kt + r] = 16 * kt + 4 * g + r;
  return 0;
}

}  // extern "C"
