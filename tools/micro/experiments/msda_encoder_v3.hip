// EXPERIMENT / A-B baseline, not part of libcodetr_hip.so since round 6 (dead weight in the product
// library): the round-3/4 encoder MSDA kernels -- the generic single-pass LDS-staged kernel (bit-identical to the general
// kernel) and the three-pass packed-half kernel ("v3") that csrc/msda_encoder4.hip replaced in round 5.  Their C-ABI entry
// points (codetr_msda_encoder_forward[_win]_{f16,bf16}, codetr_msda_encoder_lds_bytes), host wrappers and tests left the
// product with this file; build it as its own shared object to compare (profiles/r05_msda_encoder4_sweep.txt has the numbers).
// Encoder self-attention form of the fused multi-scale deformable attention for MI355X (gfx950 / CDNA4).
//
// Same arithmetic as msda_tiled_kernel<.., FUSED> (msda_forward.hip): the op of reference
// codetr/csrc/ms_deform_attn.cu:31-77, 211-261 plus the softmax / sampling-location steps of
// codetr/multi_scale_deformable_attention.py:180-196 -- bit-identical results, different data movement.
//
// In DetrTransformerEncoder (reference codetr/transformer.py:81-92) the queries ARE the pixels of the flattened
// multi-level map and the reference point of a query is its own pixel centre: the samples of a query fall in
// the neighbourhood of the same image location on every level.  The general kernel gathers each 64-byte value
// row through the vector L1 (32 KB, 40 % hit rate at the model shape): ~5 GB per call cross the L2 -> L1 path,
// which bounds it at ~510 us per 1920x1280 image.  Here the gather runs out of LDS:
//
//   * a workgroup owns one (image, REGION, head): a region is 16 x 8 level-0 pixels (64 x 32 image pixels) and
//     contains the queries of EVERY level whose centre lies in it (128 + 32 + 8 + 2 + 0.5 at the model's pyramid);
//   * it first copies, per level, the head's 64-byte rows of the region's neighbourhood (region extent +- `halo`
//     pixels of that level + the bilinear corner) into LDS: 73 KB at halo 4 -> two workgroups per CU, one loading
//     while the other gathers; every value row crosses L2 -> CU once per (region, head) instead of once per sample;
//   * the 4 lanes that serve a (query, head) pair prepare its L*P sample points in registers (softmax over the
//     quad by DPP, same formulas as build_entries) and hand corner addresses / weights to each other by DPP
//     quad broadcasts -- no LDS round trip for the entries; the 80 corner rows of a pair are then ds_read_b128's
//     (lanes of a wave are x-neighbouring queries -> neighbouring LDS rows, conflict-free) blended with
//     v_fma_mix_f32 in the same order as the general kernel;
//   * a sample whose corners leave the staged neighbourhood (offset larger than the halo, padded images whose
//     valid ratios skew the reference points) is read from global memory instead: the result never depends on the
//     halo, only the speed does.  A wave takes the checked loop only when one of its 16 pairs has such a sample;
//   * raw offsets / logits / reference points of the NEXT 16 queries of a wave are requested before the current
//     ones are consumed (one exposed latency per workgroup, not per iteration);
//   * the M heads of a region sit on one XCD back to back (their 64-byte slices of the same 512-byte pixel rows
//     meet in that XCD's L2, and so do their 64-byte slices of each output row).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#include "codetr_hip.h"

#ifdef MSDA_ENC_ABLATE
#include "../../tools/micro/diag_env.h"
#endif

namespace {

constexpr int kThreads = 256;
constexpr int kRegW = 16, kRegH = 8;  // region, in level-0 pixels
constexpr int kMaxL = 8;
constexpr int kMaxM = 16;      // heads with a staged window of their own
constexpr int kMetaInts = 28;  // per level: 7 x 16 B (see region_geometry)
constexpr int kMaxLds = 160 * 1024;
constexpr int kPad = 64;       // one row slot in front of and behind the staged rows (see pass_v3)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

struct F16 {
  using storage = _Float16;
  using vec = f16x8;
  __device__ static float to_f32(storage v) { return (float)v; }
  __device__ static storage from_f32(float v) { return (_Float16)v; }
};
struct BF16 {
  using storage = unsigned short;
  using vec = u16x8;
  __device__ static float to_f32(storage v) { return __uint_as_float(((unsigned)v) << 16); }
  __device__ static storage from_f32(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
};

struct EncGeom {
  int L, P, M, RX, RY, S, rows_cap, slots_cap, band;
  int H[kMaxL], W[kMaxL], start[kMaxL];
  float invH[kMaxL], invW[kMaxL];  // 1.0f / H, 1.0f / W, correctly rounded (what the general kernel divides out)
  // staged WINDOW per (head, level): offsets (pixels of that level, relative to the query's own location) in
  // [x lo, x hi] x [y lo, y hi] find their four corners in LDS; the symmetric "halo" h is (-h, h, -h, h)
  signed char win[kMaxM][kMaxL][4];
  // v3: levels are staged in PASSES; first[l] = first level of level l's pass (all 0: one pass)
  int first[kMaxL];
  // v3: valid pixel counts [B][L][2] (w, h) of the level masks' first row / column, fp32, or null: when given, the
  // reference points are computed in fp32 from the query's pixel centre and these (instead of being read from `ref`)
  const float* vcounts;
};

// ---- region geometry along one axis (n pixels, R regions) --------------------------------------------------
// pixel x belongs to region r iff its centre (x + 0.5) / n lies in [r / R, (r + 1) / R)
__host__ __device__ inline int q_bound(int r, int n, int R) { return (2 * r * n + R - 1) / (2 * R); }  // ceil(r n / R - 1/2)
// rows / columns a sample of a query of region r can touch when lo <= offset <= hi pixels of this level:
// floor(r n / R - 1/2 + lo) .. ceil((r + 1) n / R - 1/2 + hi), clamped to the level (and to each other)
__host__ __device__ inline int patch_lo(int r, int n, int R, int lo) {
  const int num = 2 * r * n - R + 2 * lo * R;
  const int v = num <= 0 ? 0 : num / (2 * R);
  return v > n - 1 ? n - 1 : v;
}
__host__ __device__ inline int patch_hi(int r, int n, int R, int hi) {
  const int num = 2 * (r + 1) * n - R + 2 * hi * R;
  const int v = num <= 0 ? 0 : (num + 2 * R - 1) / (2 * R);
  return v > n - 1 ? n - 1 : v;
}

__device__ __forceinline__ unsigned xcd_tile(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
  const unsigned first = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return first + i;
}

// quad (4-lane) data movement on the DPP path: no LDS, no extra latency
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);  // all lanes valid: no `old` to keep
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float(dpp_u<CTRL>(__float_as_uint(v)));
}
__device__ __forceinline__ unsigned quad_bcast_u(unsigned v, int owner) {
  switch (owner) {
    case 0: return dpp_u<0x00>(v);
    case 1: return dpp_u<0x55>(v);
    case 2: return dpp_u<0xAA>(v);
    default: return dpp_u<0xFF>(v);
  }
}
__device__ __forceinline__ float quad_bcast_f(float v, int owner) {
  return __uint_as_float(quad_bcast_u(__float_as_uint(v), owner));
}
constexpr int kXor1 = 0xB1, kXor2 = 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]

template <class TR, int KMAX>
struct Raw {
  using S = typename TR::storage;
  struct __attribute__((aligned(2 * sizeof(S)))) S2 { S a, b; };
  S2 o[KMAX], r[KMAX];
  S w[KMAX];
};

// P4 (num_points == 4): point sub + 4k of a quad lane lies on level k -- uniform, so every per-point table lookup
// and the level index need no division, and the loads of a lane are one base pointer + immediate offsets.
template <class TR, int KMAX, bool P4>
__device__ __forceinline__ void load_raw(Raw<TR, KMAX>& raw, const typename TR::storage* __restrict__ offs,
                                         const typename TR::storage* __restrict__ logits,
                                         const typename TR::storage* __restrict__ ref, size_t row, int m, int sub,
                                         int L, int P, int off_stride, int logit_stride) {
  using R = Raw<TR, KMAX>;
  const int LP = L * P;
  if (P4) {
    const typename TR::storage* po = offs + row * off_stride + 2 * (m * LP + sub);
    const typename TR::storage* pg = logits + row * logit_stride + (m * LP + sub);
    const typename TR::storage* pr = ref + row * L * 2;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < L) {   // (L == KMAX in the LFULL instantiation: the launcher passes L as a literal-equal value)
        raw.o[k] = *reinterpret_cast<const typename R::S2*>(po + 8 * k);
        raw.r[k] = *reinterpret_cast<const typename R::S2*>(pr + 2 * k);
        raw.w[k] = pg[4 * k];
      }
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int pt = sub + 4 * k;
    if (pt < LP) {
      const int col = m * LP + pt;
      raw.o[k] = *reinterpret_cast<const typename R::S2*>(offs + row * off_stride + 2 * col);
      raw.r[k] = *reinterpret_cast<const typename R::S2*>(ref + (row * L + pt / P) * 2);
      raw.w[k] = logits[row * logit_stride + col];
    }
  }
}

// floor(a / b) for 0 <= a < 2^22, 0 < b: one reciprocal + a fix-up instead of the ~35-instruction integer division.
// v_rcp_f32 (1 ulp) is enough: a * rcp(b) is then within 0.75 of a / b, the truncated quotient within one of the true
// one, and the two corrections below repair that (the correctly rounded __frcp_rn costs ~10 instructions more per call,
// a dozen calls per workgroup)
__device__ __forceinline__ int fdiv(int a, int b) {
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
  const int r = a - q * b;
  q += r >= b ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
}
__device__ __forceinline__ int q_bound_d(int r, int n, int R) { return fdiv(2 * r * n + R - 1, 2 * R); }
__device__ __forceinline__ int patch_lo_d(int r, int n, int R, int lo) {
  const int num = 2 * r * n - R + 2 * lo * R;
  const int v = num <= 0 ? 0 : fdiv(num, 2 * R);
  return v > n - 1 ? n - 1 : v;
}
__device__ __forceinline__ int patch_hi_d(int r, int n, int R, int hi) {
  const int num = 2 * (r + 1) * n - R + 2 * hi * R;
  const int v = num <= 0 ? 0 : fdiv(num + 2 * R - 1, 2 * R);
  return v > n - 1 ? n - 1 : v;
}

// slot (position of a query in the region's list: level-major, then row-major inside the level's rectangle)
// -> flattened query index
__device__ __forceinline__ int slot_query(const int* __restrict__ s_meta, int L, int slot) {
  int lv = 0;
  for (int l = 1; l < L; ++l) lv = slot >= s_meta[l * kMetaInts + 12] ? l : lv;
  const int* mt = s_meta + lv * kMetaInts;
  const int t = slot - mt[12], qw = mt[10];
  const int y = (int)(((float)t + 0.5f) * __frcp_rn((float)qw));
  return mt[2] + (mt[9] + y) * mt[1] + mt[8] + (t - y * qw);
}

// One iteration of a wave: 16 queries x this head.  raw = the quad's share of the offsets / logits / reference
// points of query q; the region's neighbourhoods are in `patch`, described by s_meta.
// LFULL (P4 only): num_levels == KMAX, so the gather loops carry no run-time level test and unroll into straight-line
// code with a static register assignment (with the test the compiler kept both row buffers alive across a loop and
// shuffled them with 24 v_mov per 4 points: ~15 % of the kernel's vector instructions).
template <class TR, int KMAX, bool P4, bool LFULL>
__device__ __forceinline__ void process_queries(const Raw<TR, KMAX>& raw, const int q, const bool valid,
                                                const int* __restrict__ s_meta, const unsigned char* __restrict__ patch,
                                                const unsigned char* __restrict__ vimg, const unsigned pix_bytes,
                                                typename TR::storage* __restrict__ out, const size_t out_row /* (b*S)*M + m */,
                                                const int L, const int P, const int M, const int sub, const int ablate) {
  using V = typename TR::vec;
  constexpr unsigned kRow = 32 * sizeof(typename TR::storage);
  const int LP = L * P;
  const unsigned lane_byte = (unsigned)sub * 16;
  // -- softmax over the pair's logits (quad reductions), as build_entries does --
  float pw_[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) pw_[k] = (P4 ? (LFULL || k < L) : sub + 4 * k < LP) ? TR::to_f32(raw.w[k]) : -INFINITY;
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) mx = fmaxf(mx, pw_[k]);
  mx = fmaxf(mx, dpp_f<kXor2>(mx));
  mx = fmaxf(mx, dpp_f<kXor1>(mx));
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    pw_[k] = __expf(pw_[k] - mx);
    sum += pw_[k];
  }
  sum += dpp_f<kXor2>(sum);
  sum += dpp_f<kXor1>(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) pw_[k] *= inv;

  // -- own points -> LDS corner addresses + weights (registers) --
  unsigned ad[KMAX][4];
  float wt[KMAX][4];
  unsigned hw[KMAX];
  unsigned bad = 0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int pt = sub + 4 * k;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ad[k][c] = 0;
      wt[k][c] = 0.f;
    }
    hw[k] = 0;
    if ((P4 ? (LFULL || k < L) : pt < LP) && !(ablate & 16)) {
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      const int* mt = s_meta + (P4 ? k : pt / P) * kMetaInts;
      const i32x4 mA = *reinterpret_cast<const i32x4*>(mt);       // H, W, start, px0
      const i32x4 mB = *reinterpret_cast<const i32x4*>(mt + 4);   // py0, pw, ph, base
      const i32x4 mD = *reinterpret_cast<const i32x4*>(mt + 12);  // slot0, px1, py1, -
      const i32x4 mE = *reinterpret_cast<const i32x4*>(mt + 16);  // 1/W, 1/H (float bits), -, -
      const int H = mA[0], W = mA[1];
      const float Hf = (float)H, Wf = (float)W;
      const float x = fmaf(TR::to_f32(raw.o[k].a), __int_as_float(mE[0]), TR::to_f32(raw.r[k].a));
      const float y = fmaf(TR::to_f32(raw.o[k].b), __int_as_float(mE[1]), TR::to_f32(raw.r[k].b));
      const float aw = pw_[k];
      const float h_im = fmaf(y, Hf, -0.5f);
      const float w_im = fmaf(x, Wf, -0.5f);
      const bool gate = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;  // cu:249
      const float hf = floorf(h_im), wf = floorf(w_im);
      const float lh = h_im - hf, lw = w_im - wf;
      // out-of-gate samples carry zero weights; keep their integer coordinates tame
      const int h0 = gate ? (int)hf : 0, w0 = gate ? (int)wf : 0;
      // corners outside the image (cu:52-71) and gated samples: zero the factor once instead of each product
      const float g_aw = gate ? aw : 0.f;
      const float hh = h0 >= 0 ? 1.f - lh : 0.f, lhe = h0 + 1 <= H - 1 ? lh : 0.f;
      const float hwt = w0 >= 0 ? 1.f - lw : 0.f, lwe = w0 + 1 <= W - 1 ? lw : 0.f;
      wt[k][0] = hh * hwt * g_aw;
      wt[k][1] = hh * lwe * g_aw;
      wt[k][2] = lhe * hwt * g_aw;
      wt[k][3] = lhe * lwe * g_aw;
      const int px0 = mA[3], py0 = mB[0], pwid = mB[1], px1 = mD[1], py1 = mD[2];
      // corners that carry weight lie inside the image; are they inside the staged neighbourhood too?
      const bool inside = max(w0, 0) >= px0 && min(w0 + 1, W - 1) <= px1 && max(h0, 0) >= py0 && min(h0 + 1, H - 1) <= py1;
      if (gate && !inside) bad |= 1u << pt;
      hw[k] = ((unsigned)h0 << 16) | ((unsigned)w0 & 0xffffu);
      // neighbourhood-relative, clamped: always a valid LDS row (the neighbourhood lies inside the image)
      const int ry0 = min(max(h0, py0), py1) - py0, ry1 = min(max(h0 + 1, py0), py1) - py0;
      const int rx0 = min(max(w0, px0), px1) - px0, rx1 = min(max(w0 + 1, px0), px1) - px0;
      const int r0 = ry0 * pwid + mB[3], r1 = ry1 * pwid + mB[3];
      ad[k][0] = (unsigned)(r0 + rx0) * kRow;
      ad[k][1] = (unsigned)(r0 + rx1) * kRow;
      ad[k][2] = (unsigned)(r1 + rx0) * kRow;
      ad[k][3] = (unsigned)(r1 + rx1) * kRow;
    }
  }
  bad |= dpp_u<kXor2>(bad);
  bad |= dpp_u<kXor1>(bad);

  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;

  const bool wave_clean = __builtin_amdgcn_ballot_w64(bad != 0) == 0;  // every sample of the 16 pairs is in LDS
  if (ablate & 2) {
  } else if (P4) {
    // L*P = 4 L: two points per step, the rows of step s + 1 requested before the FMAs of step s.  CHECK: a point
    // that some pair of the wave has outside its staged neighbourhood is re-read from global memory by those lanes.
    auto run = [&](auto check_c) {
      constexpr bool CHECK = decltype(check_c)::value;
      constexpr int NS = 2 * KMAX;  // steps
      V rows[2][2][4];
      float ww[2][2][4];
      auto fetch = [&](int s_, int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int pt = 2 * s_ + u, o = pt & 3, k = pt >> 2;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const unsigned a = quad_bcast_u(ad[k][c], o) + lane_byte;
            ww[buf][u][c] = quad_bcast_f(wt[k][c], o);
            rows[buf][u][c] = *reinterpret_cast<const V*>(patch + a);
          }
          if (CHECK) {
            const bool mine = (bad >> pt) & 1u;
            if (__builtin_amdgcn_ballot_w64(mine) != 0) {
              const unsigned hwb = quad_bcast_u(hw[k], o);
              if (mine) {
                const int* mt = s_meta + k * kMetaInts;
                const int H = mt[0], W = mt[1];
                const int h0 = (int)(short)(hwb >> 16), w0 = (int)(short)(hwb & 0xffffu);
                const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h0 + 1, 0), H - 1);
                const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w0 + 1, 0), W - 1);
                const unsigned st = (unsigned)mt[2];
                const unsigned char* vb = vimg + lane_byte;
                rows[buf][u][0] = *reinterpret_cast<const V*>(vb + (size_t)((st + (unsigned)(h0c * W + w0c)) * pix_bytes));
                rows[buf][u][1] = *reinterpret_cast<const V*>(vb + (size_t)((st + (unsigned)(h0c * W + w1c)) * pix_bytes));
                rows[buf][u][2] = *reinterpret_cast<const V*>(vb + (size_t)((st + (unsigned)(h1c * W + w0c)) * pix_bytes));
                rows[buf][u][3] = *reinterpret_cast<const V*>(vb + (size_t)((st + (unsigned)(h1c * W + w1c)) * pix_bytes));
              }
            }
          }
        }
      };
      fetch(0, 0);
      // the checked variant keeps its run-time level test on purpose: straight-line, its conditional global re-reads
      // make the compiler sink all 640 FMAs below all 20 fetches (320 row registers -> spills)
      constexpr bool STATIC = LFULL && !CHECK;
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        if (STATIC || (s_ >> 1) < L) {
          if (s_ + 1 < NS && (STATIC || ((s_ + 1) >> 1) < L)) fetch(s_ + 1, (s_ + 1) & 1);
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_fmaf(ww[s_ & 1][u][c], TR::to_f32(rows[s_ & 1][u][c][j]), acc[j]);
          // straight-line code (LFULL): keep the written order -- rows of step s + 1 requested, then the FMAs of step
          // s -- instead of letting the scheduler hoist every later step's reads (register spills)
          if (STATIC) __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if (wave_clean)
      run(std::false_type{});
    else
      run(std::true_type{});
  } else if (wave_clean) {
#pragma unroll
    for (int p0 = 0; p0 < 4 * KMAX; p0 += 4) {
      if (p0 < LP) {
        V rows[4][4];
        float ww[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (p0 + u < LP) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const unsigned a = quad_bcast_u(ad[p0 >> 2][c], u) + lane_byte;
              ww[u][c] = quad_bcast_f(wt[p0 >> 2][c], u);
              rows[u][c] = *reinterpret_cast<const V*>(patch + a);
            }
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (p0 + u < LP) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(ww[u][c], TR::to_f32(rows[u][c][j]), acc[j]);
          }
      }
    }
  } else {
    // ---- checked loop (any L, P): samples outside the staged neighbourhood come from global memory ----
#pragma unroll
    for (int p0 = 0; p0 < 4 * KMAX; p0 += 4) {
      if (p0 < LP) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pt = p0 + u;
          if (pt >= LP) continue;
          V rows[4];
          float ww[4];
          const unsigned hwb = quad_bcast_u(hw[p0 >> 2], u);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const unsigned a = quad_bcast_u(ad[p0 >> 2][c], u) + lane_byte;
            ww[c] = quad_bcast_f(wt[p0 >> 2][c], u);
            rows[c] = *reinterpret_cast<const V*>(patch + a);
          }
          if ((bad >> pt) & 1u) {
            const int* mt = s_meta + (pt / P) * kMetaInts;
            const int H = mt[0], W = mt[1];
            const int h0 = (int)(short)(hwb >> 16), w0 = (int)(short)(hwb & 0xffffu);
            const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h0 + 1, 0), H - 1);
            const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w0 + 1, 0), W - 1);
            const unsigned st = (unsigned)mt[2];
            rows[0] = *reinterpret_cast<const V*>(vimg + (size_t)((st + (unsigned)(h0c * W + w0c)) * pix_bytes + lane_byte));
            rows[1] = *reinterpret_cast<const V*>(vimg + (size_t)((st + (unsigned)(h0c * W + w1c)) * pix_bytes + lane_byte));
            rows[2] = *reinterpret_cast<const V*>(vimg + (size_t)((st + (unsigned)(h1c * W + w0c)) * pix_bytes + lane_byte));
            rows[3] = *reinterpret_cast<const V*>(vimg + (size_t)((st + (unsigned)(h1c * W + w1c)) * pix_bytes + lane_byte));
          }
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(ww[c], TR::to_f32(rows[c][j]), acc[j]);
        }
      }
    }
  }

  if (valid && !(ablate & 8)) {
    V packed;
#pragma unroll
    for (int j = 0; j < 8; ++j) packed[j] = TR::from_f32(acc[j]);
    *reinterpret_cast<V*>(reinterpret_cast<unsigned char*>(out) + (out_row + (size_t)q * M) * kRow + lane_byte) =
        packed;
  }
}

struct TileId {
  unsigned b;
  int rx, ry, m;
};
// tile -> (image, region row / column, head); tile < 2^22 (host-checked), so the cheap division applies
// Regions are walked in horizontal BANDS of kBand region rows, column by column inside a band: vertically adjacent
// regions (whose neighbourhoods overlap by 2 * halo + 1 of ~2 * halo + 9 rows) run back to back on one XCD, so the
// overlap is an L2 hit instead of a second trip over the fabric -- a raster walk re-fetched every value row about twice
// (FETCH_SIZE 1.47 x the algorithmic bytes); 4 region rows x the 8 heads of a column = 2.3 MB of neighbourhoods in
// flight per XCD, inside its 4 MB L2.
constexpr int kBand = 4;   // tiles of a band walk 4 region columns at a time (1 = raster walk measured slower; results do not depend on it)
__device__ __forceinline__ TileId decode_tile(unsigned tile, const EncGeom& g) {
  TileId t;
  const int unit = fdiv((int)tile, g.M);
  t.m = (int)tile - unit * g.M;
  const int regions = g.RX * g.RY;
  const int b = fdiv(unit, regions);
  const int reg = unit - b * regions;
  t.b = (unsigned)b;
  const int kBand = g.band;
  const int per_band = kBand * g.RX;
  const int band = fdiv(reg, per_band);
  const int r = reg - band * per_band;
  const int y0 = band * kBand;
  const int bh = min(kBand, g.RY - y0);   // the last band may be shorter
  t.rx = fdiv(r, bh);
  t.ry = y0 + (r - t.rx * bh);
  return t;
}

// region geometry -> the LDS table: thread l < L takes level l, then the running sums over the levels before it
// (two barriers inside; every thread of the workgroup must call it)
__device__ __forceinline__ void region_geometry(int* __restrict__ s_meta, const EncGeom& g, const TileId t, int tid) {
  if (tid < g.L) {
    const int W = g.W[tid], H = g.H[tid];
    const signed char* wn = g.win[t.m < kMaxM ? t.m : kMaxM - 1][tid];
    const int px0 = patch_lo_d(t.rx, W, g.RX, wn[0]), px1 = max(px0, patch_hi_d(t.rx, W, g.RX, wn[1]));
    const int py0 = patch_lo_d(t.ry, H, g.RY, wn[2]), py1 = max(py0, patch_hi_d(t.ry, H, g.RY, wn[3]));
    const int qx0 = q_bound_d(t.rx, W, g.RX), qy0 = q_bound_d(t.ry, H, g.RY);
    int* mt = s_meta + tid * kMetaInts;
    mt[0] = H;
    mt[1] = W;
    mt[2] = g.start[tid];
    mt[3] = px0;
    mt[4] = py0;
    mt[5] = px1 - px0 + 1;
    mt[6] = py1 - py0 + 1;
    mt[8] = qx0;
    mt[9] = qy0;
    mt[10] = q_bound_d(t.rx + 1, W, g.RX) - qx0;
    mt[11] = q_bound_d(t.ry + 1, H, g.RY) - qy0;
    mt[13] = px1;
    mt[14] = py1;
    mt[16] = __float_as_int(g.invW[tid]);
    mt[17] = __float_as_int(g.invH[tid]);
    // v2: floor coordinates (x0, y0) whose four corners are staged (or carry no weight: outside the image)
    mt[18] = px0 == 0 ? -1 : px0;
    mt[19] = px1 == W - 1 ? W - 1 : px1 - 1;
    mt[20] = py0 == 0 ? -1 : py0;
    mt[21] = py1 == H - 1 ? H - 1 : py1 - 1;
    if (g.vcounts) {  // get_valid_ratio (reference transformer.py:384-400) in fp32
      const float* vc = g.vcounts + ((size_t)t.b * g.L + tid) * 2;
      mt[24] = __float_as_int(vc[0] / (float)W);
      mt[25] = __float_as_int(vc[1] / (float)H);
    }
  }
  __syncthreads();
  if (tid < g.L) {
    int base = 0, slot0 = 0;
    for (int l = 0; l < tid; ++l) {
      if (l >= g.first[tid]) base += s_meta[l * kMetaInts + 5] * s_meta[l * kMetaInts + 6];
      slot0 += s_meta[l * kMetaInts + 10] * s_meta[l * kMetaInts + 11];
    }
    s_meta[tid * kMetaInts + 7] = base;
    s_meta[tid * kMetaInts + 12] = slot0;
    {  // v2: LDS byte address of staged pixel (x, y) = y * [22] + x * 64 + [23]
      const int* mt = s_meta + tid * kMetaInts;
      s_meta[tid * kMetaInts + 22] = mt[5] * 64;
      s_meta[tid * kMetaInts + 23] = (base - mt[4] * mt[5] - mt[3]) * 64;
    }
    if (tid == g.L - 1)  // queries in the region
      s_meta[kMaxL * kMetaInts] = slot0 + s_meta[tid * kMetaInts + 10] * s_meta[tid * kMetaInts + 11];
  }
  __syncthreads();
}

constexpr int kMetaStride = kMaxL * kMetaInts + 4;  // ints of the geometry table (+ the region's query count)
constexpr int kAhead = 2;                           // iterations of a wave whose raw operands are in flight

// One workgroup = one (image, region, head) tile:
//   geometry -> LDS | barrier | raw operands of the wave's first iterations requested | LDS-DMA of the neighbourhoods |
//   barrier | the wave's iterations (operands of iteration it + kAhead requested before iteration it is consumed).
// Two workgroups share a CU (73 KB of LDS each at the model shape): one gathers while the other waits for its DMA.
template <class TR, int KMAX, bool P4, bool LFULL = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void msda_encoder_kernel(
    const typename TR::storage* __restrict__ value, const typename TR::storage* __restrict__ offs,
    const typename TR::storage* __restrict__ logits, const typename TR::storage* __restrict__ ref,
    typename TR::storage* __restrict__ out, const EncGeom g, const int off_stride, const int logit_stride,
    const int ablate) {
  using S = typename TR::storage;
  constexpr unsigned kRow = 32 * sizeof(S);  // 64 B: one pixel of one head
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* patch = smem + kPad;
  int* s_meta = reinterpret_cast<int*>(smem + 2 * kPad + (size_t)g.rows_cap * kRow);

  const int tid = threadIdx.x;
  const int L = g.L, P = g.P, M = g.M;
  const int wave = tid >> 6, lane = tid & 63, sub = lane & 3, pl = lane >> 2;
  const unsigned pix_bytes = (unsigned)M * kRow;  // one pixel, all heads

  // the M heads of a region are consecutive tiles -> concurrent workgroups of one XCD
  const TileId t = decode_tile(xcd_tile(blockIdx.x, gridDim.x), g);
  region_geometry(s_meta, g, t, tid);
  const int total = s_meta[kMaxL * kMetaInts];
  const int n_it = total > wave * 16 ? (total - wave * 16 + 63) >> 6 : 0;

  const size_t row0 = (size_t)t.b * g.S;
  Raw<TR, KMAX> raws[kAhead];
  int qs[kAhead];
#pragma unroll
  for (int a = 0; a < kAhead; ++a) {
    qs[a] = 0;
    if (a < n_it && !((ablate & 4) && a > 0)) {
      const int sl = (a * 4 + wave) * 16 + pl;
      qs[a] = slot_query(s_meta, L, sl < total ? sl : total - 1);
      load_raw<TR, KMAX, P4>(raws[a], offs, logits, ref, row0 + qs[a], t.m, sub, L, P, off_stride, logit_stride);
    }
  }

  // ---- the region's neighbourhood of every level -> LDS by LDS-DMA (16 B per lane, a wave fills 1 KB) ----
  const unsigned char* vimg = reinterpret_cast<const unsigned char*>(value) + (size_t)t.b * g.S * M * kRow + t.m * kRow;
  for (int l = 0; l < ((ablate & 1) ? 0 : L); ++l) {
    const int* mt = s_meta + l * kMetaInts;
    const int W = mt[1], pw = mt[5];
    const int n = pw * mt[6] * 4;  // 16-byte pieces
    const float inv = __frcp_rn((float)pw);
    const unsigned src0 = (unsigned)(mt[2] + mt[4] * W + mt[3]) * pix_bytes + (unsigned)(lane & 3) * 16;
    unsigned char* dst0 = patch + (size_t)mt[7] * kRow;
    for (int e0 = wave * 64; e0 < n; e0 += kThreads) {
      const int row = (e0 + lane) >> 2;
      if (e0 + lane < n) {
        const int y = (int)(((float)row + 0.5f) * inv);
        const unsigned char* gp = vimg + (src0 + (unsigned)(y * (W - pw) + row) * pix_bytes);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)(dst0 + (size_t)e0 * 16), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const size_t out_row = row0 * M + t.m;
  for (int it = 0; it < n_it; ++it) {
    const bool valid = (it * 4 + wave) * 16 + pl < total;
    const Raw<TR, KMAX> raw = raws[0];
    const int q = qs[0];
#pragma unroll
    for (int a = 0; a + 1 < kAhead; ++a) {
      raws[a] = raws[a + 1];
      qs[a] = qs[a + 1];
    }
    if (it + kAhead < n_it && !(ablate & 4)) {
      const int sl = ((it + kAhead) * 4 + wave) * 16 + pl;
      qs[kAhead - 1] = slot_query(s_meta, L, sl < total ? sl : total - 1);
      load_raw<TR, KMAX, P4>(raws[kAhead - 1], offs, logits, ref, row0 + qs[kAhead - 1], t.m, sub, L, P, off_stride,
                             logit_stride);
    }
    process_queries<TR, KMAX, P4, LFULL>(raw, q, valid, s_meta, patch, vimg, pix_bytes, out, out_row, L, P, M, sub, ablate);
  }
}

// =====================================================================================================================
// Packed-half gather (fp16, num_levels == 5, num_points == 4: the model's shape) -- round 3.  Its single-pass kernel ("v2")
// was deleted in round 5 (the round-5 kernel of msda_encoder4.hip serves every 5 x 4 pyramid); the three-pass form below ("v3")
// stays as the A/B baseline behind hip_ops.MSDA_V4 = False.  What the packed form does:
//   * the blend runs on packed halves: per step (2 points = 8 corner rows) a lane multiplies its 8 channels of each
//     row by the corner's weight with v_pk_fma_f16 (2 MACs per instruction; the fp32 v_fma_mix_f32 blend issues at
//     the same ~4.4 cycles per instruction per SIMD for ONE MAC: tools/micro/valu_rates.hip) into an 8-term fp16
//     chain, and adds the chain to its fp32 accumulators -- fp16 weights (bilinear x attention, <= 1), fp32 across
//     the 10 chains of a (query, head) pair.  The reference's own half instantiation accumulates all 80 terms in
//     half (ms_deform_attn.cu:250-252 with scalar_t = c10::Half); error against the fp64 oracle: tests/test_msda_
//     encoder_gpu.py (relative L2 ~5e-4, the final fp16 rounding alone is 2e-4);
//   * corner weights travel as packed half pairs and the two row addresses as such (x1 = x0 + 1 is the next 64-byte
//     slot: an immediate offset): 4 quad broadcasts per point instead of 8;
//   * a sample that leaves the staged window no longer drags its wave through a checked loop: its weights are zeroed
//     for the gather loop, the owner lane appends a 16-byte record (floor coordinates, level, weights) to the pair's
//     queue in LDS, and the four lanes of the pair add it from global memory afterwards -- the loads of the first
//     kPre records are in flight while the gather loop runs.  Pairs with more than kQueue such samples take further
//     passes over the queue, so the result never depends on the windows, only the speed does;
//   * the window is per (head, level) (EncGeom::win): a head whose samples lean one way (the reference's
//     initialisation points head m along angle 2 pi m / M) stages that side.
// =====================================================================================================================
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ h2 as_h2(unsigned u) { return __builtin_bit_cast(h2, u); }
__device__ __forceinline__ unsigned pack_h2(float a, float b) {
  const h2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
// quad broadcast of `v` from lane `owner` of the quad, plus this lane's `add`: ONE v_add_u32_dpp (the compiler splits the
// builtin form into v_mov_b32_dpp + v_add_u32)
__device__ __forceinline__ unsigned quad_bcast_add(unsigned v, int owner, unsigned add) {
  unsigned d;
  switch (owner) {
    case 0: asm("v_add_u32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "v"(add)); break;
    case 1: asm("v_add_u32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "v"(add)); break;
    case 2: asm("v_add_u32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "v"(add)); break;
    default: asm("v_add_u32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "v"(add)); break;
  }
  return d;
}
// acc += (float)h.lo / (float)h.hi in one instruction each (v_cvt_f32_f16 + v_add_f32 otherwise)
__device__ __forceinline__ void acc_h2(float& lo, float& hi, h2 h) {
  const unsigned u = __builtin_bit_cast(unsigned, h);
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(u));
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi) : "v"(u));
}
__device__ __forceinline__ int med3_i(int a, int b, int c) {
  int d;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// =====================================================================================================================
// v3 (fp16, 5 levels x 4 points): v2's packed-half gather, restructured for OCCUPANCY.
// Measured on v2 (profiles/r03_msda_encoder_ablation.txt): the gather loop itself is 18 % of the kernel; sample
// preparation 34 %, per-tile fixed work 27 %, staging 20 % -- and a wave issues at most one vector instruction per
// ~5 cycles (~9 for the packed-math / DPP forms) however idle its SIMD is (tools/micro/valu_rates.hip), so with two
// waves per SIMD (73 KB of LDS per workgroup) the vector pipes idle half the time.  Here a workgroup walks its tile in
// three PASSES over the levels -- {0}, {1, 2}, {3, 4} -- staging only the pass's neighbourhoods: <= 40 KB of LDS and
// <= 128 registers, FOUR workgroups per CU, and the staged windows grow to +-5-6 pixels per level (a sample outside
// them is rare: the fix-up queue is the cold path).  A wave keeps the fp32 accumulators and the softmax weights of its
// (up to kMaxIt) iterations in registers across the passes.
// =====================================================================================================================
constexpr int kMaxIt = 3;    // iterations (16 queries each) per wave: regions of up to 192 queries
constexpr int kQ3 = 2;       // fix-up records per (query, head) pair and queue pass

template <class TR, int LV0, int NLV, bool CREF>
__device__ __forceinline__ void pass_v3(float (&acc)[8], const float (&aw)[5], const typename Raw<TR, 5>::S2 (&ro)[NLV],
                                        const typename Raw<TR, 5>::S2 (&rr)[NLV], const float bx, const float by,
                                        const int* __restrict__ s_meta,
                                        const unsigned char* __restrict__ patch, u32x4* __restrict__ queue,
                                        const unsigned char* __restrict__ vimg, const unsigned pix_bytes,
                                        const EncGeom& g, const int sub) {
  using V = typename TR::vec;
  using LV = const __attribute__((address_space(3))) V*;
  constexpr unsigned kRow = 64;
  // the lane's 16-byte slice of a row AND the LDS address of `patch`: the quad broadcast + add then yields the final LDS
  // address (patch + a as pointer arithmetic costs one more v_add_u32 per corner row pair: the array's LDS base is a
  // link-time constant the compiler does not fold into the ds_read offset)
  const unsigned lane_byte = (unsigned)sub * 16;
  const unsigned lds_lane = lane_byte + (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)patch;
  unsigned ad0[NLV], ad1[NLV], mw01[NLV], mw23[NLV], rw01[NLV], rw23[NLV], hw[NLV];
  unsigned bad = 0;
#pragma unroll
  for (int i = 0; i < NLV; ++i) {
    constexpr int kDummy = 0;
    (void)kDummy;
    const int k = LV0 + i;
    const int* mt = s_meta + k * kMetaInts;
    const i32x4_t mA = *reinterpret_cast<const i32x4_t*>(mt + 16);  // 1/W, 1/H (float bits), x lo, x hi
    const i32x4_t mB = *reinterpret_cast<const i32x4_t*>(mt + 20);  // y lo, y hi, row pitch, address constant
    const int H = g.H[k], W = g.W[k];
    const float Hf = (float)H, Wf = (float)W;
    // reference point on level k: read (model dtype), or the query's pixel centre scaled by the valid ratios in fp32
    const float rx = CREF ? bx * __int_as_float(mt[24]) : TR::to_f32(rr[i].a);
    const float ry = CREF ? by * __int_as_float(mt[25]) : TR::to_f32(rr[i].b);
    const float x = fmaf(TR::to_f32(ro[i].a), __int_as_float(mA[0]), rx);
    const float y = fmaf(TR::to_f32(ro[i].b), __int_as_float(mA[1]), ry);
    const float h_im = fmaf(y, Hf, -0.5f);
    const float w_im = fmaf(x, Wf, -0.5f);
    const bool gate = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;  // cu:249
    const float hf = floorf(h_im), wf = floorf(w_im);
    const float lh = h_im - hf, lw = w_im - wf;
    const int h0 = gate ? (int)hf : 0, w0 = gate ? (int)wf : 0;
    const float a = gate ? aw[k] : 0.f;
    const float wy0 = hf >= 0.f ? (1.f - lh) * a : 0.f, wy1 = hf + 1.f < Hf ? lh * a : 0.f;   // cu:52-71
    const float wx0 = wf >= 0.f ? 1.f - lw : 0.f, wx1 = wf + 1.f < Wf ? lw : 0.f;
    rw01[i] = pack_h2(wy0 * wx0, wy0 * wx1);
    rw23[i] = pack_h2(wy1 * wx0, wy1 * wx1);
    const int tx = med3_i(w0, mA[2], mA[3]), ty = med3_i(h0, mB[0], mB[1]);
    const bool isbad = gate && !(tx == w0 && ty == h0);
    mw01[i] = isbad ? 0u : rw01[i];
    mw23[i] = isbad ? 0u : rw23[i];
    hw[i] = ((unsigned)h0 << 16) | ((unsigned)w0 & 0xffffu);
    const int xo = tx * (int)kRow + mB[3];
    // (24-bit multiplies: a row index < 2^15 times a row pitch of a few KB -- v_mad_i32_i24 instead of the 64-bit mad)
    ad0[i] = (unsigned)(__mul24(max(ty, 0), mB[2]) + xo);
    ad1[i] = (unsigned)(__mul24(min(ty + 1, H - 1), mB[2]) + xo);
    bad |= isbad ? 1u << (sub + 4 * i) : 0u;
  }
  bad |= dpp_u<kXor2>(bad);
  bad |= dpp_u<kXor1>(bad);

  // -- gather: 4 * NLV points, one per step; an 8-term fp16 chain per point pair --
  {
    constexpr int NS = 4 * NLV;
    V rows[2][4];
    unsigned wA[2], wB[2];
    auto fetch = [&](int s_, int buf) {
      const int o = s_ & 3, i = s_ >> 2;
      const unsigned a0 = quad_bcast_add(ad0[i], o, lds_lane), a1 = quad_bcast_add(ad1[i], o, lds_lane);
      wA[buf] = quad_bcast_u(mw01[i], o);
      wB[buf] = quad_bcast_u(mw23[i], o);
      rows[buf][0] = *(LV)(uintptr_t)a0;
      rows[buf][1] = *(LV)(uintptr_t)(a0 + kRow);
      rows[buf][2] = *(LV)(uintptr_t)a1;
      rows[buf][3] = *(LV)(uintptr_t)(a1 + kRow);
    };
    fetch(0, 0);
    h2 h[4];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const int b = s_ & 1;
      if (s_ + 1 < NS) fetch(s_ + 1, (s_ + 1) & 1);
      const h2 a = as_h2(wA[b]), c = as_h2(wB[b]);
      const h2 w4[4] = {h2{a[0], a[0]}, h2{a[1], a[1]}, h2{c[0], c[0]}, h2{c[1], c[1]}};
#pragma unroll
      for (int cr = 0; cr < 4; ++cr) {
        const V r = rows[b][cr];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const h2 v = {r[2 * j], r[2 * j + 1]};
          h[j] = ((s_ & 1) == 0 && cr == 0) ? v * w4[cr] : __builtin_elementwise_fma(v, w4[cr], h[j]);
        }
      }
      if (s_ & 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc_h2(acc[2 * j], acc[2 * j + 1], h[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // -- samples outside the staged windows (rare): queue passes of kQ3 records per pair, read from global memory --
  if (__builtin_amdgcn_ballot_w64(bad != 0) != 0) {
    const int cnt = __builtin_popcount(bad);
    for (int base = 0; __builtin_amdgcn_ballot_w64(cnt > base) != 0; base += kQ3) {
#pragma unroll
      for (int i = 0; i < NLV; ++i) {
        const int pt = sub + 4 * i;
        if ((bad >> pt) & 1u) {
          const int pos = __builtin_popcount(bad & ((1u << pt) - 1u)) - base;
          if ((unsigned)pos < (unsigned)kQ3) queue[pos] = u32x4{hw[i], (unsigned)(LV0 + i), rw01[i], rw23[i]};
        }
      }
      for (int j = 0; j < kQ3; ++j) {
        const bool act = base + j < cnt;
        if (__builtin_amdgcn_ballot_w64(act) == 0) break;
        if (act) {
          const u32x4 rec = queue[j];
          const int* mt = s_meta + (int)rec[1] * kMetaInts;
          const int H = mt[0], W = mt[1];
          const unsigned st = (unsigned)mt[2];
          const int h0 = (int)(short)(rec[0] >> 16), w0 = (int)(short)(rec[0] & 0xffffu);
          const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h0 + 1, 0), H - 1);
          const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w0 + 1, 0), W - 1);
          const unsigned char* vb = vimg + lane_byte;
          V r4[4];
          r4[0] = *reinterpret_cast<const V*>(vb + (size_t)((st + (unsigned)(h0c * W + w0c)) * pix_bytes));
          r4[1] = *reinterpret_cast<const V*>(vb + (size_t)((st + (unsigned)(h0c * W + w1c)) * pix_bytes));
          r4[2] = *reinterpret_cast<const V*>(vb + (size_t)((st + (unsigned)(h1c * W + w0c)) * pix_bytes));
          r4[3] = *reinterpret_cast<const V*>(vb + (size_t)((st + (unsigned)(h1c * W + w1c)) * pix_bytes));
          const h2 a = as_h2(rec[2]), b = as_h2(rec[3]);
          const float w[4] = {(float)a[0], (float)a[1], (float)b[0], (float)b[1]};
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) acc[jj] = __builtin_fmaf(w[c], TR::to_f32(r4[c][jj]), acc[jj]);
        }
      }
    }
  }
}

// slot -> flattened query index, and the query's level / pixel
__device__ __forceinline__ int slot_query3(const int* __restrict__ s_meta, int L, int slot, int& lv, int& x, int& y) {
  lv = 0;
  for (int l = 1; l < L; ++l) lv = slot >= s_meta[l * kMetaInts + 12] ? l : lv;
  const int* mt = s_meta + lv * kMetaInts;
  const int t = slot - mt[12], qw = mt[10];
  const int yy = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)qw));
  y = mt[9] + yy;
  x = mt[8] + (t - yy * qw);
  return mt[2] + y * mt[1] + x;
}

// LDS: [pad 64 B | staged rows of ONE pass | pad 64 B | fix-up queues 2 KB | geometry table]
template <class TR, bool CREF>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void msda_encoder_v3_kernel(
    const typename TR::storage* __restrict__ value, const typename TR::storage* __restrict__ offs,
    const typename TR::storage* __restrict__ logits, const typename TR::storage* __restrict__ ref,
    typename TR::storage* __restrict__ out, const EncGeom g, const int off_stride, const int logit_stride) {
  using S = typename TR::storage;
  using S2 = typename Raw<TR, 5>::S2;
  using V = typename TR::vec;
  constexpr unsigned kRow = 64;
  constexpr int L = 5, LP = 20;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* patch = smem + kPad;
  u32x4* queues = reinterpret_cast<u32x4*>(smem + 2 * kPad + (size_t)g.rows_cap * kRow);
  int* s_meta = reinterpret_cast<int*>(queues + (kThreads / 4) * kQ3);

  const int tid = threadIdx.x;
  const int M = g.M;
  const int wave = tid >> 6, lane = tid & 63, sub = lane & 3, pl = lane >> 2;
  const unsigned pix_bytes = (unsigned)M * kRow;

  const TileId t = decode_tile(xcd_tile(blockIdx.x, gridDim.x), g);
  region_geometry(s_meta, g, t, tid);
  const int total = s_meta[kMaxL * kMetaInts];
  const int n_it = total > wave * 16 ? (total - wave * 16 + 63) >> 6 : 0;   // <= kMaxIt (host-checked)

  // per-image bases (uniform) + 32-bit in-image element offsets
  const S* offs_b = offs + (size_t)t.b * g.S * off_stride + 2 * (t.m * LP + sub);
  const S* logit_b = logits + (size_t)t.b * g.S * logit_stride + (t.m * LP + sub);
  const S* ref_b = ref + (size_t)t.b * g.S * (L * 2);
  const unsigned char* vimg = reinterpret_cast<const unsigned char*>(value) + (size_t)t.b * g.S * M * kRow + t.m * kRow;

  int qs[kMaxIt];
  float aw[kMaxIt][5], acc[kMaxIt][8], bx[kMaxIt], by[kMaxIt];
#pragma unroll
  for (int it = 0; it < kMaxIt; ++it) {
    qs[it] = 0;
    bx[it] = by[it] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[it][j] = 0.f;
    if (it < n_it) {
      const int sl = (it * 4 + wave) * 16 + pl;
      int lv, x, y;
      qs[it] = slot_query3(s_meta, L, sl < total ? sl : total - 1, lv, x, y);
      if (CREF) {  // get_reference_points (reference transformer.py:280-305): centre / (valid ratio * size), fp32
        const int* mt = s_meta + lv * kMetaInts;
        // (IEEE divisions on purpose: the v_rcp_f32 form is 9 instructions shorter per coordinate and measured 3 % SLOWER --
        // gpurun_out r04n, same box -- the allocation at the 128-register cap is that fragile)
        bx[it] = ((float)x + 0.5f) / (__int_as_float(mt[24]) * (float)mt[1]);
        by[it] = ((float)y + 0.5f) / (__int_as_float(mt[25]) * (float)mt[0]);
      }
    }
  }
  // softmax of every iteration's 20 logits (5 per lane; quad reductions): the attention weights live in registers
  {
    S lg[kMaxIt][5];
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it)
#pragma unroll
      for (int k = 0; k < 5; ++k) lg[it][k] = it < n_it ? logit_b[(unsigned)qs[it] * (unsigned)logit_stride + 4 * k] : S(0);
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
      float mx = TR::to_f32(lg[it][0]);
#pragma unroll
      for (int k = 1; k < 5; ++k) mx = fmaxf(mx, TR::to_f32(lg[it][k]));
      mx = fmaxf(mx, dpp_f<kXor2>(mx));
      mx = fmaxf(mx, dpp_f<kXor1>(mx));
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        aw[it][k] = __expf(TR::to_f32(lg[it][k]) - mx);
        sum += aw[it][k];
      }
      sum += dpp_f<kXor2>(sum);
      sum += dpp_f<kXor1>(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int k = 0; k < 5; ++k) aw[it][k] *= inv;
    }
  }

  u32x4* queue = queues + (size_t)(wave * 16 + pl) * kQ3;
  auto run_pass = [&](auto lv0_c, auto nlv_c) {
    constexpr int LV0 = decltype(lv0_c)::value, NLV = decltype(nlv_c)::value;
    // raw offsets / reference points of the pass's levels for all iterations: in flight while the rows are staged
    S2 ro[kMaxIt][NLV], rr[kMaxIt][NLV];
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it)
#pragma unroll
      for (int i = 0; i < NLV; ++i) {
        ro[it][i] = S2{S(0), S(0)};
        rr[it][i] = S2{S(0), S(0)};
        if (it < n_it) {
          ro[it][i] = *reinterpret_cast<const S2*>(offs_b + (unsigned)qs[it] * (unsigned)off_stride + 8 * (LV0 + i));
          if (!CREF) rr[it][i] = *reinterpret_cast<const S2*>(ref_b + (unsigned)qs[it] * (unsigned)(L * 2) + 2 * (LV0 + i));
        }
      }
    if (LV0 > 0) __syncthreads();   // every wave is done reading the previous pass's rows
#pragma unroll
    for (int i = 0; i < NLV; ++i) {
      const int* mt = s_meta + (LV0 + i) * kMetaInts;
      const int W = mt[1], pw = mt[5];
      const int n = pw * mt[6] * 4;  // 16-byte pieces
      const float inv = __builtin_amdgcn_rcpf((float)pw);
      const unsigned src0 = (unsigned)(mt[2] + mt[4] * W + mt[3]) * pix_bytes + (unsigned)(lane & 3) * 16;
      unsigned char* dst0 = patch + (size_t)mt[7] * kRow;
      for (int e0 = wave * 64; e0 < n; e0 += kThreads) {
        const int row = (e0 + lane) >> 2;
        if (e0 + lane < n) {
          const int y = (int)(((float)row + 0.5f) * inv);
          // (24-bit multiplies: rows and pixel indices < 2^24, see plan_encoder)
          const unsigned char* gp = vimg + (src0 + (unsigned)__mul24(__mul24(y, W - pw) + row, (int)pix_bytes));
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                           (__attribute__((address_space(3))) void*)(dst0 + (size_t)e0 * 16), 16, 0, 0);
        }
      }
    }
    if (tid < 8) {   // the slots in front of / behind the staged rows are read with zero weight: finite values
      const int* mt = s_meta + (LV0 + NLV - 1) * kMetaInts;
      const unsigned end = (unsigned)(mt[7] + mt[5] * mt[6]) * kRow;
      *reinterpret_cast<u32x4*>(tid < 4 ? smem + tid * 16 : patch + end + (tid - 4) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it)
      if (it < n_it)
        pass_v3<TR, LV0, NLV, CREF>(acc[it], aw[it], ro[it], rr[it], bx[it], by[it], s_meta, patch, queue, vimg, pix_bytes, g, sub);
  };
  run_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  run_pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
  run_pass(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});

  const size_t out_row = ((size_t)t.b * g.S) * M + t.m;
#pragma unroll
  for (int it = 0; it < kMaxIt; ++it)
    if (it < n_it && (it * 4 + wave) * 16 + pl < total) {
      V packed;
#pragma unroll
      for (int j = 0; j < 8; ++j) packed[j] = TR::from_f32(acc[it][j]);
      *reinterpret_cast<V*>(reinterpret_cast<unsigned char*>(out) + (out_row + (size_t)qs[it] * M) * kRow + sub * 16) = packed;
    }
}

// LDS bytes the kernels need for these windows (0: not representable), and the geometry for the launch
struct EncPlan {
  EncGeom g;
  size_t lds_v1, lds_v3;   // bytes per workgroup of the generic single-pass / the three-pass kernel
  int rows_cap3;                    // v3: the largest pass
  int rc;
};
constexpr int kPassFirst5[5] = {0, 1, 1, 3, 3};   // v3 (5 levels): passes {0}, {1, 2}, {3, 4}

inline EncPlan plan_encoder(const int64_t* shapes, int64_t S, int M, int L, int P, const signed char* win /* [M][L][4] */) {
  EncPlan pl{};
  EncGeom& g = pl.g;
  pl.rc = CODETR_E_BADARG;
  if (!shapes || !win || M <= 0 || L <= 0 || P <= 0) return pl;
  pl.rc = CODETR_E_UNSUPPORTED;
  if (L > kMaxL || L * P > 32) return pl;
  g.L = L;
  g.P = P;
  g.M = M;
  g.S = (int)S;
  int64_t sum = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t h = shapes[2 * l], w = shapes[2 * l + 1];
    if (h <= 0 || w <= 0 || h > 32767 || w > 32767) return pl.rc = CODETR_E_BADARG, pl;  // (h0, w0) travel as 16-bit halves
    g.H[l] = (int)h;
    g.W[l] = (int)w;
    g.start[l] = (int)sum;
    g.invH[l] = 1.0f / (float)h;
    g.invW[l] = 1.0f / (float)w;
    sum += h * w;
  }
  if (sum != S) return pl.rc = CODETR_E_BADARG, pl;
  int wmax = 0;
  for (int m = 0; m < M; ++m)
    for (int l = 0; l < L; ++l) {
      const signed char* wn = win + ((size_t)m * L + l) * 4;
      if (wn[0] > wn[1] || wn[2] > wn[3]) return pl.rc = CODETR_E_BADARG, pl;
      for (int c = 0; c < 4; ++c) {
        g.win[m < kMaxM ? m : kMaxM - 1][l][c] = wn[c];
        wmax = abs(wn[c]) > wmax ? abs(wn[c]) : wmax;
      }
      if (m >= kMaxM && memcmp(wn, win + ((size_t)(kMaxM - 1) * L + l) * 4, 4) != 0) return pl;  // heads beyond kMaxM share one window
    }
  // regions follow the finest level
  int fine = 0;
  for (int l = 1; l < L; ++l)
    if ((int64_t)g.H[l] * g.W[l] > (int64_t)g.H[fine] * g.W[fine]) fine = l;
  g.RX = (g.W[fine] + kRegW - 1) / kRegW;
  g.RY = (g.H[fine] + kRegH - 1) / kRegH;
  // the kernel's reciprocal-based floor division is exact below 2^22
  for (int l = 0; l < L; ++l) {
    const int64_t nx = 2 * (int64_t)(g.RX + 1) * g.W[l] + (2 * (int64_t)wmax + 3) * g.RX;
    const int64_t ny = 2 * (int64_t)(g.RY + 1) * g.H[l] + (2 * (int64_t)wmax + 3) * g.RY;
    if (nx >= (1 << 22) || ny >= (1 << 22)) return pl;
  }
  // LDS capacity: the largest neighbourhood / query count any (region, head) has, per level
  int rows_cap = 0, rows_cap3 = 0, slots = 0;
  for (int m = 0; m < (M < kMaxM ? M : kMaxM); ++m) {
    int rows = 0, rows_pass = 0;
    slots = 0;
    for (int l = 0; l < L; ++l) {
      const signed char* wn = g.win[m][l];
      int pw = 0, ph = 0, qw = 0, qh = 0;
      for (int r = 0; r < g.RX; ++r) {
        const int lo = patch_lo(r, g.W[l], g.RX, wn[0]), hi = patch_hi(r, g.W[l], g.RX, wn[1]);
        const int w = (hi > lo ? hi : lo) - lo + 1;
        const int q = q_bound(r + 1, g.W[l], g.RX) - q_bound(r, g.W[l], g.RX);
        pw = w > pw ? w : pw;
        qw = q > qw ? q : qw;
      }
      for (int r = 0; r < g.RY; ++r) {
        const int lo = patch_lo(r, g.H[l], g.RY, wn[2]), hi = patch_hi(r, g.H[l], g.RY, wn[3]);
        const int h = (hi > lo ? hi : lo) - lo + 1;
        const int q = q_bound(r + 1, g.H[l], g.RY) - q_bound(r, g.H[l], g.RY);
        ph = h > ph ? h : ph;
        qh = q > qh ? q : qh;
      }
      rows += pw * ph;
      slots += qw * qh;
      if (L == 5) {
        rows_pass = (kPassFirst5[l] == l ? 0 : rows_pass) + pw * ph;
        rows_cap3 = rows_pass > rows_cap3 ? rows_pass : rows_cap3;
      }
    }
    rows_cap = rows > rows_cap ? rows : rows_cap;
  }
  g.rows_cap = rows_cap;
  g.slots_cap = slots;
  pl.rows_cap3 = rows_cap3;
  pl.lds_v1 = 2 * kPad + (size_t)rows_cap * 64 + kMetaStride * sizeof(int);
  pl.lds_v3 = 2 * kPad + (size_t)rows_cap3 * 64 + kMetaStride * sizeof(int) + (size_t)(kThreads / 4) * kQ3 * 16;
  pl.rc = 0;
  return pl;
}

template <class TR>
int launch_encoder(hipStream_t st, const void* value, const int64_t* shapes, const void* offs, int64_t off_stride,
                   const void* logits, int64_t logit_stride, const void* ref, const float* vcounts, int64_t B, int64_t S,
                   int M, int D, int L, int P, const signed char* win, int passes, void* out) {
  using ST = typename TR::storage;
  if (!value || !shapes || !offs || !logits || !out || !win || (!ref && !(vcounts && passes == 3))) return CODETR_E_BADARG;
  if (passes != 1 && passes != 3) return CODETR_E_BADARG;
  if (B <= 0 || S <= 0 || M <= 0 || L <= 0 || P <= 0) return CODETR_E_BADARG;
  if (D != 32) return CODETR_E_UNSUPPORTED;
  if (off_stride < (int64_t)M * L * P * 2 || logit_stride < (int64_t)M * L * P || off_stride > 0x7fffffff ||
      logit_stride > 0x7fffffff || (off_stride & 1) || (reinterpret_cast<uintptr_t>(offs) & 3) ||
      (reinterpret_cast<uintptr_t>(ref) & 3))
    return CODETR_E_BADARG;
  EncPlan pl = plan_encoder(shapes, S, M, L, P, win);
  if (pl.rc != 0) return pl.rc;
  EncGeom& g = pl.g;
  if (S * M * (int64_t)(D * sizeof(ST)) > 0xffffffffLL) return CODETR_E_TOO_LARGE;  // 32-bit in-image offsets
  const int64_t blocks = B * g.RX * g.RY * M;
  if (blocks >= (1 << 22)) return CODETR_E_UNSUPPORTED;  // (the kernel's cheap tile decode)
  // (No run-time switches: the band width, the static level walk and the one-pass kernel's variants were A/B-measured in
  // rounds 2-3 -- profiles/r02_msda_encoder_ab.txt, r03_msda_encoder_ablation.txt -- and the losers are gone.  Timing
  // experiments build this file with -DMSDA_ENC_ABLATE, which reads its knobs through tools/micro/diag_env.h.)
#ifdef MSDA_ENC_ABLATE
  static const int ablate = diag_env_int("CODETR_MSDA_ENC_ABLATE", 0);
#else
  constexpr int ablate = 0;
#endif
  constexpr int band_env = kBand;
  constexpr bool static_env = true;
  g.band = band_env < 1 ? 1 : (band_env > 64 ? 64 : band_env);
  if (passes == 3) {   // three-pass kernel: fp16, 5 levels x 4 points, regions of <= 64 * kMaxIt queries
    if (!std::is_same<TR, F16>::value || P != 4 || L != 5 || g.slots_cap > 64 * kMaxIt || pl.lds_v3 > (size_t)kMaxLds ||
        S * (off_stride > logit_stride ? off_stride : logit_stride) > 0x7fffffffLL)
      return CODETR_E_UNSUPPORTED;
    for (int l = 0; l < L; ++l) g.first[l] = kPassFirst5[l];
    g.rows_cap = pl.rows_cap3;
    g.vcounts = vcounts;
    auto k3 = vcounts ? msda_encoder_v3_kernel<F16, true> : msda_encoder_v3_kernel<F16, false>;
    static std::atomic<uint32_t> done3[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0, done3[0].store(0);
    const uint32_t bit = vcounts ? 2u : 1u;
    if (!(done3[dev].load(std::memory_order_acquire) & bit)) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
      if (e != hipSuccess) return (int)e;
      done3[dev].fetch_or(bit, std::memory_order_release);
    }
    const size_t lds3 = pl.lds_v3;
    hipLaunchKernelGGL(k3, dim3((unsigned)blocks), dim3(kThreads), lds3, st, static_cast<const _Float16*>(value),
                       static_cast<const _Float16*>(offs), static_cast<const _Float16*>(logits),
                       static_cast<const _Float16*>(ref), static_cast<_Float16*>(out), g, (int)off_stride, (int)logit_stride);
    const hipError_t err3 = hipGetLastError();
    return err3 == hipSuccess ? 0 : (int)err3;
  }
  const size_t lds = pl.lds_v1;
  if (lds > (size_t)kMaxLds) return CODETR_E_UNSUPPORTED;
  const int kmax5 = L * P <= 20;
  auto kern = P == 4 ? (kmax5 ? (L == 5 && static_env ? msda_encoder_kernel<TR, 5, true, true> : msda_encoder_kernel<TR, 5, true>)
                              : msda_encoder_kernel<TR, 8, true>)
                     : (kmax5 ? msda_encoder_kernel<TR, 5, false> : msda_encoder_kernel<TR, 8, false>);
  const void* kfn = reinterpret_cast<const void*>(kern);
  // > 64 KB of dynamic LDS needs the attribute on the CURRENT device's function object: remembered per (device, kernel)
  // -- a process-wide "already set" flag would skip it when the process moves to a second GPU
  {
    static std::atomic<uint32_t> done[64];  // bit = kernel instantiation, index = device ordinal
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0, done[0].store(0);
    const uint32_t bit = 1u << (kmax5 + 2 * (P == 4) + 4 * (P == 4 && L == 5 && static_env) + 8 * std::is_same<TR, BF16>::value);
    if (!(done[dev].load(std::memory_order_acquire) & bit)) {
      const hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
      if (e != hipSuccess) return (int)e;
      done[dev].fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kThreads), lds, st, static_cast<const ST*>(value),
                     static_cast<const ST*>(offs), static_cast<const ST*>(logits), static_cast<const ST*>(ref),
                     static_cast<ST*>(out), g, (int)off_stride, (int)logit_stride, ablate);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

// symmetric windows of the `halo` entry points
struct HaloWin {
  signed char w[kMaxM > 64 ? kMaxM : 64][kMaxL][4];
};
inline int halo_windows(int M, int L, int halo, signed char* buf /* [M][L][4] */) {
  if (halo < 0 || halo > 127) return CODETR_E_BADARG;
  for (int i = 0; i < M * L; ++i) {
    buf[4 * i + 0] = buf[4 * i + 2] = (signed char)-halo;
    buf[4 * i + 1] = buf[4 * i + 3] = (signed char)halo;
  }
  return 0;
}

template <class TR>
int launch_encoder_halo(void* stream, const void* value, const int64_t* shapes, const void* offs, int64_t off_stride,
                        const void* logits, int64_t logit_stride, const void* ref, int64_t B, int64_t S, int M, int D,
                        int L, int P, int halo, void* out) {
  if (M <= 0 || L <= 0 || M > 64 || L > kMaxL) return M <= 0 || L <= 0 ? CODETR_E_BADARG : CODETR_E_UNSUPPORTED;
  signed char win[64 * kMaxL * 4];
  const int rc = halo_windows(M, L, halo, win);
  if (rc != 0) return rc;
  return launch_encoder<TR>(static_cast<hipStream_t>(stream), value, shapes, offs, off_stride, logits, logit_stride,
                            ref, nullptr, B, S, M, D, L, P, win, 1, out);
}

}  // namespace

extern "C" {

int codetr_msda_encoder_forward_f16(void* stream, const void* value_dev, const int64_t* level_shapes_host,
                                    const void* offsets_dev, int64_t offsets_row_stride, const void* logits_dev,
                                    int64_t logits_row_stride, const void* ref_dev, int64_t B, int64_t S, int M, int D,
                                    int L, int P, int halo, void* out_dev) {
  return launch_encoder_halo<F16>(stream, value_dev, level_shapes_host, offsets_dev, offsets_row_stride, logits_dev,
                                  logits_row_stride, ref_dev, B, S, M, D, L, P, halo, out_dev);
}

int codetr_msda_encoder_forward_bf16(void* stream, const void* value_dev, const int64_t* level_shapes_host,
                                     const void* offsets_dev, int64_t offsets_row_stride, const void* logits_dev,
                                     int64_t logits_row_stride, const void* ref_dev, int64_t B, int64_t S, int M,
                                     int D, int L, int P, int halo, void* out_dev) {
  return launch_encoder_halo<BF16>(stream, value_dev, level_shapes_host, offsets_dev, offsets_row_stride, logits_dev,
                                   logits_row_stride, ref_dev, B, S, M, D, L, P, halo, out_dev);
}

int codetr_msda_encoder_forward_win_f16(void* stream, const void* value_dev, const int64_t* level_shapes_host,
                                        const void* offsets_dev, int64_t offsets_row_stride, const void* logits_dev,
                                        int64_t logits_row_stride, const void* ref_dev, const float* valid_counts_dev,
                                        int64_t B, int64_t S, int M, int D, int L, int P, const int8_t* windows_host,
                                        int passes, void* out_dev) {
  return launch_encoder<F16>(static_cast<hipStream_t>(stream), value_dev, level_shapes_host, offsets_dev,
                             offsets_row_stride, logits_dev, logits_row_stride, ref_dev, valid_counts_dev, B, S, M, D, L,
                             P, reinterpret_cast<const signed char*>(windows_host), passes, out_dev);
}

int codetr_msda_encoder_forward_win_bf16(void* stream, const void* value_dev, const int64_t* level_shapes_host,
                                         const void* offsets_dev, int64_t offsets_row_stride, const void* logits_dev,
                                         int64_t logits_row_stride, const void* ref_dev, const float* valid_counts_dev,
                                         int64_t B, int64_t S, int M, int D, int L, int P, const int8_t* windows_host,
                                         int passes, void* out_dev) {
  return launch_encoder<BF16>(static_cast<hipStream_t>(stream), value_dev, level_shapes_host, offsets_dev,
                              offsets_row_stride, logits_dev, logits_row_stride, ref_dev, valid_counts_dev, B, S, M, D, L,
                              P, reinterpret_cast<const signed char*>(windows_host), passes, out_dev);
}

int64_t codetr_msda_encoder_lds_bytes(const int64_t* level_shapes_host, int M, int L, int P, const int8_t* windows_host,
                                      int variant) {
  int64_t S = 0;
  if (!level_shapes_host || L <= 0 || L > kMaxL) return CODETR_E_BADARG;
  for (int l = 0; l < L; ++l) S += level_shapes_host[2 * l] * level_shapes_host[2 * l + 1];
  const EncPlan pl = plan_encoder(level_shapes_host, S, M, L, P, reinterpret_cast<const signed char*>(windows_host));
  if (pl.rc != 0) return pl.rc;
  if (variant == 3) return L == 5 && P == 4 && pl.g.slots_cap <= 64 * kMaxIt ? (int64_t)pl.lds_v3 : (int64_t)CODETR_E_UNSUPPORTED;
  return (int64_t)pl.lds_v1;   // (variant 2, the round-3 single-pass packed kernel, is gone: its shapes take the generic one)
}

}  // extern "C"
