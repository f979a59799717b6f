// Encoder self-attention form of the fused multi-scale deformable attention, round-5 kernel ("v4") for MI355X (gfx950).
//
// The op: reference codetr/csrc/ms_deform_attn.cu:31-77, 211-261 plus the softmax / sampling-location steps of
// codetr/multi_scale_deformable_attention.py:180-196 and the reference points of codetr/transformer.py:280-305, for
// DetrTransformerEncoder (codetr/transformer.py:81-92: the queries ARE the pixels of the pyramid).
//
// What round 4's three-pass kernel (msda_encoder.hip, v3) spent its time on -- profiles/r03_msda_encoder_ablation.txt:
// sample preparation 34 %, per-tile skeleton 27 %, staging 20 %, the gather loop itself 18 %; ~1 300 vector instructions
// per 16 (query, head) pairs.  This kernel keeps v3's gather loop (LDS-staged per-(head, level) windows, packed-half
// blend, DPP quad hand-over, three passes over the levels) and removes the rest:
//   * LANE-MAJOR PROJECTION.  The (offsets | logits) GEMM's weight rows are permuted once on the host so that its
//     output row holds, per head, 4 x 32 bytes: the 16 halves a quad lane needs -- (x, y) of its point on the five
//     levels, its five logits, one pad.  A lane fetches its operands for an iteration with TWO 16-byte loads, a quad
//     reads one whole 128-byte line (v3: fifteen 2/4-byte loads per lane, 16 partial lines per instruction).
//   * GEOMETRY IN SCALAR REGISTERS.  Everything that depends on (image, region, head) only -- windows, LDS bases, the
//     region's query rectangles, valid pixel counts -- is wave-uniform: each wave computes it redundantly and keeps it
//     in SGPRs (no LDS table, no barriers, no per-level LDS reads inside the preparation).
//   * ZERO BORDER.  A staged window extends one pixel beyond the image where it touches the border and those cells are
//     zero-filled, so a sample needs ONE unsigned range test per axis (are its four corners staged?) and no per-corner
//     validity arithmetic: floor + fract + subtract + compare, 2 instructions for the LDS address, 9 for the weights.
//     Everything else -- samples outside the windows, outside the image, non-finite -- takes the fix-up path, which
//     re-derives the sample with the reference's full gate / corner logic and reads its rows from global memory.
//   * ROW-WISE STAGING from a head-major value map [B][M][S][32] (the value projection writes it that way): a wave takes
//     window rows, one LDS-DMA instruction moves 16 pixels of a row -- 1 KiB contiguous, 8 whole 128-byte lines -- from a
//     scalar base; 512 threads on a 16 x 16 region re-stage half as much halo per query as 256 on 16 x 8.
//   * reference points in fp32 from the valid pixel counts (as v3's CREF form): centre * vc_k / vc_q - 0.5 + offset.
// Numerics: as v3 -- fp16 corner weights (bilinear x attention), 8-term fp16 chains added into fp32 accumulators; tested
// against the fp64 oracle at the reference's half tolerance (tests/test_msda_encoder4_gpu.py).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#include "codetr_hip.h"

namespace {

constexpr int kL = 5, kP = 4;
constexpr int kMaxM = 16;     // heads with a staged window of their own
constexpr int kMaxIt = 3;     // iterations (16 queries each) per wave
#ifndef MSDA4_KQ
#define MSDA4_KQ 2
#endif
constexpr int kQ = MSDA4_KQ;  // fix-up records per (query, head) pair and queue round
constexpr int kMaxLds = 160 * 1024;
#ifndef MSDA4_BAND
#define MSDA4_BAND 4
#endif
constexpr int kBand = MSDA4_BAND;   // region rows per band of the tile walk (see decode_tile; 2 / 8 measured: profiles/r05_msda_encoder4_sweep.txt)
#ifndef MSDA4_ABL
#define MSDA4_ABL 0           // timing experiments only (tools/micro/build_variant.sh): parts compiled out, WRONG results
#endif
constexpr int kAbl = MSDA4_ABL;   // 1: no wait for the staging DMA, 2: no staging, 4: no gather, 8: no preparation, 16: no barriers, 32: no output stores

#ifdef MSDA4_STAMPS   // diagnostic build only (tools/msda4_stamps.py): per-workgroup cycle stamps of wave 0
__device__ unsigned long long* g_msda4_stamps;
#define M4_T(i) const unsigned long long m4_t##i = __builtin_readcyclecounter()
#define M4_PUT(k, v) m4_s[k] = (v)
#else
#define M4_T(i)
#define M4_PUT(k, v)
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Geom4 {
  int M, S, RX, RY;
  int H[kL], W[kL], start[kL];
  int first[kL];                       // first level of level l's pass
  signed char win[kMaxM][kL][4];       // per (head, level): offsets (pixels of that level) in [x lo, x hi] x [y lo, y hi] are staged
  const float* vcounts;                // [B][L][2] valid pixels (w, h) of every level, fp32
  unsigned queue_off;                  // LDS byte offset of the fix-up queues (behind the largest pass)
  int head_major;                      // value map [B][M][S][32] (a head's rows contiguous) instead of [B][S][M][32]
};

// ---- region geometry along one axis (n pixels, R regions): pixel x belongs to region r iff (x + 0.5) / n in [r / R, (r + 1) / R)
__host__ __device__ inline int q_bound(int r, int n, int R) { return (2 * r * n + R - 1) / (2 * R); }
// staged columns for offsets in [lo, hi], in EXTENDED coordinates (-1 and n are the zero border):
//   first = floor(r n / R - 1/2 + lo) clamped to [-1, n - 1]; last = ceil((r + 1) n / R - 1/2 + hi) clamped to [first + 1, n]
__host__ __device__ inline int ext_lo(int r, int n, int R, int lo) {
  const int num = 2 * r * n - R + 2 * lo * R;
  const int v = num < 0 ? -1 : num / (2 * R);
  return v > n - 1 ? n - 1 : v;
}
__host__ __device__ inline int ext_hi(int r, int n, int R, int hi, int first) {
  const int num = 2 * (r + 1) * n - R + 2 * hi * R;
  int v = num <= 0 ? 0 : (num + 2 * R - 1) / (2 * R);
  v = v > n ? n : v;
  return v < first + 1 ? first + 1 : v;
}

// floor(a / b) for 0 <= a < 2^22, 0 < b (one reciprocal + a fix-up; see msda_encoder.hip)
__device__ __forceinline__ int fdiv(int a, int b) {
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
  const int r = a - q * b;
  q += r >= b ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
}
__device__ __forceinline__ int q_bound_d(int r, int n, int R) { return fdiv(2 * r * n + R - 1, 2 * R); }
__device__ __forceinline__ int ext_lo_d(int r, int n, int R, int lo) {
  const int num = 2 * r * n - R + 2 * lo * R;
  const int v = num < 0 ? -1 : fdiv(num, 2 * R);
  return v > n - 1 ? n - 1 : v;
}
__device__ __forceinline__ int ext_hi_d(int r, int n, int R, int hi, int first) {
  const int num = 2 * (r + 1) * n - R + 2 * hi * R;
  int v = num <= 0 ? 0 : fdiv(num + 2 * R - 1, 2 * R);
  v = v > n ? n : v;
  return v < first + 1 ? first + 1 : v;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

__device__ __forceinline__ unsigned xcd_tile(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
  const unsigned first = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return first + i;
}

// quad (4-lane) data movement on the DPP path
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
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
constexpr int kXor1 = 0xB1, kXor2 = 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]
// quad broadcast of `v` from lane `owner` plus this lane's `add`: ONE v_add_u32_dpp
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
__device__ __forceinline__ h2 as_h2(unsigned u) { return __builtin_bit_cast(h2, u); }
__device__ __forceinline__ unsigned pack_h2(float a, float b) {   // v_cvt_pk_f16_f32 on gfx950 (round to nearest even)
  const h2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float h_lo(unsigned u) { return (float)as_h2(u)[0]; }
__device__ __forceinline__ float h_hi(unsigned u) { return (float)as_h2(u)[1]; }
// acc += (float)h.lo / (float)h.hi in one instruction each
__device__ __forceinline__ void acc_h2(float& lo, float& hi, h2 h) {
  const unsigned u = __builtin_bit_cast(unsigned, h);
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(u));
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi) : "v"(u));
}
// one LDS-DMA piece: 16 B per lane from (wave-uniform 64-bit base in SGPRs) + (per-lane 32-bit byte offset) to the
// wave-uniform LDS address `lds_addr` + lane * 16 (M0 is written in the statement that reads it)
__device__ __forceinline__ void lds_dma16(const unsigned char* src, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds_addr)
               : "memory", "m0");
}
__device__ __forceinline__ int floor_i(float v) {   // floor + float -> int in one instruction
  int d;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(d) : "v"(v));
  return d;
}

// Storage type of the packed projection and of the output (the staged VALUE map is fp16 in either case: the blend runs on
// packed halves, and gfx950 has no packed bf16 FMA -- a bf16 model's value projection writes fp16, three mantissa bits
// more than bf16, through codetr_linear_bf16_f16out)
struct EF16 {
  __device__ static float lo(unsigned u) { return (float)as_h2(u)[0]; }
  __device__ static float hi(unsigned u) { return (float)as_h2(u)[1]; }
  __device__ static u32x4 pack8(const float (&a)[8]) {
    return u32x4{pack_h2(a[0], a[1]), pack_h2(a[2], a[3]), pack_h2(a[4], a[5]), pack_h2(a[6], a[7])};
  }
};
struct EBF16 {
  __device__ static float lo(unsigned u) { return __uint_as_float(u << 16); }
  __device__ static float hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
  __device__ static unsigned pk(float a, float b) {   // v_cvt_pk_bf16_f32 (round to nearest even)
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
  }
  __device__ static u32x4 pack8(const float (&a)[8]) {
    return u32x4{pk(a[0], a[1]), pk(a[2], a[3]), pk(a[4], a[5]), pk(a[6], a[7])};
  }
};

// What a wave keeps about one level (all wave-uniform: SGPRs)
struct Lv {
  int W, H, start;           // level size, first pixel inside S
  int px0, py0;              // window origin, extended coordinates
  int xspan, yspan;          // largest (x0 - px0), (y0 - py0) whose four corners are staged
  int pw, ph;                // staged columns / rows
  unsigned base;             // LDS byte address of staged pixel (px0, py0)
  unsigned pitch;            // pw * 64
  float vcx, vcy;            // valid pixels along x / y
  int qx0, qy0, qw, slot0;   // the region's queries on this level: rectangle origin, width, first slot
};

struct TileId {
  int b, rx, ry, m;
};
// tile -> (image, region column / row, head): heads innermost (the M slices of a 512-byte pixel row meet in one XCD's L2),
// regions in horizontal bands of kBand rows walked column by column (vertically adjacent regions share most of their
// windows: an L2 hit instead of a second trip over the fabric)
__device__ __forceinline__ TileId decode_tile(unsigned tile, const Geom4& g) {
  TileId t;
  const int unit = fdiv((int)tile, g.M);
  t.m = (int)tile - unit * g.M;
  const int regions = g.RX * g.RY;
  t.b = fdiv(unit, regions);
  const int reg = unit - t.b * regions;
  const int per_band = kBand * g.RX;
  const int band = fdiv(reg, per_band);
  const int r = reg - band * per_band;
  const int y0 = band * kBand;
  const int bh = min(kBand, g.RY - y0);
  t.rx = fdiv(r, bh);
  t.ry = y0 + (r - t.rx * bh);
  return t;
}

// one sample of the lane (level k, the lane's point): image coordinates -> (floor, fraction)
struct Smp {
  float lw, lh;
  int x0, y0;
};
template <class ET>
__device__ __forceinline__ Smp sample_at(const Lv& v, float bx, float by, unsigned o2) {
  Smp s;
  // reference: loc = ref_k + off / (W, H); im = loc * (W, H) - 0.5 with ref_k = centre / (vr_q size_q) * vr_k and
  // vr * size = valid pixel count  ->  im = (centre / vc_q) * vc_k - 0.5 + off          (transformer.py:280-305, cu:241-247)
  // (fmaxf: a NaN coordinate becomes a huge negative one -> outside every window and outside the gate: the sample is
  // dropped, as by the reference's comparisons)
  const float w_im = fmaxf(fmaf(bx, v.vcx, -0.5f) + ET::lo(o2), -3.0e38f);
  const float h_im = fmaxf(fmaf(by, v.vcy, -0.5f) + ET::hi(o2), -3.0e38f);
  s.lw = __builtin_amdgcn_fractf(w_im);
  s.lh = __builtin_amdgcn_fractf(h_im);
  s.x0 = floor_i(w_im);
  s.y0 = floor_i(h_im);
  return s;
}

template <int T>
struct Cfg {
  static constexpr int kWaves = T / 64;
  static constexpr int kPairs = T / 4;      // (query, head) pairs per workgroup iteration
};

// ---- one PASS = levels [LV0, LV0 + NLV) ---------------------------------------------------------------------------
// prepared sample: LDS address of its (x0, y0) row, the four corner weights as two packed half pairs (zero when the sample
// is not served from LDS)
struct Prep {
  unsigned ad, w01, w23;
};

template <class ET, int LV0, int NLV>
__device__ __forceinline__ bool prepare(Prep (&pp)[NLV], const Lv (&lv)[kL], const float (&aw)[kL], const unsigned (&o2)[kL],
                                        float bx, float by) {
  bool allok = true;
#pragma unroll
  for (int i = 0; i < NLV; ++i) {
    const Lv& v = lv[LV0 + i];
    const Smp s = sample_at<ET>(v, bx, by, o2[LV0 + i]);
    const unsigned dx = (unsigned)(s.x0 - v.px0), dy = (unsigned)(s.y0 - v.py0);
    const bool ok = dx <= (unsigned)v.xspan && dy <= (unsigned)v.yspan;   // all four corners are staged (or zero border)
    const float a = aw[LV0 + i];
    const float wy1 = s.lh * a, wy0 = a - wy1, wx0 = 1.f - s.lw;
    const unsigned w01 = pack_h2(wy0 * wx0, wy0 * s.lw), w23 = pack_h2(wy1 * wx0, wy1 * s.lw);
    const unsigned a_in = __umul24(dy, v.pitch) + v.base + (dx << 6);
    pp[i].ad = ok ? a_in : v.base;   // a finite staged row for the (zero-weight) reads of a sample served elsewhere
    pp[i].w01 = ok ? w01 : 0u;
    pp[i].w23 = ok ? w23 : 0u;
    allok = allok && ok;
  }
  return allok;
}

// gather of one iteration: 4 * NLV sample slots, one per step; rows of step s + 1 requested before the arithmetic of step s
template <int LV0, int NLV>
__device__ __forceinline__ void gather(float (&acc)[8], const Prep (&pp)[NLV], const Lv (&lv)[kL], const unsigned lds_lane) {
  using LV = const __attribute__((address_space(3))) f16x8*;
  constexpr int NS = 4 * NLV, DEPTH = 1, NB = DEPTH + 1;   // (rows two / three steps ahead measured slower: registers)
  f16x8 rows[NB][4];
  unsigned wA[NB], wB[NB];
  // DPP hazard: a VGPR written by a vector instruction may be read by a DPP instruction only two wait states later.  The
  // compiler inserts them for its own DPP forms but cannot see inside quad_bcast_add's inline assembly, and the addresses
  // may have been written by the instruction right before (preparation directly in front of the gather: measured wrong
  // results without this).  One s_nop per gather, tied to the address registers so that nothing moves across it.
  unsigned adr[NLV];
#pragma unroll
  for (int i = 0; i < NLV; ++i) adr[i] = pp[i].ad;
  if (NLV == 1) asm volatile("s_nop 1" : "+v"(adr[0]));
  else asm volatile("s_nop 1" : "+v"(adr[0]), "+v"(adr[NLV - 1]));
  auto fetch = [&](int s_, int buf) {
    const int o = s_ & 3, i = s_ >> 2;
    const unsigned a0 = quad_bcast_add(adr[i], o, lds_lane);
    const unsigned a1 = a0 + lv[LV0 + i].pitch;
    wA[buf] = quad_bcast_u(pp[i].w01, o);
    wB[buf] = quad_bcast_u(pp[i].w23, o);
    rows[buf][0] = *(LV)(uintptr_t)a0;
    rows[buf][1] = *(LV)(uintptr_t)(a0 + 64);
    rows[buf][2] = *(LV)(uintptr_t)a1;
    rows[buf][3] = *(LV)(uintptr_t)(a1 + 64);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch(d, d);
  h2 h[4];
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_) {
    const int b = s_ % NB;
    if (s_ + DEPTH < NS) fetch(s_ + DEPTH, (s_ + DEPTH) % NB);
    const h2 a = as_h2(wA[b]), c = as_h2(wB[b]);
    const h2 w4[4] = {h2{a[0], a[0]}, h2{a[1], a[1]}, h2{c[0], c[0]}, h2{c[1], c[1]}};
#pragma unroll
    for (int cr = 0; cr < 4; ++cr) {
      const f16x8 r = rows[b][cr];
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

// samples the windows do not serve: re-derived with the reference's gate / corner logic (cu:52-71, 249), queued per pair
// (16-byte records in LDS), and added from global memory by the pair's four lanes, kQ records per round (requesting the
// first round's rows before the gather loop needs 20 more live registers: 91 spills at four waves per SIMD, and level at
// three -- profiles/r05_msda_encoder4_sweep.txt; not kept).
template <class ET, int LV0, int NLV>
struct Fix {
  u32x4 rec[NLV];
  unsigned bad;
  int cnt;
  f16x8 r4[kQ][4];
  u32x4 rc[kQ];

  __device__ __forceinline__ void push(u32x4* __restrict__ queue, int base, int sub) {
#pragma unroll
    for (int i = 0; i < NLV; ++i) {
      const int pt = sub + 4 * i;
      if ((bad >> pt) & 1u) {
        const int pos = __builtin_popcount(bad & ((1u << pt) - 1u)) - base;
        if ((unsigned)pos < (unsigned)kQ) queue[pos] = rec[i];
      }
    }
  }
  __device__ __forceinline__ void request(const u32x4* __restrict__ queue, int base, const unsigned char* __restrict__ vhead,
                                          unsigned pix_bytes) {
#pragma unroll
    for (int j = 0; j < kQ; ++j) {
      if (base + j < cnt) {
        rc[j] = queue[j];
        const unsigned char* p = vhead + (size_t)rc[j][0] * pix_bytes;
        const unsigned xs = (rc[j][1] >> 16) * pix_bytes, ys = (rc[j][1] & 0xffffu) * pix_bytes;
        r4[j][0] = *reinterpret_cast<const f16x8*>(p);
        r4[j][1] = *reinterpret_cast<const f16x8*>(p + xs);
        r4[j][2] = *reinterpret_cast<const f16x8*>(p + ys);
        r4[j][3] = *reinterpret_cast<const f16x8*>(p + ys + xs);
      }
    }
  }
  __device__ __forceinline__ void blend(float (&acc)[8], int base) {
#pragma unroll
    for (int j = 0; j < kQ; ++j) {
      if (base + j < cnt) {
        const h2 a = as_h2(rc[j][2]), c = as_h2(rc[j][3]);
        const h2 w4[4] = {h2{a[0], a[0]}, h2{a[1], a[1]}, h2{c[0], c[0]}, h2{c[1], c[1]}};
        h2 h[4];
#pragma unroll
        for (int cr = 0; cr < 4; ++cr)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const h2 v = {r4[j][cr][2 * jj], r4[j][cr][2 * jj + 1]};
            h[jj] = cr == 0 ? v * w4[cr] : __builtin_elementwise_fma(v, w4[cr], h[jj]);
          }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc_h2(acc[2 * jj], acc[2 * jj + 1], h[jj]);
      }
    }
  }

  __device__ __forceinline__ void derive(const Lv (&lv)[kL], const float (&aw)[kL], const unsigned (&o2)[kL], float bx, float by,
                                         const int sub) {
    bad = 0;
#pragma unroll
    for (int i = 0; i < NLV; ++i) {
      const Lv& v = lv[LV0 + i];
      const Smp s = sample_at<ET>(v, bx, by, o2[LV0 + i]);
      const unsigned dx = (unsigned)(s.x0 - v.px0), dy = (unsigned)(s.y0 - v.py0);
      const bool ok = dx <= (unsigned)v.xspan && dy <= (unsigned)v.yspan;
      // cu:249: h_im > -1 && w_im > -1 && h_im < H && w_im < W  <=>  floor in [-1, size - 1]
      const bool gate = (unsigned)(s.x0 + 1) <= (unsigned)v.W && (unsigned)(s.y0 + 1) <= (unsigned)v.H;
      const float a = aw[LV0 + i];
      const float wy0 = s.y0 >= 0 ? (1.f - s.lh) * a : 0.f, wy1 = s.y0 + 1 <= v.H - 1 ? s.lh * a : 0.f;   // cu:52-71
      const float wx0 = s.x0 >= 0 ? 1.f - s.lw : 0.f, wx1 = s.x0 + 1 <= v.W - 1 ? s.lw : 0.f;
      const int x0c = max(s.x0, 0), x1c = min(s.x0 + 1, v.W - 1), y0c = max(s.y0, 0), y1c = min(s.y0 + 1, v.H - 1);
      rec[i] = u32x4{(unsigned)(v.start + y0c * v.W + x0c), (unsigned)((y1c - y0c) * v.W) | ((unsigned)(x1c - x0c) << 16),
                     pack_h2(wy0 * wx0, wy0 * wx1), pack_h2(wy1 * wx0, wy1 * wx1)};
      bad |= (!ok && gate) ? 1u << (sub + 4 * i) : 0u;
    }
    bad |= dpp_u<kXor2>(bad);
    bad |= dpp_u<kXor1>(bad);
    cnt = __builtin_popcount(bad);
  }
};

// every round = queue, request, blend
template <class ET, int LV0, int NLV>
__device__ __forceinline__ void fixup(float (&acc)[8], const Lv (&lv)[kL], const float (&aw)[kL], const unsigned (&o2)[kL],
                                      float bx, float by, u32x4* __restrict__ queue, const unsigned char* __restrict__ vhead,
                                      const unsigned pix_bytes, const int sub) {
  Fix<ET, LV0, NLV> fx;
  fx.derive(lv, aw, o2, bx, by, sub);
  for (int base = 0; __builtin_amdgcn_ballot_w64(fx.cnt > base) != 0; base += kQ) {
    fx.push(queue, base, sub);
    fx.request(queue, base, vhead, pix_bytes);
    fx.blend(acc, base);
  }
}

// LDS: [staged rows of ONE pass | fix-up queues (T / 4 pairs x kQ x 16 B)]
template <class ET, int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) void msda_encoder_v4_kernel(
    const _Float16* __restrict__ value, const unsigned short* __restrict__ packed, unsigned short* __restrict__ out,
    const Geom4 g, const int packed_stride) {
  constexpr unsigned kRow = 64;
  constexpr int kWaves = Cfg<T>::kWaves, kPairs = Cfg<T>::kPairs;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x;
  const int wave = uni(tid >> 6), lane = tid & 63, sub = lane & 3, pl = lane >> 2;
  const int M = g.M;
#ifdef MSDA4_STAMPS
  unsigned long long m4_s[20];
  for (int i = 0; i < 20; ++i) m4_s[i] = 0;
  M4_T(k0);
#endif
  // one pixel of one head = 64 bytes; consecutive pixels of a head are 64 B apart in the head-major map (a staged window
  // row is ONE contiguous run: a DMA instruction reads 8 whole 128-byte lines), M * 64 B apart in the op's own layout
  // (16 half lines per DMA instruction)
  const unsigned pix_bytes = g.head_major ? kRow : (unsigned)M * kRow;

  // ---- wave-uniform geometry -> scalar registers ----
  const TileId t0 = decode_tile(xcd_tile(blockIdx.x, gridDim.x), g);
  const TileId t = {uni(t0.b), uni(t0.rx), uni(t0.ry), uni(t0.m)};
  Lv lv[kL];
  int total;
  {
    // 40 divisions (per level: two window bounds and two query bounds per axis), one per LANE: lane 8 l + j computes item
    // j of level l -- 0: first staged column, 1: last staged column (before the "at least two columns" clamp), 2 / 3: rows,
    // 4 / 5: first query column of this region / of the next, 6 / 7: rows -- and the results return to scalar registers
    // with v_readlane.  (Computed level by level in wave-uniform code this was 600 vector instructions per wave, a fifth of
    // the kernel.)
    const int gl = lane >> 3 > kL - 1 ? kL - 1 : lane >> 3, gj = lane & 7;
    int nW = g.W[0], nH = g.H[0], wword = 0;
    const int mw = t.m < kMaxM ? t.m : kMaxM - 1;
#pragma unroll
    for (int l = 0; l < kL; ++l) {
      int ww;
      __builtin_memcpy(&ww, g.win[mw][l], 4);
      nW = gl == l ? g.W[l] : nW;
      nH = gl == l ? g.H[l] : nH;
      wword = gl == l ? ww : wword;
    }
    const bool ay = (gj & 2) != 0, isq = gj >= 4, plus = (gj & 1) != 0;
    const int n = ay ? nH : nW, R = ay ? g.RY : g.RX, rr = (ay ? t.ry : t.rx) + (plus ? 1 : 0);
    const int wv = (int)(signed char)((unsigned)wword >> (8 * (gj & 3)));
    const int num = 2 * rr * n + (isq ? R - 1 : 2 * wv * R - R);
    const int numd = (!isq && plus) ? num + 2 * R - 1 : num;
    const int q = fdiv(numd > 0 ? numd : 0, 2 * R);
    const int res = isq ? q : plus ? (num <= 0 ? 0 : (q > n ? n : q)) : (num < 0 ? -1 : (q > n - 1 ? n - 1 : q));
    int slot = 0;
    unsigned base = lds0;
#pragma unroll
    for (int l = 0; l < kL; ++l) {
      Lv& v = lv[l];
      v.W = g.W[l];
      v.H = g.H[l];
      v.start = g.start[l];
      v.px0 = __builtin_amdgcn_readlane(res, 8 * l + 0);
      int px1 = __builtin_amdgcn_readlane(res, 8 * l + 1);
      v.py0 = __builtin_amdgcn_readlane(res, 8 * l + 2);
      int py1 = __builtin_amdgcn_readlane(res, 8 * l + 3);
      px1 = px1 < v.px0 + 1 ? v.px0 + 1 : px1;
      py1 = py1 < v.py0 + 1 ? v.py0 + 1 : py1;
      v.pw = px1 - v.px0 + 1;
      v.ph = py1 - v.py0 + 1;
      v.xspan = v.pw - 2;
      v.yspan = v.ph - 2;
      v.pitch = (unsigned)v.pw * kRow;
      if (g.first[l] == l) base = lds0;       // a new pass starts at the front of the buffer
      v.base = base;
      base += (unsigned)(v.pw * v.ph) * kRow;
      const float* vc = g.vcounts + ((size_t)t.b * kL + l) * 2;
      v.vcx = unif(vc[0]);
      v.vcy = unif(vc[1]);
      v.qx0 = __builtin_amdgcn_readlane(res, 8 * l + 4);
      v.qy0 = __builtin_amdgcn_readlane(res, 8 * l + 6);
      v.qw = __builtin_amdgcn_readlane(res, 8 * l + 5) - v.qx0;
      const int qh = __builtin_amdgcn_readlane(res, 8 * l + 7) - v.qy0;
      v.slot0 = slot;
      slot += v.qw * qh;
    }
    total = slot;
  }
  const int n_it = total > wave * 16 ? (total - wave * 16 + kPairs - 1) / kPairs : 0;   // <= kMaxIt (host-checked)

  const unsigned char* vhead0 = reinterpret_cast<const unsigned char*>(value) +
                                (g.head_major ? ((size_t)t.b * M + t.m) * g.S : (size_t)t.b * g.S * M + t.m) * kRow;   // (uniform)
  const unsigned char* vhead = vhead0 + sub * 16;
  const unsigned char* prow = reinterpret_cast<const unsigned char*>(packed) + (size_t)t.b * g.S * packed_stride * 2 +
                              (t.m * 64 + sub * 16) * 2;

  // ---- the wave's queries: slot -> (level, pixel) -> flattened index, centre / valid count; raw operands requested ----
  // per-level tables, level l in LANE l (a select between two scalars costs two vector instructions -- one constant-bus
  // operand per instruction -- so a per-lane level lookup is a ds_bpermute from these instead of a select chain)
  int tS0 = 0, tA = 0, tB = 0, tSt = 0;
  float tRx = 0.f, tRy = 0.f;
#pragma unroll
  for (int l = 0; l < kL; ++l) {
    const bool me = lane == l;
    tS0 = me ? lv[l].slot0 : tS0;
    tA = me ? (lv[l].qx0 | (lv[l].qy0 << 16)) : tA;
    tB = me ? (lv[l].W | (lv[l].qw << 16)) : tB;
    tSt = me ? lv[l].start : tSt;
    tRx = me ? lv[l].vcx : tRx;
    tRy = me ? lv[l].vcy : tRy;
  }
  tRx = __builtin_amdgcn_rcpf(tRx);
  tRy = __builtin_amdgcn_rcpf(tRy);
  int qs[kMaxIt];
  float bx[kMaxIt], by[kMaxIt];
  u32x4 rawA[kMaxIt], rawB[kMaxIt];
#pragma unroll
  for (int it = 0; it < kMaxIt; ++it) {
    qs[it] = 0;
    bx[it] = by[it] = 0.f;
    rawA[it] = rawB[it] = u32x4{0u, 0u, 0u, 0u};
    if (it < n_it) {
      int sl = (it * kWaves + wave) * 16 + pl;
      sl = sl < total ? sl : total - 1;
      int lvq = 0;
#pragma unroll
      for (int l = 1; l < kL; ++l) lvq += sl >= lv[l].slot0 ? 4 : 0;     // byte address of the level's lane
      const int s0 = __builtin_amdgcn_ds_bpermute(lvq, tS0), cA = __builtin_amdgcn_ds_bpermute(lvq, tA);
      const int cB = __builtin_amdgcn_ds_bpermute(lvq, tB), st = __builtin_amdgcn_ds_bpermute(lvq, tSt);
      const float rx_ = __int_as_float(__builtin_amdgcn_ds_bpermute(lvq, __float_as_int(tRx)));
      const float ry_ = __int_as_float(__builtin_amdgcn_ds_bpermute(lvq, __float_as_int(tRy)));
      const int tq = sl - s0, qw = cB >> 16, W = cB & 0xffff;
      const int yy = (int)(((float)tq + 0.5f) * __builtin_amdgcn_rcpf((float)qw));
      const int y = (cA >> 16) + yy, x = (cA & 0xffff) + (tq - yy * qw);
      qs[it] = st + y * W + x;
      // get_reference_points (transformer.py:280-305): centre / (valid ratio * size) = centre / valid pixel count
      bx[it] = ((float)x + 0.5f) * rx_;
      by[it] = ((float)y + 0.5f) * ry_;
      const unsigned char* pr = prow + (size_t)((unsigned)qs[it] * (unsigned)packed_stride) * 2;
      rawA[it] = *reinterpret_cast<const u32x4*>(pr);
      rawB[it] = *reinterpret_cast<const u32x4*>(pr + 16);
    }
  }

  // ---- softmax over the pair's 20 logits (5 per lane, quad reductions); the offsets stay packed ----
  float aw[kMaxIt][kL], acc[kMaxIt][8];
  unsigned o2[kMaxIt][kL];
#pragma unroll
  for (int it = 0; it < kMaxIt; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[it][j] = 0.f;
    o2[it][0] = rawA[it][0];
    o2[it][1] = rawA[it][1];
    o2[it][2] = rawA[it][2];
    o2[it][3] = rawA[it][3];
    o2[it][4] = rawB[it][0];
    const float lg[kL] = {ET::lo(rawB[it][1]), ET::hi(rawB[it][1]), ET::lo(rawB[it][2]), ET::hi(rawB[it][2]), ET::lo(rawB[it][3])};
    float mx = lg[0];
#pragma unroll
    for (int k = 1; k < kL; ++k) mx = fmaxf(mx, lg[k]);
    mx = fmaxf(mx, dpp_f<kXor2>(mx));
    mx = fmaxf(mx, dpp_f<kXor1>(mx));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < kL; ++k) {
      aw[it][k] = __expf(lg[k] - mx);
      sum += aw[it][k];
    }
    sum += dpp_f<kXor2>(sum);
    sum += dpp_f<kXor1>(sum);
    const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int k = 0; k < kL; ++k) aw[it][k] *= inv;
  }

  u32x4* queue = reinterpret_cast<u32x4*>(smem + g.queue_off) + (size_t)(wave * 16 + pl) * kQ;
  const unsigned lds_lane = (unsigned)sub * 16;

  auto run_pass = [&](auto lv0_c, auto nlv_c) {
    constexpr int LV0 = decltype(lv0_c)::value, NLV = decltype(nlv_c)::value;
    constexpr int PI = LV0 == 0 ? 0 : LV0 == 1 ? 1 : 2;
    (void)PI;
    M4_T(a);
    if (LV0 > 0 && !(kAbl & 16)) __syncthreads();   // every wave is done reading the previous pass's rows
    M4_T(b);
    // -- the pass's windows -> LDS, ROW-WISE: a wave takes window rows y = wave, wave + kWaves, ...; one LDS-DMA instruction
    // moves 16 pixels of the row (4 lanes x 16 B per pixel = its 64-byte head slice) from a wave-uniform 64-bit base in
    // SGPRs + a per-lane constant offset to the uniform LDS address of the row's chunk (+ 16 B per lane): no per-lane row /
    // column arithmetic (the piece-linear walk of rounds 2-4 spent ~14 vector instructions per DMA on a division).  Cells
    // outside the image -- the zero border -- are written by ds_write instead.
#pragma unroll
    for (int i = 0; i < NLV; ++i) {
      const Lv& v = lv[LV0 + i];
      const int px_l = lane >> 2;                                      // pixel of the chunk this lane serves
      const unsigned voff = (unsigned)px_l * pix_bytes + (unsigned)sub * 16;
      const int chunks = (v.pw + 15) >> 4;
      for (int y = wave; y < ((kAbl & 2) ? 0 : v.ph); y += kWaves) {
        const int gy = v.py0 + y;
        const bool row_in = (unsigned)gy < (unsigned)v.H;              // (uniform)
        for (int c = 0; c < chunks; ++c) {
          const int x0 = 16 * c;                                       // window column of lane 0's pixel
          const unsigned dst = (v.base - lds0) + (unsigned)(y * v.pw + x0) * kRow;
          const int gx = v.px0 + x0 + px_l;
          const bool mine = x0 + px_l < v.pw;
          const bool col_in = (unsigned)gx < (unsigned)v.W;
          if (row_in) {
            if (mine && col_in)
              lds_dma16(vhead0 + (ptrdiff_t)(v.start + gy * v.W + v.px0 + x0) * (ptrdiff_t)pix_bytes, voff, lds0 + dst);
            if (mine && !col_in) *reinterpret_cast<u32x4*>(smem + dst + lane * 16) = u32x4{0u, 0u, 0u, 0u};
          } else if (mine) {
            *reinterpret_cast<u32x4*>(smem + dst + lane * 16) = u32x4{0u, 0u, 0u, 0u};
          }
        }
      }
    }
    M4_T(c);
    if (!(kAbl & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    M4_T(d);
    if (!(kAbl & 16)) __syncthreads();
    M4_T(e);
    // -- per iteration: addresses / weights of the pass (right in front of their gather: computed for all iterations under
    // the DMA they cost 12 more live registers and 6 spills at the 128-register budget, and measured 3-5 % slower -- the DMA's
    // latency is covered by the other workgroup of the CU either way), the gather, then the samples the windows do not serve
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it)
      if (it < n_it && !(kAbl & 4)) {
        Prep pp[NLV];
#pragma unroll
        for (int i = 0; i < NLV; ++i) pp[i] = Prep{lv[LV0 + i].base, 0u, 0u};
        bool clean = true;
        if (!(kAbl & 8)) {
          const bool ok = prepare<ET, LV0, NLV>(pp, lv, aw[it], o2[it], bx[it], by[it]);
          clean = __builtin_amdgcn_ballot_w64(!ok) == 0;
        }
        gather<LV0, NLV>(acc[it], pp, lv, lds_lane);
        if (!clean) fixup<ET, LV0, NLV>(acc[it], lv, aw[it], o2[it], bx[it], by[it], queue, vhead, pix_bytes, sub);
      }
    M4_T(f);
    // 0: barrier in front, 1: DMA issue, 2: wait for the data, 3: barrier, 4: gather + fix-up
    M4_PUT(PI * 5 + 0, m4_tb - m4_ta); M4_PUT(PI * 5 + 1, m4_tc - m4_tb); M4_PUT(PI * 5 + 2, m4_td - m4_tc);
    M4_PUT(PI * 5 + 3, m4_te - m4_td); M4_PUT(PI * 5 + 4, m4_tf - m4_te);
  };
  M4_T(k1);
  run_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  run_pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
  run_pass(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});

  unsigned char* orow = reinterpret_cast<unsigned char*>(out) + ((size_t)t.b * g.S * M + t.m) * kRow + sub * 16;
#pragma unroll
  for (int it = 0; it < kMaxIt; ++it)
    if (it < n_it && (it * kWaves + wave) * 16 + pl < total) {
      const u32x4 o = ET::pack8(acc[it]);
      if (!(kAbl & 32) || o[0] == 0x12345678u) *reinterpret_cast<u32x4*>(orow + (size_t)((unsigned)qs[it] * ((unsigned)M * kRow))) = o;   // out is [B, S, M, 32] in either case
    }
#ifdef MSDA4_STAMPS
  M4_T(k2);
  m4_s[15] = m4_tk1 - m4_tk0;   // prologue (geometry, operand loads, softmax)
  m4_s[16] = m4_tk2 - m4_tk0;   // whole workgroup
  m4_s[17] = m4_tk0;            // start stamp (for the launch's span)
  m4_s[18] = m4_tk2;
  if (tid == 0 && g_msda4_stamps)
    for (int i = 0; i < 20; ++i) g_msda4_stamps[(size_t)blockIdx.x * 20 + i] = m4_s[i];
#endif
}

// ---- round-6 persistent form ("v5"): ONE 1024-thread workgroup per CU, windows double-buffered -----------------------
// What the stamps of the kernel above show (tools/msda4_stamps.py, profiles/r06_msda_encoder_stamps.txt): a workgroup spends
// a third of its life staging -- the DMA instructions of a pass stall at issue while the CU takes their data in at ~20 B/clk,
// then the workgroup waits for the last of them and for its slowest wave -- and none of that overlaps its OWN gather; the
// other workgroup of the CU hides a third of it.  Here the staging of the next pass (and, during the last pass of a tile,
// the first pass of the NEXT tile together with that tile's operand rows) is issued before the gather of the current one
// into the other of two window buffers, so that a pass boundary costs one barrier; the workgroup is persistent and walks
// the tiles of its XCD's share.  Same gather, preparation and fix-up code as above.
constexpr int kIt5 = 2;       // iterations per wave: 2 x 256 pairs = the 512 queries of a 24 x 16 region's pyramid share

// the 40 divisions of a tile's geometry, one per lane (lane 8 l + j: item j of level l; see the kernel above)
__device__ __forceinline__ int tile_res(const Geom4& g, const TileId& t, int lane) {
  const int gl = lane >> 3 > kL - 1 ? kL - 1 : lane >> 3, gj = lane & 7;
  int nW = g.W[0], nH = g.H[0], wword = 0;
  const int mw = t.m < kMaxM ? t.m : kMaxM - 1;
#pragma unroll
  for (int l = 0; l < kL; ++l) {
    int ww;
    __builtin_memcpy(&ww, g.win[mw][l], 4);
    nW = gl == l ? g.W[l] : nW;
    nH = gl == l ? g.H[l] : nH;
    wword = gl == l ? ww : wword;
  }
  const bool ay = (gj & 2) != 0, isq = gj >= 4, plus = (gj & 1) != 0;
  const int n = ay ? nH : nW, R = ay ? g.RY : g.RX, rr = (ay ? t.ry : t.rx) + (plus ? 1 : 0);
  const int wv = (int)(signed char)((unsigned)wword >> (8 * (gj & 3)));
  const int num = 2 * rr * n + (isq ? R - 1 : 2 * wv * R - R);
  const int numd = (!isq && plus) ? num + 2 * R - 1 : num;
  const int q = fdiv(numd > 0 ? numd : 0, 2 * R);
  return isq ? q : plus ? (num <= 0 ? 0 : (q > n ? n : q)) : (num < 0 ? -1 : (q > n - 1 ? n - 1 : q));
}

// scalar geometry of a tile from its division results; pass k of the tile is staged in buffer (p0 + k) & 1
__device__ __forceinline__ int build_lv(const Geom4& g, const TileId& t, int res, unsigned lds0, unsigned buf_bytes, int p0,
                                        Lv (&lv)[kL]) {
  constexpr unsigned kRow = 64;
  int slot = 0;
  unsigned base = lds0;
#pragma unroll
  for (int l = 0; l < kL; ++l) {
    Lv& v = lv[l];
    v.W = g.W[l];
    v.H = g.H[l];
    v.start = g.start[l];
    v.px0 = __builtin_amdgcn_readlane(res, 8 * l + 0);
    int px1 = __builtin_amdgcn_readlane(res, 8 * l + 1);
    v.py0 = __builtin_amdgcn_readlane(res, 8 * l + 2);
    int py1 = __builtin_amdgcn_readlane(res, 8 * l + 3);
    px1 = px1 < v.px0 + 1 ? v.px0 + 1 : px1;
    py1 = py1 < v.py0 + 1 ? v.py0 + 1 : py1;
    v.pw = px1 - v.px0 + 1;
    v.ph = py1 - v.py0 + 1;
    v.xspan = v.pw - 2;
    v.yspan = v.ph - 2;
    v.pitch = (unsigned)v.pw * kRow;
    if (g.first[l] == l) {   // a new pass starts at the front of its buffer
      const int k = l == 0 ? 0 : l == 1 ? 1 : 2;
      base = lds0 + (((p0 + k) & 1) ? buf_bytes : 0u);
    }
    v.base = base;
    base += (unsigned)(v.pw * v.ph) * kRow;
    const float* vc = g.vcounts + ((size_t)t.b * kL + l) * 2;
    v.vcx = unif(vc[0]);
    v.vcy = unif(vc[1]);
    v.qx0 = __builtin_amdgcn_readlane(res, 8 * l + 4);
    v.qy0 = __builtin_amdgcn_readlane(res, 8 * l + 6);
    v.qw = __builtin_amdgcn_readlane(res, 8 * l + 5) - v.qx0;
    const int qh = __builtin_amdgcn_readlane(res, 8 * l + 7) - v.qy0;
    v.slot0 = slot;
    slot += v.qw * qh;
  }
  return slot;
}

// the windows of levels [LV0, LV0 + NLV) -> LDS, row-wise (see the kernel above), rows shared out over KW waves
template <int LV0, int NLV, int KW>
__device__ __forceinline__ void stage_pass(const Lv (&lv)[kL], const unsigned char* __restrict__ vhead0, unsigned pix_bytes,
                                           unsigned char* smem, unsigned lds0, int wave, int lane) {
  constexpr unsigned kRow = 64;
  const int sub = lane & 3;
#pragma unroll
  for (int i = 0; i < NLV; ++i) {
    const Lv& v = lv[LV0 + i];
    const int px_l = lane >> 2;
    const unsigned voff = (unsigned)px_l * pix_bytes + (unsigned)sub * 16;
    const int chunks = (v.pw + 15) >> 4;
    for (int y = wave; y < ((kAbl & 2) ? 0 : v.ph); y += KW) {
      const int gy = v.py0 + y;
      const bool row_in = (unsigned)gy < (unsigned)v.H;
      for (int c = 0; c < chunks; ++c) {
        const int x0 = 16 * c;
        const unsigned dst = (v.base - lds0) + (unsigned)(y * v.pw + x0) * kRow;
        const int gx = v.px0 + x0 + px_l;
        const bool mine = x0 + px_l < v.pw;
        const bool col_in = (unsigned)gx < (unsigned)v.W;
        if (row_in) {
          if (mine && col_in)
            lds_dma16(vhead0 + (ptrdiff_t)(v.start + gy * v.W + v.px0 + x0) * (ptrdiff_t)pix_bytes, voff, lds0 + dst);
          if (mine && !col_in) *reinterpret_cast<u32x4*>(smem + dst + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        } else if (mine) {
          *reinterpret_cast<u32x4*>(smem + dst + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
  }
}

struct Q5 {   // a wave's queries of one tile: flattened index, reference-point numerators, raw operand rows
  int qs[kIt5];
  float bx[kIt5], by[kIt5];
  u32x4 rawA[kIt5], rawB[kIt5];
};

// slot -> (level, pixel) -> flattened query index, centre / valid count; the operand rows are requested
__device__ __forceinline__ void tile_queries(const Lv (&lv)[kL], int total, int wave, int lane, const unsigned char* prow,
                                             int packed_stride, Q5& q) {
  const int pl = lane >> 2;
  int tS0 = 0, tA = 0, tB = 0, tSt = 0;
  float tRx = 0.f, tRy = 0.f;
#pragma unroll
  for (int l = 0; l < kL; ++l) {
    const bool me = lane == l;
    tS0 = me ? lv[l].slot0 : tS0;
    tA = me ? (lv[l].qx0 | (lv[l].qy0 << 16)) : tA;
    tB = me ? (lv[l].W | (lv[l].qw << 16)) : tB;
    tSt = me ? lv[l].start : tSt;
    tRx = me ? lv[l].vcx : tRx;
    tRy = me ? lv[l].vcy : tRy;
  }
  tRx = __builtin_amdgcn_rcpf(tRx);
  tRy = __builtin_amdgcn_rcpf(tRy);
  const int n_it = total > wave * 16 ? (total - wave * 16 + 255) / 256 : 0;
#pragma unroll
  for (int it = 0; it < kIt5; ++it) {
    q.qs[it] = 0;
    q.bx[it] = q.by[it] = 0.f;
    q.rawA[it] = q.rawB[it] = u32x4{0u, 0u, 0u, 0u};
    if (it < n_it) {
      int sl = (it * 16 + wave) * 16 + pl;
      sl = sl < total ? sl : total - 1;
      int lvq = 0;
#pragma unroll
      for (int l = 1; l < kL; ++l) lvq += sl >= lv[l].slot0 ? 4 : 0;
      const int s0 = __builtin_amdgcn_ds_bpermute(lvq, tS0), cA = __builtin_amdgcn_ds_bpermute(lvq, tA);
      const int cB = __builtin_amdgcn_ds_bpermute(lvq, tB), st = __builtin_amdgcn_ds_bpermute(lvq, tSt);
      const float rx_ = __int_as_float(__builtin_amdgcn_ds_bpermute(lvq, __float_as_int(tRx)));
      const float ry_ = __int_as_float(__builtin_amdgcn_ds_bpermute(lvq, __float_as_int(tRy)));
      const int tq = sl - s0, qw = cB >> 16, W = cB & 0xffff;
      const int yy = (int)(((float)tq + 0.5f) * __builtin_amdgcn_rcpf((float)qw));
      const int y = (cA >> 16) + yy, x = (cA & 0xffff) + (tq - yy * qw);
      q.qs[it] = st + y * W + x;
      q.bx[it] = ((float)x + 0.5f) * rx_;
      q.by[it] = ((float)y + 0.5f) * ry_;
      const unsigned char* pr = prow + (size_t)((unsigned)q.qs[it] * (unsigned)packed_stride) * 2;
      q.rawA[it] = *reinterpret_cast<const u32x4*>(pr);
      q.rawB[it] = *reinterpret_cast<const u32x4*>(pr + 16);
    }
  }
}

// LDS: [window buffer 0 | window buffer 1 | fix-up queues]
template <class ET>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void msda_encoder_v5_kernel(
    const _Float16* __restrict__ value, const unsigned short* __restrict__ packed, unsigned short* __restrict__ out,
    const Geom4 g, const int packed_stride, const int ntiles) {
  constexpr unsigned kRow = 64;
  constexpr int kWaves = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  const int tid = threadIdx.x;
  const int wave = uni(tid >> 6), lane = tid & 63, sub = lane & 3, pl = lane >> 2;
  const int M = g.M;
  const unsigned pix_bytes = g.head_major ? kRow : (unsigned)M * kRow;
  const unsigned buf_bytes = g.queue_off >> 1;
  u32x4* queue = reinterpret_cast<u32x4*>(smem + g.queue_off) + (size_t)(wave * 16 + pl) * kQ;
  const unsigned lds_lane = (unsigned)sub * 16;

  // this workgroup's tiles: XCD x = blockIdx & 7 owns a contiguous share of the tile list (neighbouring regions and the
  // eight heads of a region meet in one L2), its workgroups walk it round-robin
  const unsigned xcd = blockIdx.x & 7u, per = gridDim.x >> 3;
  const unsigned tq_ = (unsigned)ntiles >> 3, tr_ = (unsigned)ntiles & 7u;
  const unsigned first = xcd < tr_ ? xcd * (tq_ + 1) : tr_ * (tq_ + 1) + (xcd - tr_) * tq_;
  const unsigned count = tq_ + (xcd < tr_ ? 1u : 0u);
  unsigned k = blockIdx.x >> 3;
  if (k >= count) return;

  auto tile_of = [&](unsigned kk) {
    const TileId t0 = decode_tile(first + kk, g);
    return TileId{uni(t0.b), uni(t0.rx), uni(t0.ry), uni(t0.m)};
  };
  auto head_base = [&](const TileId& t) {
    return reinterpret_cast<const unsigned char*>(value) +
           (g.head_major ? ((size_t)t.b * M + t.m) * g.S : (size_t)t.b * g.S * M + t.m) * kRow;
  };
  auto packed_row = [&](const TileId& t) {
    return reinterpret_cast<const unsigned char*>(packed) + (size_t)t.b * g.S * packed_stride * 2 + (t.m * 64 + sub * 16) * 2;
  };

#ifdef MSDA4_STAMPS
  unsigned long long m5[20];
  for (int i = 0; i < 20; ++i) m5[i] = 0;
  const unsigned long long m5_k0 = __builtin_readcyclecounter();
#define M5_T(i) const unsigned long long m5_t##i = __builtin_readcyclecounter()
#define M5_ACC(k, a, b) m5[k] += m5_t##b - m5_t##a
#else
#define M5_T(i)
#define M5_ACC(k, a, b)
#endif
  // ---- first tile: geometry, operand rows, first pass on its way ----
  int p0 = 0;
  TileId t = tile_of(k);
  Q5 q;
  {
    Lv lv0[kL];
    const int total0 = build_lv(g, t, tile_res(g, t, lane), lds0, buf_bytes, p0, lv0);
    tile_queries(lv0, total0, wave, lane, packed_row(t), packed_stride, q);
    stage_pass<0, 1, kWaves>(lv0, head_base(t), pix_bytes, smem, lds0, wave, lane);
  }

  for (;;) {
    M5_T(0);
    Lv lv[kL];
    const int total = build_lv(g, t, tile_res(g, t, lane), lds0, buf_bytes, p0, lv);
    const int n_it = total > wave * 16 ? (total - wave * 16 + 255) / 256 : 0;
    const unsigned char* vhead0 = head_base(t);
    const unsigned char* vhead = vhead0 + sub * 16;

    // ---- softmax over the pair's 20 logits; the offsets stay packed ----
    float aw[kIt5][kL], acc[kIt5][8];
    unsigned o2[kIt5][kL];
    int qs[kIt5];
    float bx[kIt5], by[kIt5];
#pragma unroll
    for (int it = 0; it < kIt5; ++it) {
      qs[it] = q.qs[it];
      bx[it] = q.bx[it];
      by[it] = q.by[it];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[it][j] = 0.f;
      o2[it][0] = q.rawA[it][0];
      o2[it][1] = q.rawA[it][1];
      o2[it][2] = q.rawA[it][2];
      o2[it][3] = q.rawA[it][3];
      o2[it][4] = q.rawB[it][0];
      const float lg[kL] = {ET::lo(q.rawB[it][1]), ET::hi(q.rawB[it][1]), ET::lo(q.rawB[it][2]), ET::hi(q.rawB[it][2]),
                            ET::lo(q.rawB[it][3])};
      float mx = lg[0];
#pragma unroll
      for (int kk = 1; kk < kL; ++kk) mx = fmaxf(mx, lg[kk]);
      mx = fmaxf(mx, dpp_f<kXor2>(mx));
      mx = fmaxf(mx, dpp_f<kXor1>(mx));
      float sum = 0.f;
#pragma unroll
      for (int kk = 0; kk < kL; ++kk) {
        aw[it][kk] = __expf(lg[kk] - mx);
        sum += aw[it][kk];
      }
      sum += dpp_f<kXor2>(sum);
      sum += dpp_f<kXor1>(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int kk = 0; kk < kL; ++kk) aw[it][kk] *= inv;
    }

    auto gather_pass = [&](auto lv0_c, auto nlv_c) {
      constexpr int LV0 = decltype(lv0_c)::value, NLV = decltype(nlv_c)::value;
#pragma unroll
      for (int it = 0; it < kIt5; ++it)
        if (it < n_it && !(kAbl & 4)) {
          Prep pp[NLV];
#pragma unroll
          for (int i = 0; i < NLV; ++i) pp[i] = Prep{lv[LV0 + i].base, 0u, 0u};
          bool clean = true;
          if (!(kAbl & 8)) {
            const bool ok = prepare<ET, LV0, NLV>(pp, lv, aw[it], o2[it], bx[it], by[it]);
            clean = __builtin_amdgcn_ballot_w64(!ok) == 0;
          }
          gather<LV0, NLV>(acc[it], pp, lv, lds_lane);
          if (!clean) fixup<ET, LV0, NLV>(acc[it], lv, aw[it], o2[it], bx[it], by[it], queue, vhead, pix_bytes, sub);
        }
    };
    auto landed = [&]() {   // my DMA has landed; behind the barrier everybody's has, and everybody is done with the other buffer
      if (!(kAbl & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(kAbl & 16)) __syncthreads();
    };

    M5_T(1);
    landed();                                                                    // pass 0 of this tile
    M5_T(2);
    stage_pass<1, 2, kWaves>(lv, vhead0, pix_bytes, smem, lds0, wave, lane);     // -> the other buffer
    M5_T(3);
    gather_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    M5_T(4);
    landed();
    M5_T(5);
    stage_pass<3, 2, kWaves>(lv, vhead0, pix_bytes, smem, lds0, wave, lane);     // -> pass 0's buffer
    M5_T(6);
    gather_pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
    M5_T(7);
    landed();
    M5_T(8);
    // the next tile: geometry, operand rows, first pass -> the buffer pass 1 has left
    const unsigned kn = k + per;
    const bool more = kn < count;   // (uniform)
    TileId tn = t;
    if (more) {
      tn = tile_of(kn);
      Lv lvn[kL];
      const int totaln = build_lv(g, tn, tile_res(g, tn, lane), lds0, buf_bytes, p0 ^ 1, lvn);
      tile_queries(lvn, totaln, wave, lane, packed_row(tn), packed_stride, q);
      stage_pass<0, 1, kWaves>(lvn, head_base(tn), pix_bytes, smem, lds0, wave, lane);
    }
    M5_T(9);
    gather_pass(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
    M5_T(10);

    unsigned char* orow = reinterpret_cast<unsigned char*>(out) + ((size_t)t.b * g.S * M + t.m) * kRow + sub * 16;
#pragma unroll
    for (int it = 0; it < kIt5; ++it)
      if (it < n_it && (it * kWaves + wave) * 16 + pl < total) {
        const u32x4 o = ET::pack8(acc[it]);
        if (!(kAbl & 32) || o[0] == 0x12345678u) *reinterpret_cast<u32x4*>(orow + (size_t)((unsigned)qs[it] * ((unsigned)M * kRow))) = o;
      }
    M5_T(11);
    // 0: geometry + softmax, 1-3: landed (wait + barrier) of passes 0-2, 4-5: staging issue of passes 1, 2, 6: next tile (geometry,
    // operand rows, pass-0 issue), 7-9: gathers, 10: output, 11: tiles
    M5_ACC(0, 0, 1); M5_ACC(1, 1, 2); M5_ACC(4, 2, 3); M5_ACC(7, 3, 4); M5_ACC(2, 4, 5); M5_ACC(5, 5, 6); M5_ACC(8, 6, 7);
    M5_ACC(3, 7, 8); M5_ACC(6, 8, 9); M5_ACC(9, 9, 10); M5_ACC(10, 10, 11);
#ifdef MSDA4_STAMPS
    m5[11] += 1;
#endif
    if (!more) break;
    k = kn;
    t = tn;
    p0 ^= 1;
  }
#ifdef MSDA4_STAMPS
  m5[12] = __builtin_readcyclecounter() - m5_k0;
  if (tid == 0 && g_msda4_stamps)
    for (int i = 0; i < 20; ++i) g_msda4_stamps[(size_t)blockIdx.x * 20 + i] = m5[i];
#endif
}

// ---- host side ----------------------------------------------------------------------------------------------------
constexpr int kPassFirst[kL] = {0, 1, 1, 3, 3};

struct Plan4 {
  Geom4 g;
  int rc;
  int slots_cap;       // most queries any region has
  size_t lds;          // bytes per workgroup
};

inline Plan4 plan4(const int64_t* shapes, int64_t S, int M, int L, int P, const signed char* win, int region_w, int region_h,
                   int threads) {
  Plan4 pl{};
  Geom4& g = pl.g;
  pl.rc = CODETR_E_BADARG;
  if (!shapes || !win || M <= 0 || L <= 0 || P <= 0 || region_w <= 0 || region_h <= 0) return pl;
  pl.rc = CODETR_E_UNSUPPORTED;
  if (L != kL || P != kP || (threads != 256 && threads != 512 && threads != 1024)) return pl;
  g.M = M;
  g.S = (int)S;
  int64_t sum = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t h = shapes[2 * l], w = shapes[2 * l + 1];
    if (h <= 0 || w <= 0 || h > 32767 || w > 32767) return pl.rc = CODETR_E_BADARG, pl;
    g.H[l] = (int)h;
    g.W[l] = (int)w;
    g.start[l] = (int)sum;
    g.first[l] = kPassFirst[l];
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
      if (m >= kMaxM && memcmp(wn, win + ((size_t)(kMaxM - 1) * L + l) * 4, 4) != 0) return pl;
    }
  int fine = 0;
  for (int l = 1; l < L; ++l)
    if ((int64_t)g.H[l] * g.W[l] > (int64_t)g.H[fine] * g.W[fine]) fine = l;
  g.RX = (g.W[fine] + region_w - 1) / region_w;
  g.RY = (g.H[fine] + region_h - 1) / region_h;
  for (int l = 0; l < L; ++l) {   // the kernel's reciprocal-based floor division is exact below 2^22
    const int64_t nx = 2 * (int64_t)(g.RX + 1) * g.W[l] + (2 * (int64_t)wmax + 3) * g.RX;
    const int64_t ny = 2 * (int64_t)(g.RY + 1) * g.H[l] + (2 * (int64_t)wmax + 3) * g.RY;
    if (nx >= (1 << 22) || ny >= (1 << 22)) return pl;
  }
  int rows_cap = 0, slots = 0;
  for (int m = 0; m < (M < kMaxM ? M : kMaxM); ++m) {
    int rows_pass = 0;
    slots = 0;
    for (int l = 0; l < L; ++l) {
      const signed char* wn = g.win[m][l];
      int pw = 0, ph = 0, qw = 0, qh = 0;
      for (int r = 0; r < g.RX; ++r) {
        const int lo = ext_lo(r, g.W[l], g.RX, wn[0]), hi = ext_hi(r, g.W[l], g.RX, wn[1], lo);
        const int q = q_bound(r + 1, g.W[l], g.RX) - q_bound(r, g.W[l], g.RX);
        pw = hi - lo + 1 > pw ? hi - lo + 1 : pw;
        qw = q > qw ? q : qw;
      }
      for (int r = 0; r < g.RY; ++r) {
        const int lo = ext_lo(r, g.H[l], g.RY, wn[2]), hi = ext_hi(r, g.H[l], g.RY, wn[3], lo);
        const int q = q_bound(r + 1, g.H[l], g.RY) - q_bound(r, g.H[l], g.RY);
        ph = hi - lo + 1 > ph ? hi - lo + 1 : ph;
        qh = q > qh ? q : qh;
      }
      slots += qw * qh;
      rows_pass = (kPassFirst[l] == l ? 0 : rows_pass) + pw * ph;
      rows_cap = rows_pass > rows_cap ? rows_pass : rows_cap;
    }
  }
  pl.slots_cap = slots;
  const int buffers = threads == 1024 ? 2 : 1;   // the persistent form double-buffers the windows
  g.queue_off = (unsigned)(buffers * rows_cap) * 64u;
  pl.lds = (size_t)buffers * rows_cap * 64 + (size_t)(threads / 4) * kQ * 16;
  pl.rc = 0;
  return pl;
}

template <class ET>
int launch4(hipStream_t st, const void* value, const int64_t* shapes, const void* packed, int64_t packed_stride,
            const float* vcounts, int64_t B, int64_t S, int M, int D, int L, int P, const signed char* win, int region_w,
            int region_h, int threads, int head_major, void* out) {
  if (!value || !shapes || !packed || !vcounts || !win || !out) return CODETR_E_BADARG;
  if (B <= 0 || S <= 0 || M <= 0 || L <= 0 || P <= 0) return CODETR_E_BADARG;
  if (D != 32) return CODETR_E_UNSUPPORTED;
  if (packed_stride < (int64_t)M * 64 || (packed_stride & 7) || packed_stride > 0x7fffffff ||
      (reinterpret_cast<uintptr_t>(packed) & 15) || (reinterpret_cast<uintptr_t>(value) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
    return CODETR_E_BADARG;
  Plan4 pl = plan4(shapes, S, M, L, P, win, region_w, region_h, threads);
  if (pl.rc != 0) return pl.rc;
  if (pl.slots_cap > (threads / 4) * (threads == 1024 ? kIt5 : kMaxIt) || pl.lds > (size_t)kMaxLds) return CODETR_E_UNSUPPORTED;
  if (S * M * (int64_t)64 > 0xffffffffLL || S * packed_stride * 2 > 0xffffffffLL) return CODETR_E_TOO_LARGE;   // 32-bit in-image offsets
  const int64_t blocks = B * pl.g.RX * pl.g.RY * M;
  if (blocks >= (1 << 22)) return CODETR_E_UNSUPPORTED;
  pl.g.vcounts = vcounts;
  pl.g.head_major = head_major ? 1 : 0;
  typedef void (*Kern)(const _Float16*, const unsigned short*, unsigned short*, const Geom4, const int);
  const Kern kern = threads == 512 ? msda_encoder_v4_kernel<ET, 512> : msda_encoder_v4_kernel<ET, 256>;
  const void* kfn = threads == 1024 ? reinterpret_cast<const void*>(msda_encoder_v5_kernel<ET>) : reinterpret_cast<const void*>(kern);
  {
    static std::atomic<uint32_t> done[64];   // > 64 KB of dynamic LDS: the attribute is per (device, function); one table per ET
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0, done[0].store(0);
    const uint32_t bit = threads == 1024 ? 4u : threads == 512 ? 2u : 1u;
    if (!(done[dev].load(std::memory_order_acquire) & bit)) {
      const hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
      if (e != hipSuccess) return (int)e;
      done[dev].fetch_or(bit, std::memory_order_release);
    }
  }
  if (threads == 1024) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
      cus = 256;
    const unsigned grid = (unsigned)(cus / 8 * 8);   // one persistent workgroup per CU, a multiple of the 8 XCDs
    hipLaunchKernelGGL(msda_encoder_v5_kernel<ET>, dim3(grid), dim3(1024), pl.lds, st, static_cast<const _Float16*>(value),
                       static_cast<const unsigned short*>(packed), static_cast<unsigned short*>(out), pl.g, (int)packed_stride,
                       (int)blocks);
  } else {
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3((unsigned)threads), pl.lds, st, static_cast<const _Float16*>(value),
                       static_cast<const unsigned short*>(packed), static_cast<unsigned short*>(out), pl.g, (int)packed_stride);
  }
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int codetr_msda_encoder_forward_packed_f16(void* stream, const void* value_dev, const int64_t* level_shapes_host,
                                           const void* packed_dev, int64_t packed_row_stride, const float* valid_counts_dev,
                                           int64_t B, int64_t S, int M, int D, int L, int P, const int8_t* windows_host,
                                           int region_w, int region_h, int threads, int value_head_major, void* out_dev) {
  return launch4<EF16>(static_cast<hipStream_t>(stream), value_dev, level_shapes_host, packed_dev, packed_row_stride,
                       valid_counts_dev, B, S, M, D, L, P, reinterpret_cast<const signed char*>(windows_host), region_w,
                       region_h, threads, value_head_major, out_dev);
}

int codetr_msda_encoder_forward_packed_bf16(void* stream, const void* value_f16_dev, const int64_t* level_shapes_host,
                                            const void* packed_dev, int64_t packed_row_stride, const float* valid_counts_dev,
                                            int64_t B, int64_t S, int M, int D, int L, int P, const int8_t* windows_host,
                                            int region_w, int region_h, int threads, int value_head_major, void* out_dev) {
  return launch4<EBF16>(static_cast<hipStream_t>(stream), value_f16_dev, level_shapes_host, packed_dev, packed_row_stride,
                        valid_counts_dev, B, S, M, D, L, P, reinterpret_cast<const signed char*>(windows_host), region_w,
                        region_h, threads, value_head_major, out_dev);
}

int64_t codetr_msda_encoder_packed_lds_bytes(const int64_t* level_shapes_host, int M, int L, int P, const int8_t* windows_host,
                                             int region_w, int region_h, int threads) {
  int64_t S = 0;
  if (!level_shapes_host || L != kL) return L == kL ? CODETR_E_BADARG : CODETR_E_UNSUPPORTED;
  for (int l = 0; l < L; ++l) S += level_shapes_host[2 * l] * level_shapes_host[2 * l + 1];
  const Plan4 pl = plan4(level_shapes_host, S, M, L, P, reinterpret_cast<const signed char*>(windows_host), region_w, region_h, threads);
  if (pl.rc != 0) return pl.rc;
  if (pl.slots_cap > (threads / 4) * (threads == 1024 ? kIt5 : kMaxIt)) return CODETR_E_UNSUPPORTED;
  return (int64_t)pl.lds;
}

int codetr_msda_pack_projection_index(int M, int L, int P, int32_t* idx_host) {
  if (!idx_host || M <= 0) return CODETR_E_BADARG;
  if (L != kL || P != kP) return CODETR_E_UNSUPPORTED;
  const int n_off = M * L * P * 2;
  for (int m = 0; m < M; ++m)
    for (int p = 0; p < P; ++p) {
      int32_t* d = idx_host + m * 64 + p * 16;
      for (int l = 0; l < L; ++l) {
        d[2 * l] = ((m * L + l) * P + p) * 2;
        d[2 * l + 1] = ((m * L + l) * P + p) * 2 + 1;
        d[10 + l] = n_off + (m * L + l) * P + p;
      }
      d[15] = -1;
    }
  return 0;
}

#ifdef MSDA4_STAMPS
int codetr_msda4_set_stamps(void* dev_ptr) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_msda4_stamps), &dev_ptr, sizeof(dev_ptr)) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
