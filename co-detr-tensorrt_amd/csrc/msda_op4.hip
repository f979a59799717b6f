// Windowed form of the PUBLIC multi-scale deformable attention op for MI355X (gfx950), round 6.
//
// The op: reference codetr/csrc/ms_deform_attn.cu:31-77 (bilinear sampler), 211-261 (im2col kernel), launched by :762-779
// behind torch.ops.codetr.multi_scale_deformable_attention (codetr/csrc/deformable_attention_torch.cpp:16-31):
//   out[b, q, m, :] = sum over (level l, point p) of  weight[b, q, m, l, p] * bilinear(value[b, level l, m, :], loc[b, q, m, l, p])
// with ready-made normalised sampling locations and softmax-ed weights.  The general kernel of csrc/msda_forward.hip
// (msda_tiled_kernel) gathers every corner row from global memory: 0.08 of the HBM roofline at the encoder's shape, unchanged
// for five rounds.  The model's own encoder path got its speed from csrc/msda_encoder4.hip, which needs
// the producers' cooperation (packed projection, head-major value map, windows from the offset bias).  This kernel carries
// that kernel's structure to the op's OWN operand layouts, for encoder-shaped calls (Nq == S, 5 levels x 4 points, 32-channel
// heads, fp16):
//   * a 512-thread workgroup serves one (16 x 16 region of the finest level, head): its queries are the pixels of the region
//     on all five levels (341 at most on a 2x pyramid); four lanes serve a (query, head) pair, lane p = point p on every level.
//   * per pass (levels {0}, {1, 2}, {3, 4}) the value rows the region's samples can reach -- the region's footprint on the level
//     grown by a margin, one extra zero-filled pixel where it touches the image border -- are staged in LDS with LDS-DMA
//     (16 pixels x the head's 64-byte slice per instruction); a sample whose four corners are staged is blended from LDS
//     (4 ds_read_b128, one unsigned range test per axis, no per-corner validity arithmetic); every other sample takes the
//     fix-up path -- the reference's full gate / corner logic, rows from global memory -- so ANY location is served exactly;
//     that query i sits at pixel i of the pyramid only decides how many samples hit their window.
//   * sampling locations and attention weights reach the lanes THROUGH LDS (north star: "sampling_locations / attention_
//     weights staged in LDS"): a pair's 80 + 40 bytes are fetched with one 16-byte and two 8-byte loads per lane (a quad reads
//     whole 16-byte pieces of the rows), written to a 144-byte record and read back transposed (lane p: its point's (x, y) on
//     the five levels, its five weights) -- 3 coalesced loads instead of the ten 2- / 4-byte ones the layout suggests.
//   * numerics as the general kernel: fp32 corner weights (bilinear x attention), every term added in fp32 with
//     v_fma_mix_f32 (the fp16 value read in place), one rounding at the end -- the op's tests hold it to one fp16 ulp of the
//     exactly rounded result (tests/test_msda_gpu.py), which the encoder kernel's packed-half chains do not meet.
//   * the pyramid's shapes are a DEVICE tensor: the grid is persistent (two workgroups per CU) and every workgroup derives
//     regions, windows and its tile list from the device-side shapes (msda_op4_plan.h); where the plan does not apply the
//     workgroups return at once and the general kernel, launched behind this one with the same plan as its skip test, serves
//     the call.
// Tested against the fp64 oracle and the reference's golden vectors through the op (tests/test_msda_gpu.py, unchanged, and
// tests/test_msda_op4_gpu.py).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "codetr_hip.h"
#include "msda_op4_plan.h"

namespace {

using namespace codetr_op4;

constexpr int kQ = 1;                         // fix-up records per (query, head) pair and queue round
constexpr int kWaves = kThreads / 64;
constexpr int kRecBytes = 144;                // operand record of a pair: 128 bytes used, 144 apart (bank spread)
constexpr unsigned kWinBytes = kWinPixels * 64u;
constexpr unsigned kQueueBytes = kPairs * kQ * 32u;
constexpr unsigned kLdsBytes = kWinBytes + kQueueBytes;   // 80 896 B: two workgroups per CU
static_assert(kWaves * 16 * kRecBytes <= (int)kWinBytes, "the operand records alias the front of the window area");
#ifndef MSDA_OP4_ABL
#define MSDA_OP4_ABL 0     // timing experiments only: 2: no staging, 4: no gather, 8: no preparation, 32: no output stores (WRONG results)
#endif
constexpr int kAbl = MSDA_OP4_ABL;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// floor(a / b) for 0 <= a < 2^22, 0 < b (one reciprocal + a fix-up)
__device__ __forceinline__ int fdiv(int a, int b) {
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
  const int r = a - q * b;
  q += r >= b ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned quad_bcast_u(unsigned v, int owner) {
  switch (owner) {
    case 0: return dpp_u<0x00>(v);
    case 1: return dpp_u<0x55>(v);
    case 2: return dpp_u<0xAA>(v);
    default: return dpp_u<0xFF>(v);
  }
}
__device__ __forceinline__ float quad_bcast_f(float v, int owner) { return __uint_as_float(quad_bcast_u(__float_as_uint(v), owner)); }
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
__device__ __forceinline__ unsigned pack_h2(float a, float b) {   // v_cvt_pk_f16_f32 (round to nearest even)
  const h2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
// acc += (float)(half of `u`) * w in ONE instruction each: the fp16 value is read in place
__device__ __forceinline__ void fma_h2(float& lo, float& hi, unsigned u, float w) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(u), "v"(w));
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi) : "v"(u), "v"(w));
}
__device__ __forceinline__ void lds_dma16(const unsigned char* src, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds_addr)
               : "memory", "m0");
}
__device__ __forceinline__ int floor_i(float v) {   // floor + float -> int in one instruction
  int d;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(d) : "v"(v));
  return d;
}

// What a wave keeps about one level (all wave-uniform: SGPRs)
struct Lv {
  int W, H, start;           // level size, first pixel inside S
  float fW, fH;
  int px0, py0;              // window origin, extended coordinates (-1 / size = the zero border)
  int xspan, yspan;          // largest (x0 - px0), (y0 - py0) whose four corners are staged
  int pw, ph;                // staged columns / rows
  unsigned base;             // LDS byte address of staged pixel (px0, py0)
  unsigned pitch;            // pw * 64
  int qx0, qy0, qw, slot0;   // the region's queries on this level: rectangle origin, width, first slot
};

struct TileId {
  int b, rx, ry, m;
};
constexpr int kBand = 4;
// tile -> (image, region column / row, head): heads innermost (the M slices of a 512-byte pixel row meet in one XCD's L2),
// regions in horizontal bands of kBand rows walked column by column (vertically adjacent regions share most of their windows)
__device__ __forceinline__ TileId decode_tile(int tile, int M, int RX, int RY) {
  TileId t;
  const int unit = fdiv(tile, M);
  t.m = tile - unit * M;
  const int regions = RX * RY;
  t.b = fdiv(unit, regions);
  const int reg = unit - t.b * regions;
  const int per_band = kBand * RX;
  const int band = fdiv(reg, per_band);
  const int r = reg - band * per_band;
  const int y0 = band * kBand;
  const int bh = min(kBand, RY - y0);
  t.rx = fdiv(r, bh);
  t.ry = y0 + (r - t.rx * bh);
  return t;
}

// one sample of the lane (level k, the lane's point): image coordinates -> (floor, fraction)
struct Smp {
  float lw, lh;
  int x0, y0;
};
__device__ __forceinline__ Smp sample_at(const Lv& v, unsigned o2) {
  Smp s;
  // cu:241-247: h_im = loc_h * H - 0.5, w_im = loc_w * W - 0.5 (x first in memory).  fmaxf: a NaN coordinate becomes a huge
  // negative one -> outside every window and outside the gate: the sample is dropped, as by the reference's comparisons
  const float w_im = fmaxf(fmaf((float)as_h2(o2)[0], v.fW, -0.5f), -3.0e38f);
  const float h_im = fmaxf(fmaf((float)as_h2(o2)[1], v.fH, -0.5f), -3.0e38f);
  s.lw = __builtin_amdgcn_fractf(w_im);
  s.lh = __builtin_amdgcn_fractf(h_im);
  s.x0 = floor_i(w_im);
  s.y0 = floor_i(h_im);
  return s;
}

// weight of level k out of the lane's five packed halves
__device__ __forceinline__ float weight_of(const unsigned (&awp)[3], int k) { return (float)as_h2(awp[k >> 1])[k & 1]; }

// prepared sample: LDS address of its (x0, y0) row, the four corner weights (zero when the sample is not served from LDS)
struct Prep {
  unsigned ad;
  float w00, w01, w10, w11;
};

// one level's sample of the lane: window test, LDS address, corner weights.  Returns false when the sample is not served
// from LDS (its weights are then zero and its address a finite staged row)
__device__ __forceinline__ bool prepare(Prep& pp, const Lv& v, const float a, const unsigned o2) {
  const Smp s = sample_at(v, o2);
  const unsigned dx = (unsigned)(s.x0 - v.px0), dy = (unsigned)(s.y0 - v.py0);
  const bool ok = dx <= (unsigned)v.xspan && dy <= (unsigned)v.yspan;   // all four corners are staged (or zero border)
  const float wy1 = s.lh * a, wy0 = a - wy1, wx0 = 1.f - s.lw;
  const unsigned a_in = __umul24(dy, v.pitch) + v.base + (dx << 6);
  pp.ad = ok ? a_in : v.base;
  pp.w00 = ok ? wy0 * wx0 : 0.f;
  pp.w01 = ok ? wy0 * s.lw : 0.f;
  pp.w10 = ok ? wy1 * wx0 : 0.f;
  pp.w11 = ok ? wy1 * s.lw : 0.f;
  return ok;
}

// gather of one iteration and one level: the quad's four points, one per step; the rows of step s + 1 are requested before
// the arithmetic of step s, the step's four weights broadcast when its rows are due (one level's Prep live at a time: the
// kernel runs at four waves per SIMD, 128 registers)
__device__ __forceinline__ void gather_level(float (&acc)[8], const Prep& pp, const unsigned pitch, const unsigned lds_lane) {
  using LV = const __attribute__((address_space(3))) u32x4*;
  // 16 rows (4 points x 4 corners), a ring of 5 registers sets: row k + 4 is requested right before row k is consumed (a
  // double buffer of whole steps needs 8 sets: the kernel runs at four waves per SIMD, 128 registers, and spilled)
  constexpr int RING = 5;
  u32x4 rows[RING];
  // DPP hazard (see msda_encoder4.hip): the inline-assembly DPP add reads an address register the instruction right before
  // may have written; one s_nop tied to it
  unsigned adr = pp.ad;
  asm volatile("s_nop 1" : "+v"(adr));
  unsigned a0[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) a0[o] = quad_bcast_add(adr, o, lds_lane);
  auto fetch = [&](int k) {   // row k = point k >> 2, corner k & 3
    const unsigned a = a0[k >> 2] + ((k & 2) ? pitch : 0u) + ((k & 1) ? 64u : 0u);
    rows[k % RING] = *(LV)(uintptr_t)a;
  };
#pragma unroll
  for (int k = 0; k < RING - 1; ++k) fetch(k);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k + RING - 1 < 16) fetch(k + RING - 1);
    const int o = k >> 2, cr = k & 3;
    const float w = quad_bcast_f(cr == 0 ? pp.w00 : cr == 1 ? pp.w01 : cr == 2 ? pp.w10 : pp.w11, o);
#pragma unroll
    for (int j = 0; j < 4; ++j) fma_h2(acc[2 * j], acc[2 * j + 1], rows[k % RING][j], w);
    if (cr == 3) __builtin_amdgcn_sched_barrier(0);
  }
}

// samples the windows do not serve: re-derived with the reference's gate / corner logic (cu:52-71, 249), queued per pair
// (32-byte records in LDS: first pixel, corner strides, four fp32 weights) and added from global memory by the pair's four
// lanes, kQ records per round
// One sample re-derived the reference's way: (is it served by the windows?, does it pass the gate?) and its record
struct FixRec {
  u32x4 ra, rb;
  bool ok, gate;
};
__device__ __forceinline__ FixRec fix_record(const Lv& v, const float a, const unsigned o2) {
  FixRec f;
  const Smp s = sample_at(v, o2);
  const unsigned dx = (unsigned)(s.x0 - v.px0), dy = (unsigned)(s.y0 - v.py0);
  f.ok = dx <= (unsigned)v.xspan && dy <= (unsigned)v.yspan;
  // cu:249: h_im > -1 && w_im > -1 && h_im < H && w_im < W  <=>  floor in [-1, size - 1]
  f.gate = (unsigned)(s.x0 + 1) <= (unsigned)v.W && (unsigned)(s.y0 + 1) <= (unsigned)v.H;
  const float wy0 = s.y0 >= 0 ? (1.f - s.lh) * a : 0.f, wy1 = s.y0 + 1 <= v.H - 1 ? s.lh * a : 0.f;   // cu:52-71
  const float wx0 = s.x0 >= 0 ? 1.f - s.lw : 0.f, wx1 = s.x0 + 1 <= v.W - 1 ? s.lw : 0.f;
  const int x0c = max(s.x0, 0), x1c = min(s.x0 + 1, v.W - 1), y0c = max(s.y0, 0), y1c = min(s.y0 + 1, v.H - 1);
  f.ra = u32x4{(unsigned)(v.start + y0c * v.W + x0c), (unsigned)((y1c - y0c) * v.W) | ((unsigned)(x1c - x0c) << 16), 0u, 0u};
  f.rb = u32x4{__float_as_uint(wy0 * wx0), __float_as_uint(wy0 * wx1), __float_as_uint(wy1 * wx0), __float_as_uint(wy1 * wx1)};
  return f;
}

// The rare path: one queued sample per pair and round.  Nothing but the pair's `bad` mask stays live between rounds -- the
// record of a sample is re-derived when its turn comes (the kernel runs at 128 registers; a path that kept the records and
// two rounds' rows cost the main path 27 spilled registers).
template <int LV0, int NLV>
__device__ __forceinline__ void fixup(float (&acc)[8], const Lv (&lv)[kL], const unsigned (&awp)[3], const unsigned (&o2)[kL],
                                      u32x4* __restrict__ queue, const unsigned char* __restrict__ vhead, const unsigned pix_bytes,
                                      const int sub) {
  unsigned bad = 0;
#pragma unroll
  for (int i = 0; i < NLV; ++i) {
    const FixRec f = fix_record(lv[LV0 + i], weight_of(awp, LV0 + i), o2[LV0 + i]);
    bad |= (!f.ok && f.gate) ? 1u << (sub + 4 * i) : 0u;
  }
  bad |= dpp_u<kXor2>(bad);
  bad |= dpp_u<kXor1>(bad);
  const int cnt = __builtin_popcount(bad);
  for (int base = 0; __builtin_amdgcn_ballot_w64(cnt > base) != 0; ++base) {
#pragma unroll
    for (int i = 0; i < NLV; ++i) {
      const int pt = sub + 4 * i;
      if (((bad >> pt) & 1u) && __builtin_popcount(bad & ((1u << pt) - 1u)) == base) {
        const FixRec f = fix_record(lv[LV0 + i], weight_of(awp, LV0 + i), o2[LV0 + i]);
        queue[0] = f.ra;
        queue[1] = f.rb;
      }
    }
    __builtin_amdgcn_wave_barrier();   // (the quad reads what one of its lanes just queued: one wave, LDS in order -- pins the compiler)
    if (base < cnt) {
      const u32x4 rc = queue[0], wts = queue[1];
      const unsigned char* p = vhead + (size_t)rc[0] * pix_bytes;
      const unsigned xs = (rc[1] >> 16) * pix_bytes, ys = (rc[1] & 0xffffu) * pix_bytes;
      const u32x4 r0 = *reinterpret_cast<const u32x4*>(p), r1 = *reinterpret_cast<const u32x4*>(p + xs);
      const u32x4 r2 = *reinterpret_cast<const u32x4*>(p + ys), r3 = *reinterpret_cast<const u32x4*>(p + ys + xs);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        fma_h2(acc[2 * jj], acc[2 * jj + 1], r0[jj], __uint_as_float(wts[0]));
        fma_h2(acc[2 * jj], acc[2 * jj + 1], r1[jj], __uint_as_float(wts[1]));
        fma_h2(acc[2 * jj], acc[2 * jj + 1], r2[jj], __uint_as_float(wts[2]));
        fma_h2(acc[2 * jj], acc[2 * jj + 1], r3[jj], __uint_as_float(wts[3]));
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// LDS: [staged rows of ONE pass (its front doubles as the operand records) | fix-up queues]
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void msda_op4_kernel(
    const _Float16* __restrict__ value, const int64_t* __restrict__ spatial_shapes, const int64_t* __restrict__ level_start,
    const unsigned short* __restrict__ loc, const unsigned short* __restrict__ weight, unsigned short* __restrict__ out,
    const int B, const int S, const int M) {
  constexpr unsigned kRow = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;

  Plan plan;
  if (!make_plan(spatial_shapes, level_start, S, (int64_t)B * M, plan)) return;   // (uniform for the whole grid: the general kernel serves the call)
  // the plan is wave-uniform, but its integer divisions run on the vector unit: back into scalar registers, or 22 vector
  // registers stay occupied for the whole kernel (measured: 55 spills at the 128-register budget)
#pragma unroll
  for (int l = 0; l < kL; ++l) {
    plan.W[l] = uni(plan.W[l]);
    plan.H[l] = uni(plan.H[l]);
    plan.start[l] = uni(plan.start[l]);
    plan.mg[l] = uni(plan.mg[l]);
  }
  plan.RX = uni(plan.RX);
  plan.RY = uni(plan.RY);

  const int ntiles = B * plan.RX * plan.RY * M;

  // persistent tile list: the workgroups of one XCD (blockIdx & 7 under round-robin placement -- speed only) walk one
  // contiguous eighth of the tiles, 64 consecutive tiles (8 regions x 8 heads) in flight per XCD
  const int per_xcd = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, chunk = (ntiles + 7) >> 3;
  const int t_end = min((xcd + 1) * chunk, ntiles);

  for (int tile = xcd * chunk + ((int)blockIdx.x >> 3); tile < t_end; tile += per_xcd) {
    // Everything below is re-derived per tile from OPAQUE copies of the thread id and of the plan: left visible, the
    // compiler hoists two dozen loop-invariant values (reciprocals of the divisions, per-lane masks and offsets, float
    // copies of the level sizes) out of this loop and keeps them live across it -- 55 spills at the 128-register budget.
    int tid = threadIdx.x, Mo = M;
    asm volatile("" : "+v"(tid), "+s"(Mo));
#pragma unroll
    for (int l = 0; l < kL; ++l) asm volatile("" : "+s"(plan.W[l]), "+s"(plan.H[l]), "+s"(plan.start[l]), "+s"(plan.mg[l]));
    asm volatile("" : "+s"(plan.RX), "+s"(plan.RY));
    const int M = Mo, RX = plan.RX, RY = plan.RY;
    const int wave = uni(tid >> 6), lane = tid & 63, sub = lane & 3, pl = lane >> 2;
    const unsigned pix_bytes = (unsigned)M * kRow;   // the op's value layout [B, S, M, 32]: a head's slices are M * 64 B apart
    u32x4* const queue = reinterpret_cast<u32x4*>(smem + kWinBytes) + (size_t)(wave * 16 + pl) * kQ * 2;
    const unsigned lds_lane = (unsigned)sub * 16;
    unsigned char* const rec = smem + (wave * 16 + pl) * kRecBytes;
    // ---- wave-uniform geometry -> scalar registers ----
    const TileId t0 = decode_tile(tile, M, RX, RY);
    const TileId t = {uni(t0.b), uni(t0.rx), uni(t0.ry), uni(t0.m)};
    Lv lv[kL];
    int total;
    {
      // 40 divisions (per level: two window bounds and two query bounds per axis), one per LANE: lane 8 l + j computes item
      // j of level l -- 0: first staged column, 1: last staged column (before the "at least two columns" clamp), 2 / 3: rows,
      // 4 / 5: first query column of this region / of the next, 6 / 7: rows -- and the results return to scalar registers
      // with v_readlane
      const int gl = lane >> 3 > kL - 1 ? kL - 1 : lane >> 3, gj = lane & 7;
      int nW = plan.W[0], nH = plan.H[0], mgl = plan.mg[0];
#pragma unroll
      for (int l = 1; l < kL; ++l) {
        nW = gl == l ? plan.W[l] : nW;
        nH = gl == l ? plan.H[l] : nH;
        mgl = gl == l ? plan.mg[l] : mgl;
      }
      const bool ay = (gj & 2) != 0, isq = gj >= 4, plus = (gj & 1) != 0;
      const int n = ay ? nH : nW, R = ay ? RY : RX, rr = (ay ? t.ry : t.rx) + (plus ? 1 : 0);
      const int wv = plus ? mgl : -mgl;
      const int num = 2 * rr * n + (isq ? R - 1 : 2 * wv * R - R);
      const int numd = (!isq && plus) ? num + 2 * R - 1 : num;
      const int q = fdiv(numd > 0 ? numd : 0, 2 * R);
      const int res = isq ? q : plus ? (num <= 0 ? 0 : (q > n ? n : q)) : (num < 0 ? -1 : (q > n - 1 ? n - 1 : q));
      int slot = 0;
      unsigned base = lds0;
#pragma unroll
      for (int l = 0; l < kL; ++l) {
        Lv& v = lv[l];
        v.W = plan.W[l];
        v.H = plan.H[l];
        v.fW = (float)v.W;
        v.fH = (float)v.H;
        v.start = plan.start[l];
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
        if (pass_first(l) == l) base = lds0;       // a new pass starts at the front of the buffer
        v.base = base;
        base += (unsigned)(v.pw * v.ph) * kRow;
        v.qx0 = __builtin_amdgcn_readlane(res, 8 * l + 4);
        v.qy0 = __builtin_amdgcn_readlane(res, 8 * l + 6);
        v.qw = __builtin_amdgcn_readlane(res, 8 * l + 5) - v.qx0;
        const int qh = __builtin_amdgcn_readlane(res, 8 * l + 7) - v.qy0;
        v.slot0 = slot;
        slot += v.qw * qh;
      }
      total = slot;
    }
    const int n_it = total > wave * 16 ? (total - wave * 16 + kPairs - 1) / kPairs : 0;   // <= kMaxIt (the plan's slot bound)

    const unsigned char* vhead0 = reinterpret_cast<const unsigned char*>(value) + ((size_t)t.b * S * M + t.m) * kRow;   // (uniform)
    const unsigned char* vhead = vhead0 + sub * 16;
    // the image's (query, head) pairs: pair (q, m) = q * M + m; 80 bytes of locations, 40 of weights
    const unsigned char* loc_b = reinterpret_cast<const unsigned char*>(loc) + ((size_t)t.b * S * M + t.m) * 80;
    const unsigned char* w_b = reinterpret_cast<const unsigned char*>(weight) + ((size_t)t.b * S * M + t.m) * 40;

    // ---- the wave's queries: slot -> (level, pixel) -> flattened index; raw operands requested ----
    int tS0 = 0, tA = 0, tB = 0, tSt = 0;   // per-level tables, level l in LANE l (looked up with ds_bpermute)
#pragma unroll
    for (int l = 0; l < kL; ++l) {
      const bool me = lane == l;
      tS0 = me ? lv[l].slot0 : tS0;
      tA = me ? (lv[l].qx0 | (lv[l].qy0 << 16)) : tA;
      tB = me ? (lv[l].W | (lv[l].qw << 16)) : tB;
      tSt = me ? lv[l].start : tSt;
    }
    int qs[kMaxIt];
    u32x4 rawA[kMaxIt];
    u32x2 rawB[kMaxIt], rawC[kMaxIt];
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
      qs[it] = 0;
      rawA[it] = u32x4{0u, 0u, 0u, 0u};
      rawB[it] = rawC[it] = u32x2{0u, 0u};
      if (it < n_it) {
        int sl = (it * kWaves + wave) * 16 + pl;
        sl = sl < total ? sl : total - 1;
        int lvq = 0;
#pragma unroll
        for (int l = 1; l < kL; ++l) lvq += sl >= lv[l].slot0 ? 4 : 0;     // byte address of the level's lane
        const int s0 = __builtin_amdgcn_ds_bpermute(lvq, tS0), cA = __builtin_amdgcn_ds_bpermute(lvq, tA);
        const int cB = __builtin_amdgcn_ds_bpermute(lvq, tB), st = __builtin_amdgcn_ds_bpermute(lvq, tSt);
        const int tq = sl - s0, qw = cB >> 16, W = cB & 0xffff;
        const int yy = (int)(((float)tq + 0.5f) * __builtin_amdgcn_rcpf((float)qw));
        const int y = (cA >> 16) + yy, x = (cA & 0xffff) + (tq - yy * qw);
        qs[it] = st + y * W + x;
        // a quad fetches whole 16-byte pieces of the pair's rows: lane p the four (x, y) of level p; then lane 0 level 4, lanes
        // 1 and 2 weights 0-7 / 8-15, lane 3 weights 16-19 (its second 8-byte load repeats the first: nothing past the row)
        const unsigned pair = (unsigned)qs[it] * (unsigned)M;
        const unsigned char* lp = loc_b + (size_t)pair * 80;
        const unsigned char* wp = w_b + (size_t)pair * 40;
        rawA[it] = *reinterpret_cast<const u32x4*>(lp + sub * 16);
        const unsigned char* pb = sub == 0 ? lp + 64 : wp + (sub - 1) * 16;
        rawB[it] = *reinterpret_cast<const u32x2*>(pb);
        rawC[it] = *reinterpret_cast<const u32x2*>(pb + (sub == 3 ? 0 : 8));
      }
    }

    // every wave is done with the previous tile's staged rows and queue before the records overwrite the window area
    __syncthreads();
    // ---- through LDS: the pair's record [(x, y) of level 0-4: 80 B | weights: 40 B], read back transposed ----
    float acc[kMaxIt][8];
    unsigned o2[kMaxIt][kL], awp[kMaxIt][3];   // the five weights stay packed halves (three registers per iteration)
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[it][j] = 0.f;
#pragma unroll
      for (int k = 0; k < kL; ++k) o2[it][k] = 0u;
      awp[it][0] = awp[it][1] = awp[it][2] = 0u;
      if (it < n_it) {
        *reinterpret_cast<u32x4*>(rec + sub * 16) = rawA[it];
        *reinterpret_cast<u32x2*>(rec + 64 + sub * 16) = rawB[it];
        *reinterpret_cast<u32x2*>(rec + 72 + sub * 16) = rawC[it];
        __builtin_amdgcn_wave_barrier();   // (the LDS serves one wave's operations in order; this only pins the compiler)
#pragma unroll
        for (int k = 0; k < kL; ++k) o2[it][k] = *reinterpret_cast<const unsigned*>(rec + k * 16 + sub * 4);
        unsigned short wh[kL];
#pragma unroll
        for (int k = 0; k < kL; ++k) wh[k] = *reinterpret_cast<const unsigned short*>(rec + 80 + (k * 4 + sub) * 2);
        awp[it][0] = (unsigned)wh[0] | ((unsigned)wh[1] << 16);
        awp[it][1] = (unsigned)wh[2] | ((unsigned)wh[3] << 16);
        awp[it][2] = (unsigned)wh[4];
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();   // the records are consumed: the first pass may stage over them

    auto run_pass = [&](auto lv0_c, auto nlv_c) {
      constexpr int LV0 = decltype(lv0_c)::value, NLV = decltype(nlv_c)::value;
      if (LV0 > 0) __syncthreads();   // every wave is done reading the previous pass's rows
      // -- the pass's windows -> LDS, ROW-WISE: a wave takes window rows y = wave, wave + 8, ...; one LDS-DMA instruction moves
      // 16 pixels of the row (4 lanes x 16 B per pixel = its 64-byte head slice) from a wave-uniform base + a per-lane constant
      // offset; cells outside the image -- the zero border -- are written by ds_write instead
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
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int it = 0; it < kMaxIt; ++it)
        if (it < n_it && !(kAbl & 4)) {
          bool clean = true;
#pragma unroll
          for (int i = 0; i < NLV; ++i) {
            const Lv& v = lv[LV0 + i];
            Prep pp = Prep{v.base, 0.f, 0.f, 0.f, 0.f};
            if (!(kAbl & 8)) clean = prepare(pp, v, weight_of(awp[it], LV0 + i), o2[it][LV0 + i]) && clean;
            gather_level(acc[it], pp, v.pitch, lds_lane);
          }
          clean = __builtin_amdgcn_ballot_w64(!clean) == 0;
          if (!clean) fixup<LV0, NLV>(acc[it], lv, awp[it], o2[it], queue, vhead, pix_bytes, sub);
        }
    };
    run_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    run_pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
    run_pass(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});

    unsigned char* orow = reinterpret_cast<unsigned char*>(out) + ((size_t)t.b * S * M + t.m) * kRow + sub * 16;
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it)
      if (it < n_it && (it * kWaves + wave) * 16 + pl < total) {
        const float (&a)[8] = acc[it];
        const u32x4 o = {pack_h2(a[0], a[1]), pack_h2(a[2], a[3]), pack_h2(a[4], a[5]), pack_h2(a[6], a[7])};
        if (!(kAbl & 32) || o[0] == 0x12345678u)
          *reinterpret_cast<u32x4*>(orow + (size_t)((unsigned)qs[it] * ((unsigned)M * kRow))) = o;
      }
  }
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

}  // namespace

extern "C" {

int codetr_msda_op4_supported(int elem_bytes, int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P) {
  // the host-visible part of the test; the device-side plan (msda_op4_plan.h) decides on the pyramid itself
  if (elem_bytes != 2 || D != 32 || L != kL || P != kP || Nq != S || M <= 0 || B <= 0) return 0;
  if (S < 4096) return 0;                                              // a launch-bound call: the general kernel alone
  if (S * M * (int64_t)80 > 0xffffffffLL || B * S * M >= (int64_t)1 << 31) return 0;   // 32-bit in-image offsets
  return 1;
}

// fp16 only.  Launches the windowed kernel; the caller launches the general kernel behind it with its skip test enabled.
int codetr_msda_op4_forward_f16(void* stream, const void* value_dev, const int64_t* spatial_shapes_dev, const int64_t* level_start_dev,
                                const void* loc_dev, const void* weight_dev, int64_t B, int64_t S, int M, int D, int L, int64_t Nq,
                                int P, void* out_dev) {
  if (!value_dev || !spatial_shapes_dev || !level_start_dev || !loc_dev || !weight_dev || !out_dev) return CODETR_E_BADARG;
  if (!codetr_msda_op4_supported(2, B, S, M, D, L, Nq, P)) return CODETR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(value_dev) | reinterpret_cast<uintptr_t>(loc_dev) | reinterpret_cast<uintptr_t>(out_dev)) & 15)
    return CODETR_E_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(weight_dev) & 7) return CODETR_E_UNSUPPORTED;
  {
    static std::atomic<uint32_t> done[64];   // > 64 KB of dynamic LDS: the attribute is per (device, function)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0, done[0].store(0);
    if (!done[dev].load(std::memory_order_acquire)) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_op4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)kLdsBytes);
      if (e != hipSuccess) return (int)e;
      done[dev].store(1, std::memory_order_release);
    }
  }
  const int grid = device_cus() / 8 * 8 * 2;   // two workgroups per CU, a multiple of 8
  hipLaunchKernelGGL(msda_op4_kernel, dim3((unsigned)grid), dim3(kThreads), kLdsBytes, static_cast<hipStream_t>(stream),
                     static_cast<const _Float16*>(value_dev), spatial_shapes_dev, level_start_dev,
                     static_cast<const unsigned short*>(loc_dev), static_cast<const unsigned short*>(weight_dev),
                     static_cast<unsigned short*>(out_dev), (int)B, (int)S, M);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // extern "C"
