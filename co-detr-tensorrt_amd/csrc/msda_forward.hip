// Multi-scale deformable attention forward for MI355X (gfx950 / CDNA4).
//
// Replaces the reference's ms_deformable_im2col_gpu_kernel + launcher + ATen wrapper
// (reference codetr/csrc/ms_deform_attn.cu:31-77, 211-261, 762-779, 899-973) behind the C ABI
// declared in include/codetr_hip.h.  Written for wave64 / LDS / the XCD-partitioned L2, not
// translated from the reference's one-thread-per-output-scalar CUDA kernel.
//
// Shape of the work.  One "pair" = one (batch, query, head): it needs L*P sample points, each
// a bilinear blend of 4 value rows of D channels (D*sizeof(T) contiguous bytes = 64 B for the
// model's D=32 fp16).  The op is a gather: ~80 scattered 64-B reads per 64-B output row, no
// reuse inside a pair, heavy reuse ACROSS neighbouring queries -> bound by the vector-memory
// path (L1/L2), not by HBM and not by math.
//
// Tiled kernel (the model path):
//   * a 256-thread workgroup owns PAIRS = 256/LANES consecutive pairs; a pair is served by
//     LANES = D*sizeof(T)/16 adjacent lanes, each lane owning 16 B (8 fp16 / 4 fp32 channels)
//     of every value row -> every global_load_dwordx4 of a wave covers 64/LANES complete,
//     contiguous 64-B value rows (coalesced per row, no partial sectors).
//   * phase 1 (once per workgroup): each lane converts a few (pair, point) entries from
//     (x, y, w) to {4 byte-offsets, 4 fp32 weights} = 32 B and parks them in LDS.  Level
//     shapes are read ON DEVICE from the int64 tensors with scalar loads (uniform level loop),
//     as the reference does (cu:236-239) -- no host sync.  Out-of-image corners keep a
//     clamped, always-valid address and weight 0, so phase 2 is branch-free.
//   * phase 2: per point 2 x ds_read_b128 (broadcast inside the pair's lanes, conflict-free
//     across pairs), 4 x global_load_dwordx4 (SGPR base + 32-bit VGPR offset) and
//     4*VEC v_fma_mix; fp32 accumulators, one rounding at the 16-B coalesced store.
//   * blockIdx is remapped so that each XCD (blocks b, b+8, ... share one) walks its own
//     contiguous eighth of the query range: neighbouring queries sample neighbouring value
//     rows, so each row is pulled into ONE XCD's L2 instead of all eight.
//
// Scalar kernel: any (M, D, L, P) and fp64 -- one thread per output element in the tensor's
// own arithmetic type; used for odd channel counts and by the fp64 parity tests.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "codetr_hip.h"
#include "msda_op4_plan.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxTiledLdsBytes = 64 * 1024;

// Tuning knobs (A/B-measured on MI355X, see DESIGN.md "MSDA kernel"): sample points whose row
// loads are issued back-to-back, and the occupancy the register allocator is told to aim for.
// 40 KiB of LDS per workgroup already caps the model shape at 4 workgroups = 4 waves/SIMD.
#ifndef MSDA_GROUP
#define MSDA_GROUP 4
#endif
#ifndef MSDA_WAVES_PER_EU
#define MSDA_WAVES_PER_EU 4
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct F16 {
  using storage = _Float16;
  using vec = f16x8;  // 16 B
  static constexpr int VEC = 8;
  __device__ static float to_f32(storage v) { return (float)v; }
  __device__ static storage from_f32(float v) { return (_Float16)v; }
};
struct BF16 {
  using storage = unsigned short;
  using vec = u16x8;  // 16 B
  static constexpr int VEC = 8;
  __device__ static float to_f32(storage v) { return __uint_as_float(((unsigned)v) << 16); }
  __device__ static storage from_f32(float v) {
    // round-to-nearest-even; NaN stays NaN (quiet)
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
};
struct F32 {
  using storage = float;
  using vec = f32x4;  // 16 B
  static constexpr int VEC = 4;
  __device__ static float to_f32(storage v) { return v; }
  __device__ static storage from_f32(float v) { return v; }
};

// ------------------------------------------------------------------------------------------
// XCD-aware tile order: hardware deals workgroups round-robin over the 8 XCDs, so blocks with
// equal (blockIdx % 8) share an L2.  Give each such group one contiguous run of tiles.
// Bijective for any grid size (cdna_hip_programming.md T1).  Placement only affects speed.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned xcd_tile(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
  const unsigned first = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return first + i;
}

// One LDS entry = 32 B: byte offsets of the 4 corners (clamped, always dereferenceable) and
// their fp32 weights (bilinear * attention, 0 for out-of-image corners / gated points).
struct __attribute__((aligned(16))) Entry {
  u32x4 off;
  f32x4 w;
};

// Extra inputs of the FUSED variant: instead of ready-made sampling locations and
// softmax-ed weights the kernel takes what the two projections produce and the reference points, and
// performs reference multi_scale_deformable_attention.py:180-196 itself, in fp32:
//   weights = softmax over the L*P logits of a (query, head);
//   loc     = ref_xy + off / (W_l, H_l)                  (2-d reference points)
//           = ref_xy + off / P * ref_wh * 0.5            (4-d reference points)
// `loc` then points at the offsets and `weight` at the logits; both are addressed as rows of one
// (possibly wider, fused-projection) matrix with the strides below.
struct FusedArgs {
  const void* ref;      // [B*Nq, L, ref_dim]: the tensors' 16-bit type, or fp32 (ref_f32)
  int ref_dim;          // 2 or 4
  int off_stride;       // elements between consecutive (b, q) rows of the offsets matrix
  int logit_stride;     // same for the logits matrix
  int ref_f32;          // reference points are fp32 (a coordinate in [0.5, 1) resolves to 1/2048 in fp16: a quarter
                        // pixel on a 480-wide level)
};

constexpr int kMaxLevels = 16;  // level table kept in LDS

// Phase 1 of both gather kernels, shared.  The PL lanes that serve a pair in phase 2 also prepare that pair's
// L*P sample points: lane `sub` takes points sub, sub+PL, ... (at most KMAX of them).  ALL global loads of a lane
// (locations / offsets, weights / logits, reference points) are issued before the first use, so the prologue
// exposes one memory latency, not one per level (a level-by-level loop measured 239 us of a 680 us launch).
// Softmax over the pair's logits (FUSED) is two shuffle reductions inside the PL-lane group.  Every point becomes
// a 32-byte Entry in LDS: byte offsets of its 4 corners (clamped: always dereferenceable) + 4 fp32 weights
// (bilinear x attention; 0 for out-of-image corners and gated points), so phase 2 is branch-free.
//   pair_base  byte offset of (image b, head m) inside the value tensor, without the level start
//   row_bytes  bytes between horizontally adjacent pixels of one head
template <class TR, int PL, int KMAX, bool FUSED>
__device__ __forceinline__ void build_entries(Entry* __restrict__ entries, int PAIRS, const int* __restrict__ s_meta,
                                              int pl, int sub, unsigned g, int M, int L, int P, unsigned pair_base,
                                              unsigned row_bytes, const typename TR::storage* __restrict__ loc,
                                              const typename TR::storage* __restrict__ weight, const FusedArgs& fa) {
  using S = typename TR::storage;
  struct __attribute__((aligned(2 * sizeof(S)))) S2 { S a, b; };
  const int LP = L * P;
  const unsigned m = g % (unsigned)M, row = g / (unsigned)M;
  float px[KMAX], py[KMAX], pw[KMAX], rw[KMAX], rh[KMAX];
  int lvl[KMAX];
  // ---- all loads first ----
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int pt = sub + k * PL;
    lvl[k] = 0;
    px[k] = py[k] = 0.f;
    pw[k] = FUSED ? -INFINITY : 0.f;
    rw[k] = rh[k] = 0.f;
    if (pt < LP) {
      const int l = pt / P;
      lvl[k] = l;
      if (FUSED) {
        const int col = (int)m * LP + pt;
        const S2 o2 = *reinterpret_cast<const S2*>(loc + (size_t)row * fa.off_stride + 2 * col);
        px[k] = TR::to_f32(o2.a);
        py[k] = TR::to_f32(o2.b);
        pw[k] = TR::to_f32(weight[(size_t)row * fa.logit_stride + col]);
        if (fa.ref_f32) {
          const float* rp = static_cast<const float*>(fa.ref) + ((size_t)row * L + l) * fa.ref_dim;
          const float2 r2 = *reinterpret_cast<const float2*>(rp);
          rw[k] = r2.x;
          rh[k] = r2.y;
          if (fa.ref_dim == 4) {
            const float2 wh = *reinterpret_cast<const float2*>(rp + 2);
            px[k] *= wh.x * (0.5f / (float)P);
            py[k] *= wh.y * (0.5f / (float)P);
          }
        } else {
          const S* rp = static_cast<const S*>(fa.ref) + ((size_t)row * L + l) * fa.ref_dim;
          const S2 r2 = *reinterpret_cast<const S2*>(rp);
          rw[k] = TR::to_f32(r2.a);
          rh[k] = TR::to_f32(r2.b);
          if (fa.ref_dim == 4) {  // fold (w, h) into the offsets now: off / P * wh * 0.5
            const S2 wh = *reinterpret_cast<const S2*>(rp + 2);
            px[k] *= TR::to_f32(wh.a) * (0.5f / (float)P);
            py[k] *= TR::to_f32(wh.b) * (0.5f / (float)P);
          }
        }
      } else {
        const size_t e = (size_t)g * LP + pt;
        const S2 l2 = *reinterpret_cast<const S2*>(loc + 2 * e);
        px[k] = TR::to_f32(l2.a);
        py[k] = TR::to_f32(l2.b);
        pw[k] = TR::to_f32(weight[e]);
      }
    }
  }
  // ---- softmax over the pair's logits ----
  if (FUSED) {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) mx = fmaxf(mx, pw[k]);
#pragma unroll
    for (int o = PL / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      pw[k] = __expf(pw[k] - mx);  // exp(-inf) = 0 for the unused slots
      sum += pw[k];
    }
#pragma unroll
    for (int o = PL / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) pw[k] *= inv;
  }
  // ---- entries ----
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int pt = sub + k * PL;
    if (pt >= LP) continue;
    const int H = s_meta[4 * lvl[k]], W = s_meta[4 * lvl[k] + 1];
    const unsigned start = (unsigned)s_meta[4 * lvl[k] + 2];
    const float Hf = (float)H, Wf = (float)W;
    float x = px[k], y = py[k];
    if (FUSED) {
      if (fa.ref_dim == 2) {
        x = fmaf(x, 1.0f / Wf, rw[k]);
        y = fmaf(y, 1.0f / Hf, rh[k]);
      } else {
        x += rw[k];
        y += rh[k];
      }
    }
    const float aw = pw[k];
    // pixel coordinates (reference cu:246-247), fp32 regardless of T
    const float h_im = fmaf(y, Hf, -0.5f);
    const float w_im = fmaf(x, Wf, -0.5f);
    const bool gate = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;  // cu:249
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h0 = (int)hf, w0 = (int)wf;
    const float lh = h_im - hf, lw = w_im - wf;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const bool h0ok = h0 >= 0, w0ok = w0 >= 0, h1ok = h0 + 1 <= H - 1, w1ok = w0 + 1 <= W - 1;  // cu:52-71
    const float g_aw = gate ? aw : 0.f;
    Entry en;
    en.w[0] = (h0ok && w0ok) ? hh * hw * g_aw : 0.f;
    en.w[1] = (h0ok && w1ok) ? hh * lw * g_aw : 0.f;
    en.w[2] = (h1ok && w0ok) ? lh * hw * g_aw : 0.f;
    en.w[3] = (h1ok && w1ok) ? lh * lw * g_aw : 0.f;
    // clamped coordinates keep every address inside level l of image b
    const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h0 + 1, 0), H - 1);
    const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w0 + 1, 0), W - 1);
    const unsigned base = pair_base + start * row_bytes;
    en.off[0] = base + (unsigned)(h0c * W + w0c) * row_bytes;
    en.off[1] = base + (unsigned)(h0c * W + w1c) * row_bytes;
    en.off[2] = base + (unsigned)(h1c * W + w0c) * row_bytes;
    en.off[3] = base + (unsigned)(h1c * W + w1c) * row_bytes;
    entries[pt * PAIRS + pl] = en;
  }
}

// level table (H, W, start, -) in LDS; spatial_shapes / level_start stay on device (reference cu:236-239)
__device__ __forceinline__ void load_level_table(int* s_meta, const int64_t* __restrict__ spatial_shapes,
                                                 const int64_t* __restrict__ level_start, int L) {
  if ((int)threadIdx.x < L) {
    s_meta[4 * threadIdx.x] = (int)spatial_shapes[2 * threadIdx.x];
    s_meta[4 * threadIdx.x + 1] = (int)spatial_shapes[2 * threadIdx.x + 1];
    s_meta[4 * threadIdx.x + 2] = (int)level_start[threadIdx.x];
    s_meta[4 * threadIdx.x + 3] = 0;
  }
  __syncthreads();
}

template <class TR, int LANES, bool FUSED>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(MSDA_WAVES_PER_EU, MSDA_WAVES_PER_EU))) void msda_tiled_kernel(
    const typename TR::storage* __restrict__ value, const int64_t* __restrict__ spatial_shapes,
    const int64_t* __restrict__ level_start, const typename TR::storage* __restrict__ loc,
    const typename TR::storage* __restrict__ weight, typename TR::storage* __restrict__ out,
    unsigned n_pairs /* B*Nq*M */, unsigned pairs_per_image /* Nq*M */, unsigned image_bytes /* S*M*D*sizeof */,
    int M, int L, int P, FusedArgs fa, int skip_S, int skip_BM) {
  using S = typename TR::storage;
  // skip_S != 0: the windowed kernel (csrc/msda_op4.hip) was launched in front of this one for the same call; where its
  // device-side plan applies it has done the work (the pyramid's shapes are a device tensor: the host cannot tell)
  if (skip_S != 0) {
    codetr_op4::Plan plan;
    if (codetr_op4::make_plan(spatial_shapes, level_start, skip_S, skip_BM, plan, false)) return;   // (the verdict alone)
  }
  constexpr int VEC = TR::VEC;
  constexpr int D = VEC * LANES;
  constexpr int PAIRS = kThreads / LANES;  // pairs per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Entry* entries = reinterpret_cast<Entry*>(smem_raw);  // [L*P][PAIRS]

  const int LP = L * P;
  const unsigned row_bytes = (unsigned)(M * D * sizeof(S));  // one pixel, all heads
  int* s_meta = reinterpret_cast<int*>(entries + (size_t)LP * PAIRS);
  load_level_table(s_meta, spatial_shapes, level_start, L);
  const int pl = threadIdx.x / LANES;
  const int sub = threadIdx.x % LANES;
  // One tile per workgroup in the ordinary launch (grid = tiles).  Behind the windowed kernel the grid is capped (a skipped
  // launch of 25 000 workgroups costs ~30 us of dispatch alone) and a workgroup walks tiles blockIdx, blockIdx + grid, ...
  // (measured: with the ordinary launch routed through the stride loop as well the decoder-shaped call went from 12 to
  // 30 us -- the body is therefore instantiated twice, once straight-line and once inside the loop)
  auto do_tile = [&](const unsigned tile) {
  const unsigned pair0 = tile * PAIRS;

  // ---------------- phase 1: sample points -> {corner offsets, weights} in LDS ----------------
  {
    unsigned g = pair0 + pl;
    g = g < n_pairs ? g : n_pairs - 1;  // tail: recompute the last pair, its store is masked
    const unsigned b = g / pairs_per_image, m = g % (unsigned)M;
    build_entries<TR, LANES, 8, FUSED>(entries, PAIRS, s_meta, pl, sub, g, M, L, P,
                                       b * image_bytes + m * (unsigned)(D * sizeof(S)), row_bytes, loc, weight, fa);
  }
  __syncthreads();

  // ---------------- phase 2: gather + blend ----------------
  const unsigned lane_byte = (unsigned)(sub * 16);
  const unsigned char* vbase = reinterpret_cast<const unsigned char*>(value);
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;

  // Points are consumed in groups of GROUP: all 4*GROUP row loads of a group are issued
  // before the first FMA so that each wave keeps 16 x 1 KiB of gathers in flight.
  using V = typename TR::vec;
  constexpr int GROUP = MSDA_GROUP;
  const Entry* my = entries + pl;
  int i = 0;
#if defined(MSDA_ABLATE) && MSDA_ABLATE == 1  // timing experiment only: no gather phase
  i = LP;
#endif
  for (; i + GROUP <= LP; i += GROUP) {
    Entry en[GROUP];
    V raw[GROUP][4];
#pragma unroll
    for (int g = 0; g < GROUP; ++g) en[g] = my[(i + g) * PAIRS];
#pragma unroll
    for (int g = 0; g < GROUP; ++g)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#if defined(MSDA_ABLATE) && MSDA_ABLATE == 2  // timing experiment only: every lane re-reads one hot line
        raw[g][k] = *reinterpret_cast<const V*>(vbase + (size_t)((en[g].off[k] & 0) + lane_byte));
#else
        raw[g][k] = *reinterpret_cast<const V*>(vbase + (size_t)(en[g].off[k] + lane_byte));
#endif
#pragma unroll
    for (int g = 0; g < GROUP; ++g)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float wk = en[g].w[k];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = __builtin_fmaf(wk, TR::to_f32(raw[g][k][j]), acc[j]);
      }
  }
  for (; i < LP; ++i) {
    const Entry en = my[i * PAIRS];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const V raw = *reinterpret_cast<const V*>(vbase + (size_t)(en.off[k] + lane_byte));
      const float wk = en.w[k];
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = __builtin_fmaf(wk, TR::to_f32(raw[j]), acc[j]);
    }
  }

  const unsigned g = pair0 + pl;
  if (g < n_pairs) {
    V packed;
#pragma unroll
    for (int j = 0; j < VEC; ++j) packed[j] = TR::from_f32(acc[j]);
    *reinterpret_cast<V*>(reinterpret_cast<unsigned char*>(out) + ((size_t)g * D * sizeof(S) + lane_byte)) = packed;
  }
  };
  if (skip_S == 0) {
    do_tile(xcd_tile(blockIdx.x, gridDim.x));
  } else {
    const unsigned n_tiles = (n_pairs + PAIRS - 1) / PAIRS;
    for (unsigned tile = xcd_tile(blockIdx.x, gridDim.x); tile < n_tiles; tile += gridDim.x) {
      do_tile(tile);
      __syncthreads();   // the entries are rewritten for the next tile
    }
  }
}

// Measured and rejected for this kernel: a "head-split" pair mapping (each workgroup 2 heads x 32 consecutive queries
// instead of 8 heads x 8 queries, so that the lines of one corner load are 8 neighbouring pixels of the same heads):
// 652 us vs 614 us on the +-3 px synthetic encoder case -- the per-pair prologue loads lose their contiguity and the
// L1 reuse gained inside a wave does not pay for it.
// ------------------------------------------------------------------------------------------
// Head-major fused variant.  Value map laid out [B, M, S, D] (each head's map contiguous, written that
// way by the value projection's epilogue): the two horizontal neighbours (x0, x1) of a sample are then
// 2*D*sizeof(T) = 128 contiguous bytes = one whole cache line, where the reference layout [B, S, M, D]
// puts them 512 B apart and every 64-B corner read uses half a line.  A pair is served by 2*LANES lanes:
// the low LANES lanes take the x0 column, the high LANES lanes the x1 column, so a wave instruction
// covers 8 pairs x 128 B (8 full lines instead of 16 half lines) and a point needs two loads (rows y0,
// y1) per lane instead of four; the two column partial sums meet in one shuffle at the end.
// Clamping / zero weights per corner as in the tiled kernel, so x0 / x1 out of range or W == 1 only cost
// contiguity, never correctness.
// ------------------------------------------------------------------------------------------
template <class TR, int LANES>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(MSDA_WAVES_PER_EU, MSDA_WAVES_PER_EU))) void msda_fused_headmajor_kernel(
    const typename TR::storage* __restrict__ value /* [B,M,S,D] */, const int64_t* __restrict__ spatial_shapes,
    const int64_t* __restrict__ level_start, const typename TR::storage* __restrict__ offs,
    const typename TR::storage* __restrict__ logits, typename TR::storage* __restrict__ out, unsigned n_pairs,
    unsigned pairs_per_image, unsigned plane_bytes /* S*D*sizeof */, int M, int L, int P, FusedArgs fa) {
  using S = typename TR::storage;
  using V = typename TR::vec;
  constexpr int VEC = TR::VEC;
  constexpr int D = VEC * LANES;
  constexpr int PL = 2 * LANES;            // lanes per pair
  constexpr int PAIRS = kThreads / PL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Entry* entries = reinterpret_cast<Entry*>(smem_raw);  // [L*P][PAIRS]
  const int LP = L * P;

  const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const unsigned pair0 = tile * PAIRS;
  const unsigned row_bytes = (unsigned)(D * sizeof(S));  // one pixel of one head

  int* s_meta = reinterpret_cast<int*>(entries + (size_t)LP * PAIRS);
  load_level_table(s_meta, spatial_shapes, level_start, L);
  const int pl = threadIdx.x / PL;
  {
    unsigned g = pair0 + pl;
    g = g < n_pairs ? g : n_pairs - 1;
    const unsigned b = g / pairs_per_image, m = g % (unsigned)M;
    build_entries<TR, PL, 4, true>(entries, PAIRS, s_meta, pl, threadIdx.x % PL, g, M, L, P,
                                   (b * (unsigned)M + m) * plane_bytes, row_bytes, offs, logits, fa);
  }
  __syncthreads();

  // phase 2: lane = (pair, x column, 16-byte channel group)
  const int xh = (threadIdx.x / LANES) & 1;
  const int sub = threadIdx.x % LANES;
  const unsigned lane_byte = (unsigned)(sub * 16);
  const unsigned char* vbase = reinterpret_cast<const unsigned char*>(value);
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  constexpr int GROUP = 2 * MSDA_GROUP;  // 2 loads per point -> same 16 loads in flight per lane
  const Entry* my = entries + pl;
  int i = 0;
  for (; i + GROUP <= LP; i += GROUP) {
    unsigned o0[GROUP], o1[GROUP];
    float w0[GROUP], w1[GROUP];
    V r0[GROUP], r1[GROUP];
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
      const Entry en = my[(i + g) * PAIRS];
      o0[g] = xh ? en.off[1] : en.off[0];
      o1[g] = xh ? en.off[3] : en.off[2];
      w0[g] = xh ? en.w[1] : en.w[0];
      w1[g] = xh ? en.w[3] : en.w[2];
    }
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
      r0[g] = *reinterpret_cast<const V*>(vbase + (size_t)(o0[g] + lane_byte));
      r1[g] = *reinterpret_cast<const V*>(vbase + (size_t)(o1[g] + lane_byte));
    }
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = __builtin_fmaf(w0[g], TR::to_f32(r0[g][j]), acc[j]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = __builtin_fmaf(w1[g], TR::to_f32(r1[g][j]), acc[j]);
    }
  }
  for (; i < LP; ++i) {
    const Entry en = my[i * PAIRS];
    const V r0 = *reinterpret_cast<const V*>(vbase + (size_t)((xh ? en.off[1] : en.off[0]) + lane_byte));
    const V r1 = *reinterpret_cast<const V*>(vbase + (size_t)((xh ? en.off[3] : en.off[2]) + lane_byte));
    const float w0 = xh ? en.w[1] : en.w[0], w1 = xh ? en.w[3] : en.w[2];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = __builtin_fmaf(w0, TR::to_f32(r0[j]), acc[j]);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = __builtin_fmaf(w1, TR::to_f32(r1[j]), acc[j]);
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] += __shfl_xor(acc[j], LANES, 64);  // x0 column + x1 column
  const unsigned g = pair0 + pl;
  if (g < n_pairs && xh == 0) {
    V packed;
#pragma unroll
    for (int j = 0; j < VEC; ++j) packed[j] = TR::from_f32(acc[j]);
    *reinterpret_cast<V*>(reinterpret_cast<unsigned char*>(out) + ((size_t)g * D * sizeof(S) + lane_byte)) = packed;
  }
}

template <class TR, int LANES>
int launch_headmajor(hipStream_t st, const void* value, const int64_t* ss, const int64_t* ls, const void* off,
                     const void* logits, void* out, int64_t B, int64_t S, int M, int L, int64_t Nq, int P,
                     FusedArgs fa) {
  using ST = typename TR::storage;
  constexpr int D = TR::VEC * LANES;
  constexpr int PAIRS = kThreads / (2 * LANES);
  const int64_t image_elems = S * M * D;
  const int64_t image_bytes = image_elems * (int64_t)sizeof(ST);
  const int64_t pairs_per_image = Nq * M;
  if (image_bytes > 0xffffffffLL || pairs_per_image * (int64_t)PAIRS > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  if (L * P > 4 * 2 * LANES || L > kMaxLevels) return CODETR_E_UNSUPPORTED;
  int64_t bc = 0xffffffffLL / image_bytes;
  const int64_t bc_pairs = 0x7fffffffLL / pairs_per_image;
  if (bc_pairs < bc) bc = bc_pairs;
  if (bc < 1) return CODETR_E_TOO_LARGE;
  const size_t lds = (size_t)PAIRS * L * P * sizeof(Entry) + kMaxLevels * 4 * sizeof(int);
  for (int64_t b0 = 0; b0 < B; b0 += bc) {
    const int64_t nb = (B - b0) < bc ? (B - b0) : bc;
    const unsigned n_pairs = (unsigned)(nb * pairs_per_image);
    const unsigned grid = (n_pairs + PAIRS - 1) / PAIRS;
    FusedArgs fb = fa;
    fb.ref = fa.ref_f32 ? static_cast<const void*>(static_cast<const float*>(fa.ref) + b0 * Nq * L * fa.ref_dim)
                        : static_cast<const void*>(static_cast<const ST*>(fa.ref) + b0 * Nq * L * fa.ref_dim);
    hipLaunchKernelGGL((msda_fused_headmajor_kernel<TR, LANES>), dim3(grid), dim3(kThreads), lds, st,
                       static_cast<const ST*>(value) + b0 * image_elems, ss, ls,
                       static_cast<const ST*>(off) + b0 * Nq * (int64_t)fa.off_stride,
                       static_cast<const ST*>(logits) + b0 * Nq * (int64_t)fa.logit_stride,
                       static_cast<ST*>(out) + b0 * pairs_per_image * D, n_pairs, (unsigned)pairs_per_image,
                       (unsigned)(S * D * (int64_t)sizeof(ST)), M, L, P, fb);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
// Scalar kernel: one thread per output element, arithmetic in AT (float for f16/bf16/f32
// storage, double for f64).  Same maths as above, any M/D/L/P.
// ------------------------------------------------------------------------------------------
template <typename ST, typename AT, typename CV>
__global__ __launch_bounds__(kThreads) void msda_scalar_kernel(
    const ST* __restrict__ value, const int64_t* __restrict__ spatial_shapes, const int64_t* __restrict__ level_start,
    const ST* __restrict__ loc, const ST* __restrict__ weight, ST* __restrict__ out, int64_t n, int64_t S, int M,
    int D, int L, int64_t Nq, int P) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= n) return;
  int64_t t = idx;
  const int c = (int)(t % D);
  t /= D;
  const int64_t pair = t;
  const int m = (int)(t % M);
  t /= M;
  const int64_t b = t / Nq;
  const int64_t row = (int64_t)M * D;
  const ST* vb = value + b * S * row + (int64_t)m * D + c;
  AT acc = 0;
  int64_t pt = pair * L * P;
  for (int l = 0; l < L; ++l) {
    const int H = (int)spatial_shapes[2 * l], W = (int)spatial_shapes[2 * l + 1];
    const ST* vl = vb + level_start[l] * row;
    for (int p = 0; p < P; ++p, ++pt) {
      const AT x = CV::up(loc[2 * pt]), y = CV::up(loc[2 * pt + 1]), aw = CV::up(weight[pt]);
      const AT h_im = y * (AT)H - (AT)0.5, w_im = x * (AT)W - (AT)0.5;
      if (h_im > (AT)-1 && w_im > (AT)-1 && h_im < (AT)H && w_im < (AT)W) {
        const AT hf = floor(h_im), wf = floor(w_im);
        const int h0 = (int)hf, w0 = (int)wf;
        const AT lh = h_im - hf, lw = w_im - wf, hh = (AT)1 - lh, hw = (AT)1 - lw;
        AT v1 = 0, v2 = 0, v3 = 0, v4 = 0;
        if (h0 >= 0 && w0 >= 0) v1 = CV::up(vl[((int64_t)h0 * W + w0) * row]);
        if (h0 >= 0 && w0 + 1 <= W - 1) v2 = CV::up(vl[((int64_t)h0 * W + w0 + 1) * row]);
        if (h0 + 1 <= H - 1 && w0 >= 0) v3 = CV::up(vl[((int64_t)(h0 + 1) * W + w0) * row]);
        if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) v4 = CV::up(vl[((int64_t)(h0 + 1) * W + w0 + 1) * row]);
        acc += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * aw;
      }
    }
  }
  out[idx] = CV::down(acc);
}

struct CvF16 {
  __device__ static float up(_Float16 v) { return (float)v; }
  __device__ static _Float16 down(float v) { return (_Float16)v; }
};
struct CvBF16 {
  __device__ static float up(unsigned short v) { return BF16::to_f32(v); }
  __device__ static unsigned short down(float v) { return BF16::from_f32(v); }
};
struct CvF32 {
  __device__ static float up(float v) { return v; }
  __device__ static float down(float v) { return v; }
};
struct CvF64 {
  __device__ static double up(double v) { return v; }
  __device__ static double down(double v) { return v; }
};

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
int tiled_lanes(int elem_bytes, int D, int L, int P) {
  // lanes per pair if the tiled kernel applies, else 0
  if (elem_bytes != 2 && elem_bytes != 4) return 0;
  const int vec = 16 / elem_bytes;
  if (D % vec) return 0;
  const int lanes = D / vec;
  if (lanes < 1 || lanes > 32 || (lanes & (lanes - 1))) return 0;
  const long lds = (long)(kThreads / lanes) * L * P * (long)sizeof(Entry);
  if (lds > kMaxTiledLdsBytes || L > kMaxLevels || L * P > 8 * lanes) return 0;
  return lanes;
}

template <class TR, int LANES, bool FUSED>
int launch_tiled(hipStream_t st, const void* value, const int64_t* ss, const int64_t* ls, const void* loc,
                 const void* w, void* out, int64_t B, int64_t S, int M, int L, int64_t Nq, int P, FusedArgs fa,
                 bool behind_op4 = false) {
  using ST = typename TR::storage;
  constexpr int D = TR::VEC * LANES;
  constexpr int PAIRS = kThreads / LANES;
  const int64_t image_elems = S * M * D;
  const int64_t image_bytes = image_elems * (int64_t)sizeof(ST);
  const int64_t pairs_per_image = Nq * M;
  // 32-bit in-chunk addressing: value bytes < 4 GiB and pair count < 2^31 per launch.  Larger
  // batches are cut into chunks (the reference cuts by im2col_step for the same reason its
  // `int` offsets would overflow, cu:939-955).
  if (image_bytes > 0xffffffffLL || pairs_per_image * (int64_t)PAIRS > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  int64_t bc = 0xffffffffLL / image_bytes;
  const int64_t bc_pairs = 0x7fffffffLL / pairs_per_image;
  if (bc_pairs < bc) bc = bc_pairs;
  if (bc < 1) return CODETR_E_TOO_LARGE;
  const size_t lds = (size_t)PAIRS * L * P * sizeof(Entry) + kMaxLevels * 4 * sizeof(int);
  // per-image strides of the loc / weight operands (rows of the projection matrices when fused)
  const int64_t loc_per_image = FUSED ? Nq * (int64_t)fa.off_stride : pairs_per_image * L * P * 2;
  const int64_t w_per_image = FUSED ? Nq * (int64_t)fa.logit_stride : pairs_per_image * L * P;
  for (int64_t b0 = 0; b0 < B; b0 += bc) {
    const int64_t nb = (B - b0) < bc ? (B - b0) : bc;
    const unsigned n_pairs = (unsigned)(nb * pairs_per_image);
    unsigned grid = (n_pairs + PAIRS - 1) / PAIRS;
    if (behind_op4 && grid > 2048u) grid = 2048u;   // (the kernel walks tiles with a grid stride)
    FusedArgs fb = fa;
    if (FUSED)
      fb.ref = fa.ref_f32 ? static_cast<const void*>(static_cast<const float*>(fa.ref) + b0 * Nq * L * fa.ref_dim)
                          : static_cast<const void*>(static_cast<const ST*>(fa.ref) + b0 * Nq * L * fa.ref_dim);
    hipLaunchKernelGGL((msda_tiled_kernel<TR, LANES, FUSED>), dim3(grid), dim3(kThreads), lds, st,
                       static_cast<const ST*>(value) + b0 * image_elems, ss, ls,
                       static_cast<const ST*>(loc) + b0 * loc_per_image, static_cast<const ST*>(w) + b0 * w_per_image,
                       static_cast<ST*>(out) + b0 * pairs_per_image * D, n_pairs, (unsigned)pairs_per_image,
                       (unsigned)image_bytes, M, L, P, fb, behind_op4 ? (int)S : 0, behind_op4 ? (int)(B * M) : 0);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
  }
  return 0;
}

template <class TR>
int dispatch_tiled(int lanes, hipStream_t st, const void* value, const int64_t* ss, const int64_t* ls,
                   const void* loc, const void* w, void* out, int64_t B, int64_t S, int M, int L, int64_t Nq, int P,
                   bool behind_op4 = false) {
  const FusedArgs none{nullptr, 0, 0, 0};
  switch (lanes) {
    case 1: return launch_tiled<TR, 1, false>(st, value, ss, ls, loc, w, out, B, S, M, L, Nq, P, none);
    case 2: return launch_tiled<TR, 2, false>(st, value, ss, ls, loc, w, out, B, S, M, L, Nq, P, none);
    case 4: return launch_tiled<TR, 4, false>(st, value, ss, ls, loc, w, out, B, S, M, L, Nq, P, none, behind_op4);
    case 8: return launch_tiled<TR, 8, false>(st, value, ss, ls, loc, w, out, B, S, M, L, Nq, P, none);
    case 16: return launch_tiled<TR, 16, false>(st, value, ss, ls, loc, w, out, B, S, M, L, Nq, P, none);
    case 32: return launch_tiled<TR, 32, false>(st, value, ss, ls, loc, w, out, B, S, M, L, Nq, P, none);
  }
  return CODETR_E_UNSUPPORTED;
}

// fused variant: only the lane counts the model family uses (D = 32 / 64 in 16-bit storage)
template <class TR>
int dispatch_fused(int lanes, hipStream_t st, const void* value, const int64_t* ss, const int64_t* ls,
                   const void* off, const void* logits, void* out, int64_t B, int64_t S, int M, int L, int64_t Nq,
                   int P, FusedArgs fa) {
  switch (lanes) {
    case 2: return launch_tiled<TR, 2, true>(st, value, ss, ls, off, logits, out, B, S, M, L, Nq, P, fa);
    case 4: return launch_tiled<TR, 4, true>(st, value, ss, ls, off, logits, out, B, S, M, L, Nq, P, fa);
    case 8: return launch_tiled<TR, 8, true>(st, value, ss, ls, off, logits, out, B, S, M, L, Nq, P, fa);
  }
  return CODETR_E_UNSUPPORTED;
}

template <typename ST, typename AT, typename CV>
int launch_scalar(hipStream_t st, const void* value, const int64_t* ss, const int64_t* ls, const void* loc,
                  const void* w, void* out, int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P) {
  const int64_t n = B * Nq * M * D;
  const int64_t grid = (n + kThreads - 1) / kThreads;
  if (grid > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL((msda_scalar_kernel<ST, AT, CV>), dim3((unsigned)grid), dim3(kThreads), 0, st,
                     static_cast<const ST*>(value), ss, ls, static_cast<const ST*>(loc), static_cast<const ST*>(w),
                     static_cast<ST*>(out), n, S, M, D, L, Nq, P);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

int check_args(const void* value, const int64_t* ss, const int64_t* ls, const void* loc, const void* w,
               const void* out, int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P, int64_t im2col_step) {
  if (!value || !ss || !ls || !loc || !w || !out) return CODETR_E_BADARG;
  if (B <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Nq <= 0 || P <= 0 || im2col_step <= 0)
    return CODETR_E_BADARG;
  const int64_t step = B < im2col_step ? B : im2col_step;  // reference cu:924-926
  if (B % step != 0) return CODETR_E_IM2COL_STEP;
  return 0;
}

}  // namespace

extern "C" {

const char* codetr_msda_variant(int elem_bytes, int M, int D, int L, int P) {
  (void)M;
  if (elem_bytes == 8) return "scalar";
  switch (tiled_lanes(elem_bytes, D, L, P)) {
    case 1: return "tiled_x1";
    case 2: return "tiled_x2";
    case 4: return "tiled_x4";
    case 8: return "tiled_x8";
    case 16: return "tiled_x16";
    case 32: return "tiled_x32";
  }
  return "scalar";
}

#define CODETR_MSDA_ENTRY(NAME, TR, ST, AT, CV, EB, OP4)                                                              \
  int NAME(void* stream, const void* value_dev, const int64_t* spatial_shapes_dev, const int64_t* level_start_dev, \
           const void* loc_dev, const void* weight_dev, int64_t B, int64_t S, int M, int D, int L, int64_t Nq,   \
           int P, int64_t im2col_step, void* out_dev) {                                                          \
    const int rc = check_args(value_dev, spatial_shapes_dev, level_start_dev, loc_dev, weight_dev, out_dev, B, S, \
                              M, D, L, Nq, P, im2col_step);                                                      \
    if (rc) return rc;                                                                                           \
    hipStream_t st = static_cast<hipStream_t>(stream);                                                           \
    const int lanes = tiled_lanes(EB, D, L, P);                                                                  \
    /* encoder-shaped fp16 calls: the windowed kernel first, this file's kernel behind it with the skip test on */ \
    bool behind_op4 = false;                                                                                     \
    if (OP4 && lanes == 4 && codetr_msda_op4_supported(EB, B, S, M, D, L, Nq, P)) {                               \
      const int orc = codetr_msda_op4_forward_f16(stream, value_dev, spatial_shapes_dev, level_start_dev, loc_dev, \
                                                  weight_dev, B, S, M, D, L, Nq, P, out_dev);                     \
      if (orc == 0) behind_op4 = true;                                                                           \
      else if (orc != CODETR_E_UNSUPPORTED) return orc;                                                          \
    }                                                                                                            \
    if (lanes) {                                                                                                 \
      const int trc = dispatch_tiled<TR>(lanes, st, value_dev, spatial_shapes_dev, level_start_dev, loc_dev,     \
                                         weight_dev, out_dev, B, S, M, L, Nq, P, behind_op4);                    \
      if (trc != CODETR_E_TOO_LARGE) return trc;                                                                 \
    }                                                                                                            \
    return launch_scalar<ST, AT, CV>(st, value_dev, spatial_shapes_dev, level_start_dev, loc_dev, weight_dev,    \
                                     out_dev, B, S, M, D, L, Nq, P);                                             \
  }

CODETR_MSDA_ENTRY(codetr_msda_forward_f16, F16, _Float16, float, CvF16, 2, true)
CODETR_MSDA_ENTRY(codetr_msda_forward_bf16, BF16, unsigned short, float, CvBF16, 2, false)
CODETR_MSDA_ENTRY(codetr_msda_forward_f32, F32, float, float, CvF32, 4, false)

#define CODETR_MSDA_FUSED_ENTRY(NAME, TR, REF32)                                                                 \
  int NAME(void* stream, const void* value_dev, const int64_t* spatial_shapes_dev, const int64_t* level_start_dev, \
           const void* offsets_dev, int64_t offsets_row_stride, const void* logits_dev, int64_t logits_row_stride, \
           const void* ref_dev, int ref_dim, int value_head_major, int64_t B, int64_t S, int M, int D, int L,    \
           int64_t Nq, int P, void* out_dev) {                                                                   \
    const int rc = check_args(value_dev, spatial_shapes_dev, level_start_dev, offsets_dev, logits_dev, out_dev, B, \
                              S, M, D, L, Nq, P, 1);                                                             \
    if (rc) return rc;                                                                                           \
    if (!ref_dev || (ref_dim != 2 && ref_dim != 4)) return CODETR_E_BADARG;                                      \
    if (offsets_row_stride < (int64_t)M * L * P * 2 || logits_row_stride < (int64_t)M * L * P ||                  \
        offsets_row_stride > 0x7fffffff || logits_row_stride > 0x7fffffff)                                       \
      return CODETR_E_BADARG;                                                                                    \
    const int lanes = tiled_lanes(2, D, L, P);                                                                   \
    if (!lanes || L * P > 8 * lanes) return CODETR_E_UNSUPPORTED;                                                \
    if ((offsets_row_stride & 1) || (reinterpret_cast<uintptr_t>(offsets_dev) & 3) ||                            \
        (reinterpret_cast<uintptr_t>(ref_dev) & (REF32 ? 7 : 3)))                                                \
      return CODETR_E_BADARG; /* (x, y) pairs are read with 32-bit (fp32: 64-bit) loads */                       \
    const FusedArgs fa{ref_dev, ref_dim, (int)offsets_row_stride, (int)logits_row_stride, REF32};                \
    if (value_head_major) {                                                                                      \
      hipStream_t hst = static_cast<hipStream_t>(stream);                                                        \
      if (lanes == 4)                                                                                            \
        return launch_headmajor<TR, 4>(hst, value_dev, spatial_shapes_dev, level_start_dev, offsets_dev,         \
                                       logits_dev, out_dev, B, S, M, L, Nq, P, fa);                              \
      if (lanes == 8)                                                                                            \
        return launch_headmajor<TR, 8>(hst, value_dev, spatial_shapes_dev, level_start_dev, offsets_dev,         \
                                       logits_dev, out_dev, B, S, M, L, Nq, P, fa);                              \
      return CODETR_E_UNSUPPORTED;                                                                               \
    }                                                                                                            \
    return dispatch_fused<TR>(lanes, static_cast<hipStream_t>(stream), value_dev, spatial_shapes_dev,            \
                              level_start_dev, offsets_dev, logits_dev, out_dev, B, S, M, L, Nq, P, fa);         \
  }

CODETR_MSDA_FUSED_ENTRY(codetr_msda_fused_forward_f16, F16, 0)
CODETR_MSDA_FUSED_ENTRY(codetr_msda_fused_forward_bf16, BF16, 0)
CODETR_MSDA_FUSED_ENTRY(codetr_msda_fused_forward_ref32_f16, F16, 1)
CODETR_MSDA_FUSED_ENTRY(codetr_msda_fused_forward_ref32_bf16, BF16, 1)

int codetr_msda_forward_f64(void* stream, const void* value_dev, const int64_t* spatial_shapes_dev,
                            const int64_t* level_start_dev, const void* loc_dev, const void* weight_dev, int64_t B,
                            int64_t S, int M, int D, int L, int64_t Nq, int P, int64_t im2col_step, void* out_dev) {
  const int rc = check_args(value_dev, spatial_shapes_dev, level_start_dev, loc_dev, weight_dev, out_dev, B, S, M, D,
                            L, Nq, P, im2col_step);
  if (rc) return rc;
  return launch_scalar<double, double, CvF64>(static_cast<hipStream_t>(stream), value_dev, spatial_shapes_dev,
                                              level_start_dev, loc_dev, weight_dev, out_dev, B, S, M, D, L, Nq, P);
}

}  // extern "C"
