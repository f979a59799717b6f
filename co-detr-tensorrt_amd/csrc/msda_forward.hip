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
  const void* ref;      // [B*Nq, L, ref_dim]
  int ref_dim;          // 2 or 4
  int off_stride;       // elements between consecutive (b, q) rows of the offsets matrix
  int logit_stride;     // same for the logits matrix
};

template <class TR, int LANES, bool FUSED>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(MSDA_WAVES_PER_EU, MSDA_WAVES_PER_EU))) void msda_tiled_kernel(
    const typename TR::storage* __restrict__ value, const int64_t* __restrict__ spatial_shapes,
    const int64_t* __restrict__ level_start, const typename TR::storage* __restrict__ loc,
    const typename TR::storage* __restrict__ weight, typename TR::storage* __restrict__ out,
    unsigned n_pairs /* B*Nq*M */, unsigned pairs_per_image /* Nq*M */, unsigned image_bytes /* S*M*D*sizeof */,
    int M, int L, int P, FusedArgs fa) {
  using S = typename TR::storage;
  constexpr int VEC = TR::VEC;
  constexpr int D = VEC * LANES;
  constexpr int PAIRS = kThreads / LANES;  // pairs per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Entry* entries = reinterpret_cast<Entry*>(smem_raw);  // [L*P][PAIRS]

  const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const unsigned pair0 = tile * PAIRS;
  const int LP = L * P;
  const unsigned row_bytes = (unsigned)(M * D * sizeof(S));  // one pixel, all heads

  // ---------------- phase 0 (fused only): softmax statistics of every pair's L*P logits ----------------
  float* sm_stats = reinterpret_cast<float*>(entries + (size_t)LP * PAIRS);  // [PAIRS][2] = (max, 1/sum)
  if (FUSED) {
    // the LANES lanes that will serve a pair in phase 2 also split its L*P logits here; max and sum are
    // combined with shuffles inside that (power-of-two, aligned) lane group -- every thread is busy
    const int pl0 = threadIdx.x / LANES, sub0 = threadIdx.x % LANES;
    unsigned g = pair0 + pl0;
    g = g < n_pairs ? g : n_pairs - 1;
    const S* lg = weight + (size_t)(g / (unsigned)M) * fa.logit_stride + (size_t)(g % (unsigned)M) * LP;
    float mx = -INFINITY;
    for (int i = sub0; i < LP; i += LANES) mx = fmaxf(mx, TR::to_f32(lg[i]));
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int i = sub0; i < LP; i += LANES) sum += __expf(TR::to_f32(lg[i]) - mx);
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (sub0 == 0) {
      sm_stats[2 * pl0] = mx;
      sm_stats[2 * pl0 + 1] = 1.0f / sum;
    }
    __syncthreads();
  }

  // ---------------- phase 1: (x, y, w) -> {offsets, weights} into LDS ----------------
  for (int l = 0; l < L; ++l) {
    // uniform index -> scalar loads; shapes stay on device (reference cu:236-239)
    const int H = (int)spatial_shapes[2 * l];
    const int W = (int)spatial_shapes[2 * l + 1];
    const unsigned start = (unsigned)level_start[l];
    const float Hf = (float)H, Wf = (float)W;
    const float invH = 1.0f / Hf, invW = 1.0f / Wf, half_over_p = 0.5f / (float)P;
    for (int e = threadIdx.x; e < PAIRS * P; e += kThreads) {
      const int pl = e / P, p = e - pl * P;
      unsigned g = pair0 + pl;
      g = g < n_pairs ? g : n_pairs - 1;  // tail: recompute the last pair, store is masked
      const unsigned b = g / pairs_per_image;
      const unsigned m = g % (unsigned)M;
      float x, y, aw;
      if (FUSED) {
        const unsigned row = g / (unsigned)M;  // (b, q)
        const int col = ((int)m * L + l) * P + p;
        const float ox = TR::to_f32(loc[(size_t)row * fa.off_stride + 2 * col]);
        const float oy = TR::to_f32(loc[(size_t)row * fa.off_stride + 2 * col + 1]);
        const S* rp = static_cast<const S*>(fa.ref) + ((size_t)row * L + l) * fa.ref_dim;
        const float rx = TR::to_f32(rp[0]), ry = TR::to_f32(rp[1]);
        if (fa.ref_dim == 2) {
          x = fmaf(ox, invW, rx);
          y = fmaf(oy, invH, ry);
        } else {
          x = fmaf(ox * half_over_p, TR::to_f32(rp[2]), rx);
          y = fmaf(oy * half_over_p, TR::to_f32(rp[3]), ry);
        }
        aw = __expf(TR::to_f32(weight[(size_t)row * fa.logit_stride + col]) - sm_stats[2 * pl]) * sm_stats[2 * pl + 1];
      } else {
        const size_t pt = ((size_t)g * L + l) * P + p;
        x = TR::to_f32(loc[2 * pt]);
        y = TR::to_f32(loc[2 * pt + 1]);
        aw = TR::to_f32(weight[pt]);
      }
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
      const unsigned base = b * image_bytes + start * row_bytes + m * (unsigned)(D * sizeof(S));
      en.off[0] = base + (unsigned)(h0c * W + w0c) * row_bytes;
      en.off[1] = base + (unsigned)(h0c * W + w1c) * row_bytes;
      en.off[2] = base + (unsigned)(h1c * W + w0c) * row_bytes;
      en.off[3] = base + (unsigned)(h1c * W + w1c) * row_bytes;
      entries[(l * P + p) * PAIRS + pl] = en;
    }
  }
  __syncthreads();

  // ---------------- phase 2: gather + blend ----------------
  const int pl = threadIdx.x / LANES;
  const int sub = threadIdx.x % LANES;
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
  for (; i + GROUP <= LP; i += GROUP) {
    Entry en[GROUP];
    V raw[GROUP][4];
#pragma unroll
    for (int g = 0; g < GROUP; ++g) en[g] = my[(i + g) * PAIRS];
#pragma unroll
    for (int g = 0; g < GROUP; ++g)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        raw[g][k] = *reinterpret_cast<const V*>(vbase + (size_t)(en[g].off[k] + lane_byte));
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
  if (lds > kMaxTiledLdsBytes) return 0;
  return lanes;
}

template <class TR, int LANES, bool FUSED>
int launch_tiled(hipStream_t st, const void* value, const int64_t* ss, const int64_t* ls, const void* loc,
                 const void* w, void* out, int64_t B, int64_t S, int M, int L, int64_t Nq, int P, FusedArgs fa) {
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
  const size_t lds = (size_t)PAIRS * L * P * sizeof(Entry) + (FUSED ? (size_t)PAIRS * 2 * sizeof(float) : 0);
  // per-image strides of the loc / weight operands (rows of the projection matrices when fused)
  const int64_t loc_per_image = FUSED ? Nq * (int64_t)fa.off_stride : pairs_per_image * L * P * 2;
  const int64_t w_per_image = FUSED ? Nq * (int64_t)fa.logit_stride : pairs_per_image * L * P;
  for (int64_t b0 = 0; b0 < B; b0 += bc) {
    const int64_t nb = (B - b0) < bc ? (B - b0) : bc;
    const unsigned n_pairs = (unsigned)(nb * pairs_per_image);
    const unsigned grid = (n_pairs + PAIRS - 1) / PAIRS;
    FusedArgs fb = fa;
    if (FUSED) fb.ref = static_cast<const ST*>(fa.ref) + b0 * Nq * L * fa.ref_dim;
    hipLaunchKernelGGL((msda_tiled_kernel<TR, LANES, FUSED>), dim3(grid), dim3(kThreads), lds, st,
                       static_cast<const ST*>(value) + b0 * image_elems, ss, ls,
                       static_cast<const ST*>(loc) + b0 * loc_per_image, static_cast<const ST*>(w) + b0 * w_per_image,
                       static_cast<ST*>(out) + b0 * pairs_per_image * D, n_pairs, (unsigned)pairs_per_image,
                       (unsigned)image_bytes, M, L, P, fb);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
  }
  return 0;
}

template <class TR>
int dispatch_tiled(int lanes, hipStream_t st, const void* value, const int64_t* ss, const int64_t* ls,
                   const void* loc, const void* w, void* out, int64_t B, int64_t S, int M, int L, int64_t Nq, int P) {
  const FusedArgs none{nullptr, 0, 0, 0};
  switch (lanes) {
    case 1: return launch_tiled<TR, 1, false>(st, value, ss, ls, loc, w, out, B, S, M, L, Nq, P, none);
    case 2: return launch_tiled<TR, 2, false>(st, value, ss, ls, loc, w, out, B, S, M, L, Nq, P, none);
    case 4: return launch_tiled<TR, 4, false>(st, value, ss, ls, loc, w, out, B, S, M, L, Nq, P, none);
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

#define CODETR_MSDA_ENTRY(NAME, TR, ST, AT, CV, EB)                                                              \
  int NAME(void* stream, const void* value_dev, const int64_t* spatial_shapes_dev, const int64_t* level_start_dev, \
           const void* loc_dev, const void* weight_dev, int64_t B, int64_t S, int M, int D, int L, int64_t Nq,   \
           int P, int64_t im2col_step, void* out_dev) {                                                          \
    const int rc = check_args(value_dev, spatial_shapes_dev, level_start_dev, loc_dev, weight_dev, out_dev, B, S, \
                              M, D, L, Nq, P, im2col_step);                                                      \
    if (rc) return rc;                                                                                           \
    hipStream_t st = static_cast<hipStream_t>(stream);                                                           \
    const int lanes = tiled_lanes(EB, D, L, P);                                                                  \
    if (lanes) {                                                                                                 \
      const int trc = dispatch_tiled<TR>(lanes, st, value_dev, spatial_shapes_dev, level_start_dev, loc_dev,     \
                                         weight_dev, out_dev, B, S, M, L, Nq, P);                                \
      if (trc != CODETR_E_TOO_LARGE) return trc;                                                                 \
    }                                                                                                            \
    return launch_scalar<ST, AT, CV>(st, value_dev, spatial_shapes_dev, level_start_dev, loc_dev, weight_dev,    \
                                     out_dev, B, S, M, D, L, Nq, P);                                             \
  }

CODETR_MSDA_ENTRY(codetr_msda_forward_f16, F16, _Float16, float, CvF16, 2)
CODETR_MSDA_ENTRY(codetr_msda_forward_bf16, BF16, unsigned short, float, CvBF16, 2)
CODETR_MSDA_ENTRY(codetr_msda_forward_f32, F32, float, float, CvF32, 4)

#define CODETR_MSDA_FUSED_ENTRY(NAME, TR)                                                                        \
  int NAME(void* stream, const void* value_dev, const int64_t* spatial_shapes_dev, const int64_t* level_start_dev, \
           const void* offsets_dev, int64_t offsets_row_stride, const void* logits_dev, int64_t logits_row_stride, \
           const void* ref_dev, int ref_dim, int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P,       \
           void* out_dev) {                                                                                      \
    const int rc = check_args(value_dev, spatial_shapes_dev, level_start_dev, offsets_dev, logits_dev, out_dev, B, \
                              S, M, D, L, Nq, P, 1);                                                             \
    if (rc) return rc;                                                                                           \
    if (!ref_dev || (ref_dim != 2 && ref_dim != 4)) return CODETR_E_BADARG;                                      \
    if (offsets_row_stride < (int64_t)M * L * P * 2 || logits_row_stride < (int64_t)M * L * P ||                  \
        offsets_row_stride > 0x7fffffff || logits_row_stride > 0x7fffffff)                                       \
      return CODETR_E_BADARG;                                                                                    \
    const int lanes = tiled_lanes(2, D, L, P);                                                                   \
    if (!lanes) return CODETR_E_UNSUPPORTED;                                                                     \
    const FusedArgs fa{ref_dev, ref_dim, (int)offsets_row_stride, (int)logits_row_stride};                       \
    return dispatch_fused<TR>(lanes, static_cast<hipStream_t>(stream), value_dev, spatial_shapes_dev,            \
                              level_start_dev, offsets_dev, logits_dev, out_dev, B, S, M, L, Nq, P, fa);         \
  }

CODETR_MSDA_FUSED_ENTRY(codetr_msda_fused_forward_f16, F16)
CODETR_MSDA_FUSED_ENTRY(codetr_msda_fused_forward_bf16, BF16)

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
