// Dense multi-head softmax attention for the decoder's self-attention (reference codetr/transformer_mmcv.py:319-428:
// nn.MultiheadAttention(256, 8) on 900 queries -- the part between its in- and out-projections, which the host runs
// as codetr_linear_* GEMMs).  head_dim 32, no mask, softmax scale 1/sqrt(32); replaces the SDPA library call.
//
// One workgroup = (image, head, 128 queries): the head's K and V ([Nk, 32] each, 57.6 KB at Nk = 900) are staged once
// in LDS (all row loads requested before the first LDS write); each of the 8 waves owns 16 queries and walks the keys
// in chunks of 128 with an online softmax (running max / sum in the log2 domain, accumulator rescaled per chunk).
// Same MFMA formulation as window_attention.hip: S^T = K . Q^T (v_mfma_f32_16x16x32, K = head_dim, one MFMA per 16
// keys) so that a lane owns 4 consecutive keys of one query and the rounded probabilities ARE the B operand of
// O^T = V^T . P^T; V^T fragments through ds_read_b64_tr_b16 from the row-major V image.  fp32 scores / softmax /
// accumulation, one rounding at the store.
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

constexpr int HD = 32;
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kQPerWg = 16 * kWaves;  // 128 queries per workgroup
constexpr int kChunkTiles = 8;        // 128 keys per online-softmax step
constexpr int kMaxKeys = 1024;        // K and V images: 2 x 64 KB of LDS

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// XOR swizzle of the 16-byte chunk of a 64-byte (head_dim 32) LDS row.  With j = (row >> 2) & 3 the key f(j) = j ^ ((j & 1) << 1)
// = {0, 3, 2, 1} keeps both read shapes conflict-free under the hardware's lane groups (tests/test_lds_bank_model.py):
// the K fragments (ds_read_b128: a group of 16 lanes holds rows {0-3, 12-15} at chunk g and rows {4-11} at chunk g ^ 1)
// and the transposing V reads (ds_read_b64_tr_b16: 32 lanes = 8 rows x two chunks; rows r and r + 4 alias mod 256 B and
// need keys that differ in bit 1).  The plain key f(j) = j is 2-way conflicted on both.
__device__ __forceinline__ int swz(int row, int chunk) {
  const int j = (row >> 2) & 3;
  return chunk ^ j ^ ((j & 1) << 1);
}

template <class ET>
__global__ __launch_bounds__(kThreads) void mha_attention_kernel(const typename ET::e* __restrict__ q,
                                                                 const typename ET::e* __restrict__ k,
                                                                 const typename ET::e* __restrict__ v,
                                                                 typename ET::e* __restrict__ out, int Nq, int Nk, int H,
                                                                 int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                                                 int64_t o_stride, int q_tiles, float scale_log2e) {
  using E = typename ET::e;
  using V8 = typename ET::v8;
  using V4 = typename ET::v4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, grp = lane >> 4;
  const int qt = blockIdx.x % q_tiles;
  const int head = (blockIdx.x / q_tiles) % H;
  const int b = blockIdx.x / (q_tiles * H);
  const int NP = (Nk + 31) & ~31;  // rows of the K / V images (zero-filled beyond Nk)
  unsigned char* ldsK = lds;
  unsigned char* ldsV = lds + (size_t)NP * 64;
  const int hoff = head * HD;

  // ---- this wave's query fragment (B operand: lane (j = l15, g) holds Q[query][8g .. 8g+7]) ----
  const int qi = qt * kQPerWg + wave * 16 + l15;
  const bool q_in = qi < Nq;
  const V8 qf = *reinterpret_cast<const V8*>(q + ((size_t)b * Nq + (q_in ? qi : Nq - 1)) * q_stride + hoff + grp * 8);

  // ---- stage K and V of (b, head): every row load is requested before the first LDS write ----
  {
    const E* kb = k + (size_t)b * Nk * k_stride + hoff;
    const E* vb = v + (size_t)b * Nk * v_stride + hoff;
    constexpr int ITER = (kMaxKeys * 4) / kThreads;  // 16-byte pieces per thread at the largest Nk
    s16x8 kv[ITER], vv[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int c = tid + it * kThreads;
      const int row = c >> 2, chunk = c & 3;
      kv[it] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
      vv[it] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (row < Nk) {
        kv[it] = *reinterpret_cast<const s16x8*>(kb + (size_t)row * k_stride + chunk * 8);
        vv[it] = *reinterpret_cast<const s16x8*>(vb + (size_t)row * v_stride + chunk * 8);
      }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int c = tid + it * kThreads;
      const int row = c >> 2, chunk = c & 3;
      if (row < NP) {
        const int pos = swz(row, chunk) * 16;
        *reinterpret_cast<s16x8*>(ldsK + row * 64 + pos) = kv[it];
        *reinterpret_cast<s16x8*>(ldsV + row * 64 + pos) = vv[it];
      }
    }
  }
  __syncthreads();

  // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4 x 16 block
  const int tr_q = l15 >> 2, tr_p = l15 & 3;
  float m_run = -INFINITY, l_run = 0.f;  // running max (log2 domain) and sum of this lane's query
  f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int n_tiles = NP >> 4;
  for (int t0 = 0; t0 < n_tiles; t0 += kChunkTiles) {
    // ---- S^T tiles of this chunk: D[i = key][j = query] ----
    f32x4 s[kChunkTiles];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < kChunkTiles; ++kt) {
      const int row = (t0 + kt) * 16 + l15;
      s[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (t0 + kt < n_tiles) {
        const V8 kf = *reinterpret_cast<const V8*>(ldsK + row * 64 + swz(row, grp) * 16);
        s[kt] = ET::mfma(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f});
        const int key0 = (t0 + kt) * 16 + grp * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float val = key0 + r < Nk ? s[kt][r] * scale_log2e : -INFINITY;
          s[kt][r] = val;
          mx = fmaxf(mx, val);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);  // finite: every chunk holds at least one real key
    const float corr = __builtin_amdgcn_exp2f(m_run - m_new);  // exp2(-inf) = 0 on the first chunk
    // ---- exp, chunk sum, pack P^T as MFMA B fragments (32 keys per fragment) ----
    float sum = 0.f;
    V8 pf[kChunkTiles / 2];
#pragma unroll
    for (int ks = 0; ks < kChunkTiles / 2; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[2 * ks + h][r] - m_new);  // masked / absent keys: exp2(-inf) = 0
          sum += p;
          pf[ks][h * 4 + r] = (E)p;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * corr + sum;
    m_run = m_new;
    // ---- O^T = corr * O^T + V^T . P^T : D[i = channel][j = query] ----
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] *= corr;
#pragma unroll
    for (int ks = 0; ks < kChunkTiles / 2; ++ks) {
      if (t0 + 2 * ks < n_tiles) {  // (NP is a multiple of 32: both tiles of the step exist)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int chunk = dt * 2 + (tr_p >> 1);
          const int row0 = (t0 + 2 * ks) * 16 + grp * 4 + tr_q;
          const int row1 = row0 + 16;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(ldsV + row0 * 64 + swz(row0, chunk) * 16 + (tr_p & 1) * 8));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(ldsV + row1 * 64 + swz(row1, chunk) * 16 + (tr_p & 1) * 8));
          const s16x8 vf8 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          V8 vf;
          __builtin_memcpy(&vf, &vf8, 16);
          o[dt] = ET::mfma(vf, pf[ks], o[dt]);
        }
      }
    }
  }
  // ---- normalise and store: lane holds channels 16*dt + 4*grp + r of query l15 ----
  if (q_in) {
    const float inv = 1.0f / l_run;
    E* dst = out + ((size_t)b * Nq + qi) * o_stride + hoff + grp * 4;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      V4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = (E)(o[dt][r] * inv);
      *reinterpret_cast<V4*>(dst + dt * 16) = ov;
    }
  }
}

template <class ET>
int mha_entry(void* stream, const void* q, const void* k, const void* v, void* out, int64_t B, int64_t Nq, int64_t Nk,
              int H, int head_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride) {
  using E = typename ET::e;
  if (!q || !k || !v || !out || B <= 0 || Nq <= 0 || Nk <= 0 || H <= 0) return CODETR_E_BADARG;
  if (head_dim != HD || Nk > kMaxKeys) return CODETR_E_UNSUPPORTED;
  const int64_t C = (int64_t)H * HD;
  if (q_stride < C || k_stride < C || v_stride < C || o_stride < C || (q_stride | k_stride | v_stride) % 8 != 0 ||
      o_stride % 4 != 0)
    return CODETR_E_BADARG;  // 16-byte row chunks / 8-byte output groups
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15 ||
      reinterpret_cast<uintptr_t>(out) & 7)
    return CODETR_E_BADARG;
  const int q_tiles = (int)((Nq + kQPerWg - 1) / kQPerWg);
  const int64_t blocks = B * H * q_tiles;
  if (blocks > 0x7fffffffLL || Nq > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  const int NP = ((int)Nk + 31) & ~31;
  const size_t lds = (size_t)NP * 128;
  {
    // per device (the attribute belongs to the current device's function object), not per process
    static std::atomic<bool> done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0, done[0].store(false);
    if (!done[dev].load(std::memory_order_acquire)) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mha_attention_kernel<ET>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kMaxKeys * 128);
      if (e != hipSuccess) return (int)e;
      done[dev].store(true, std::memory_order_release);
    }
  }
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)HD);
  hipLaunchKernelGGL(mha_attention_kernel<ET>, dim3((unsigned)blocks), dim3(kThreads), lds,
                     static_cast<hipStream_t>(stream), static_cast<const E*>(q), static_cast<const E*>(k),
                     static_cast<const E*>(v), static_cast<E*>(out), (int)Nq, (int)Nk, H, q_stride, k_stride, v_stride,
                     o_stride, q_tiles, scale_log2e);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int codetr_mha_attention_f16(void* stream, const void* q_dev, const void* k_dev, const void* v_dev, void* out_dev,
                             int64_t B, int64_t Nq, int64_t Nk, int num_heads, int head_dim, int64_t q_row_stride,
                             int64_t k_row_stride, int64_t v_row_stride, int64_t out_row_stride) {
  return mha_entry<F16E>(stream, q_dev, k_dev, v_dev, out_dev, B, Nq, Nk, num_heads, head_dim, q_row_stride, k_row_stride,
                         v_row_stride, out_row_stride);
}

int codetr_mha_attention_bf16(void* stream, const void* q_dev, const void* k_dev, const void* v_dev, void* out_dev,
                              int64_t B, int64_t Nq, int64_t Nk, int num_heads, int head_dim, int64_t q_row_stride,
                              int64_t k_row_stride, int64_t v_row_stride, int64_t out_row_stride) {
  return mha_entry<BF16E>(stream, q_dev, k_dev, v_dev, out_dev, B, Nq, Nk, num_heads, head_dim, q_row_stride,
                          k_row_stride, v_row_stride, out_row_stride);
}

}  // extern "C"
