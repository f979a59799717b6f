// Small element-wise / gather kernels of the detection head and decoder for MI355X (gfx950): the last operations of the
// fp16 inference path that still ran as ATen launches (about 55 per forward) -- `query + query_pos` of the decoder layers
// (reference codetr/transformer_mmcv.py:400-404, multi_scale_deformable_attention.py:161-162), the row gathers of the
// two-stage selection (transformer.py:562-566), the score sigmoid and the box decode of the head
// (co_dino_head.py:169-209) and get_valid_ratio's division (transformer.py:384-400).  With them the whole forward is a
// sequence of C-ABI launches, which is what lets it be recorded into a launch plan and replayed without Python
// (runner/).  fp16 storage; every operation rounds to fp16 exactly where the ATen formulation does (element-wise ATen
// kernels compute in fp32 and round once per operation), so results are bit-identical to it -- tested.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

typedef short s16x8 __attribute__((ext_vector_type(8)));

// 16-bit storage <-> float; BF: bfloat16 (the bf16 model's instantiations), else fp16
template <bool BF>
__device__ __forceinline__ float h2f(unsigned short bits) {
  if (BF) return __uint_as_float(((unsigned)bits) << 16);
  _Float16 h;
  __builtin_memcpy(&h, &bits, 2);
  return (float)h;
}
template <bool BF>
__device__ __forceinline__ unsigned short f2h(float v) {
  if (BF) {
    const unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
  _Float16 h = (_Float16)v;
  unsigned short bits;
  __builtin_memcpy(&bits, &h, 2);
  return bits;
}
template <bool BF>
__device__ __forceinline__ float rh(float v) { return h2f<BF>(f2h<BF>(v)); }  // round to the storage type, keep as float

// out[i] = a[i % a_period] + b[i]   (a_period = n: plain add; a_period < n: `a` broadcast over the leading dimension)
template <bool BF>
__global__ __launch_bounds__(256) void add_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                                  unsigned short* __restrict__ out, int64_t n8, int64_t period8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const s16x8 va = *reinterpret_cast<const s16x8*>(a + (i % period8) * 8);
  const s16x8 vb = *reinterpret_cast<const s16x8*>(b + i * 8);
  s16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (short)f2h<BF>(h2f<BF>((unsigned short)va[e]) + h2f<BF>((unsigned short)vb[e]));
  *reinterpret_cast<s16x8*>(out + i * 8) = o;
}

template <bool BF>
__global__ __launch_bounds__(256) void sigmoid_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ out,
                                                      int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = f2h<BF>(1.0f / (1.0f + expf(-h2f<BF>(x[i]))));
}

// out[b, k, :] = src[b, idx[b, k], :]   (16-bit elements, C % 8 == 0 or C == 4)
__global__ __launch_bounds__(256) void gather_rows_kernel(const unsigned short* __restrict__ src,
                                                          const int64_t* __restrict__ idx, unsigned short* __restrict__ out,
                                                          int64_t B, int64_t S, int64_t K, int C, int pieces, int piece_elems) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / pieces;
  const int p = (int)(t % pieces);
  if (row >= B * K) return;
  const int64_t b = row / K;
  int64_t s = idx[row];
  s = s < 0 ? 0 : (s >= S ? S - 1 : s);
  const unsigned short* g = src + (b * S + s) * C + p * piece_elems;
  unsigned short* o = out + row * C + p * piece_elems;
  if (piece_elems == 8) *reinterpret_cast<s16x8*>(o) = *reinterpret_cast<const s16x8*>(g);
  else *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(g);
}

// head decode (reference co_dino_head.py:177-209), one thread per kept detection:
//   q = idx / C, label = idx % C, (cx, cy, w, h) = sigmoid(coords_unact[b, q]) (fp16), xyxy = (cx - 0.5 w, cy - 0.5 h,
//   cx + 0.5 w, cy + 0.5 h), scaled by (W, H, W, H), clamped to [0, W] x [0, H]; every step rounded to fp16 like the
//   ATen sequence sigmoid / mul / sub / add / cat / mul / clamp / minimum.
template <bool BF>
__global__ __launch_bounds__(256) void decode_boxes_kernel(const unsigned short* __restrict__ coords_unact,
                                                           const int64_t* __restrict__ idx, unsigned short* __restrict__ boxes,
                                                           int64_t* __restrict__ labels, int64_t B, int64_t Nq, int64_t K,
                                                           int num_classes, float Wimg, float Himg) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= B * K) return;
  const int64_t b = t / K, id = idx[t];
  int64_t q = id / num_classes;
  labels[t] = id % num_classes;
  q = q < 0 ? 0 : (q >= Nq ? Nq - 1 : q);
  const uint2 raw = *reinterpret_cast<const uint2*>(coords_unact + (b * Nq + q) * 4);
  float c[4];
  c[0] = rh<BF>(1.0f / (1.0f + expf(-h2f<BF>((unsigned short)(raw.x & 0xffffu)))));
  c[1] = rh<BF>(1.0f / (1.0f + expf(-h2f<BF>((unsigned short)(raw.x >> 16)))));
  c[2] = rh<BF>(1.0f / (1.0f + expf(-h2f<BF>((unsigned short)(raw.y & 0xffffu)))));
  c[3] = rh<BF>(1.0f / (1.0f + expf(-h2f<BF>((unsigned short)(raw.y >> 16)))));
  const float hw = rh<BF>(0.5f * c[2]), hh = rh<BF>(0.5f * c[3]);
  float v[4] = {rh<BF>(c[0] - hw), rh<BF>(c[1] - hh), rh<BF>(c[0] + hw), rh<BF>(c[1] + hh)};
  const float Wh = rh<BF>(Wimg), Hh = rh<BF>(Himg);   // the scale tensor is fp16 too
  const float sc[4] = {Wh, Hh, Wh, Hh};
  unsigned short o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = rh<BF>(v[e] * sc[e]);
    x = x != x ? x : (x < 0.f ? 0.f : x);   // clamp(min=0) keeps NaN
    x = (x != x || sc[e] != sc[e]) ? __builtin_nanf("") : fminf(x, sc[e]);   // torch.minimum propagates NaN
    o[e] = f2h<BF>(x);
  }
  *reinterpret_cast<uint2*>(boxes + t * 4) = uint2{(unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16)};
}

// valid_ratios[b, l, j] = fp16(counts[b, l, j]) / wh[l, j]   (get_valid_ratio: sum(~mask row/col) / W or H, in fp16)
// out32 (optional): the same ratio in fp32, counts / size unrounded (what the fp32 reference computes)
template <bool BF>
__global__ __launch_bounds__(256) void valid_ratios_kernel(const float* __restrict__ counts, const unsigned short* __restrict__ wh,
                                                           unsigned short* __restrict__ out, float* __restrict__ out32, int n,
                                                           int L2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = f2h<BF>(rh<BF>(counts[i]) / h2f<BF>(wh[i % L2]));
  if (out32) out32[i] = counts[i] / h2f<BF>(wh[i % L2]);
}

}  // namespace

namespace {


template <bool BF>
int add_impl(void* stream, const void* a_dev, const void* b_dev, void* out_dev, int64_t n, int64_t a_period) {
  if (!a_dev || !b_dev || !out_dev || n < 0 || a_period <= 0) return CODETR_E_BADARG;
  if (n == 0) return 0;
  if (n % 8 != 0 || a_period % 8 != 0 || n % a_period != 0) return CODETR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a_dev) | reinterpret_cast<uintptr_t>(b_dev) | reinterpret_cast<uintptr_t>(out_dev)) & 15)
    return CODETR_E_UNSUPPORTED;
  const int64_t n8 = n / 8, blocks = (n8 + 255) / 256;
  if (blocks > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(add_kernel<BF>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(a_dev), static_cast<const unsigned short*>(b_dev),
                     static_cast<unsigned short*>(out_dev), n8, a_period / 8);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

template <bool BF>
int sigmoid_impl(void* stream, const void* x_dev, void* out_dev, int64_t n) {
  if (!x_dev || !out_dev || n < 0) return CODETR_E_BADARG;
  if (n == 0) return 0;
  const int64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(sigmoid_kernel<BF>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(x_dev), static_cast<unsigned short*>(out_dev), n);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

template <bool BF>
int decode_boxes_impl(void* stream, const void* coords_unact_dev, const int64_t* idx_dev, void* boxes_dev,
                            int64_t* labels_dev, int64_t B, int64_t Nq, int64_t K, int num_classes, float img_w,
                            float img_h) {
  if (!coords_unact_dev || !idx_dev || !boxes_dev || !labels_dev || B < 0 || Nq <= 0 || K < 0 || num_classes <= 0)
    return CODETR_E_BADARG;
  if (B == 0 || K == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(coords_unact_dev) | reinterpret_cast<uintptr_t>(boxes_dev)) & 7) return CODETR_E_UNSUPPORTED;
  const int64_t blocks = (B * K + 255) / 256;
  if (blocks > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(decode_boxes_kernel<BF>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(coords_unact_dev), idx_dev, static_cast<unsigned short*>(boxes_dev),
                     labels_dev, B, Nq, K, num_classes, img_w, img_h);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

template <bool BF>
int valid_ratios_impl(void* stream, const float* counts_dev, const void* level_wh_f16_dev, void* out_dev,
                            float* out32_dev, int64_t B, int L) {
  if (!counts_dev || !level_wh_f16_dev || !out_dev || B < 0 || L <= 0) return CODETR_E_BADARG;
  if (B == 0) return 0;
  const int64_t n = B * L * 2;
  if (n > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(valid_ratios_kernel<BF>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     counts_dev, static_cast<const unsigned short*>(level_wh_f16_dev), static_cast<unsigned short*>(out_dev),
                     out32_dev, (int)n, 2 * L);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int codetr_gather_rows_b16(void* stream, const void* src_dev, const int64_t* idx_dev, void* out_dev, int64_t B, int64_t S,
                           int64_t K, int64_t C) {
  if (!src_dev || !idx_dev || !out_dev || B < 0 || S <= 0 || K < 0 || C <= 0) return CODETR_E_BADARG;
  if (B == 0 || K == 0) return 0;
  int piece = 0;
  if (C % 8 == 0 && !((reinterpret_cast<uintptr_t>(src_dev) | reinterpret_cast<uintptr_t>(out_dev)) & 15)) piece = 8;
  else if (C % 4 == 0 && !((reinterpret_cast<uintptr_t>(src_dev) | reinterpret_cast<uintptr_t>(out_dev)) & 7)) piece = 4;
  else return CODETR_E_UNSUPPORTED;
  const int pieces = (int)(C / piece);
  const int64_t threads = B * K * pieces, blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffLL || C > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(src_dev), idx_dev, static_cast<unsigned short*>(out_dev), B, S, K,
                     (int)C, pieces, piece);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

int codetr_add_f16(void* stream, const void* a_dev, const void* b_dev, void* out_dev, int64_t n, int64_t a_period) {
  return add_impl<false>(stream, a_dev, b_dev, out_dev, n, a_period);
}
int codetr_add_bf16(void* stream, const void* a_dev, const void* b_dev, void* out_dev, int64_t n, int64_t a_period) {
  return add_impl<true>(stream, a_dev, b_dev, out_dev, n, a_period);
}
int codetr_sigmoid_f16(void* stream, const void* x_dev, void* out_dev, int64_t n) { return sigmoid_impl<false>(stream, x_dev, out_dev, n); }
int codetr_sigmoid_bf16(void* stream, const void* x_dev, void* out_dev, int64_t n) { return sigmoid_impl<true>(stream, x_dev, out_dev, n); }
int codetr_decode_boxes_f16(void* stream, const void* coords_unact_dev, const int64_t* idx_dev, void* boxes_dev,
                            int64_t* labels_dev, int64_t B, int64_t Nq, int64_t K, int num_classes, float img_w,
                            float img_h) {
  return decode_boxes_impl<false>(stream, coords_unact_dev, idx_dev, boxes_dev, labels_dev, B, Nq, K, num_classes, img_w, img_h);
}
int codetr_decode_boxes_bf16(void* stream, const void* coords_unact_dev, const int64_t* idx_dev, void* boxes_dev,
                             int64_t* labels_dev, int64_t B, int64_t Nq, int64_t K, int num_classes, float img_w,
                             float img_h) {
  return decode_boxes_impl<true>(stream, coords_unact_dev, idx_dev, boxes_dev, labels_dev, B, Nq, K, num_classes, img_w, img_h);
}
int codetr_valid_ratios_f16(void* stream, const float* counts_dev, const void* level_wh_f16_dev, void* out_dev,
                            float* out32_dev, int64_t B, int L) {
  return valid_ratios_impl<false>(stream, counts_dev, level_wh_f16_dev, out_dev, out32_dev, B, L);
}
int codetr_valid_ratios_bf16(void* stream, const float* counts_dev, const void* level_wh_bf16_dev, void* out_dev,
                             float* out32_dev, int64_t B, int L) {
  return valid_ratios_impl<true>(stream, counts_dev, level_wh_bf16_dev, out_dev, out32_dev, B, L);
}

}  // extern "C"
