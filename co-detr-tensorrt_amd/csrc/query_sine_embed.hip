// Decoder query positions of one DINO decoder layer, one launch.
//
// Replaces the head of every DinoTransformerDecoder layer iteration (reference codetr/transformer.py:208-217:
//   reference_points_input = reference_points[:, :, None].sigmoid()-space * cat([valid_ratios, valid_ratios], -1)[:, None]
//   query_sine_embed = gen_sineembed_for_position(reference_points_input[:, :, 0, :], embed_dims // 2)
// and gen_sineembed_for_position itself, codetr/transformer.py:157-190: arange / floor-div / pow for dim_t, then per
// coordinate  * 2 pi, / dim_t, sin, cos, stack, flatten, and a final cat) -- 27 ATen launches of ~5 us each per
// layer at 900 queries -- with
//   ref_in[b, q, l, c]  = f16(f16(sigmoid(ref[b, q, c])) * valid_ratios[b, l, c & 1])           [B, Nq, L, ref_dim]
//   embed[b, q, j*F + i] = sin | cos (ref_in[b, q, 0, order[j]] * 2 pi / T^(2 (i/2) / F)),  i even -> sin, odd -> cos
//                          order = (y, x, w, h) = coordinate (1, 0, 2, 3)                       [B, Nq, ref_dim*F]
// The trigonometry runs in fp32 on the fp16-rounded ref_in (the values the reference feeds it); one lane = 8 channels.
// With fp32 valid ratios and an fp32 output given (ref_in32), the layer's MSDA reference points are ALSO written
// unrounded -- sigmoid and the scaling in fp32 -- and the embedding is taken from those: a coordinate in [0.5, 1)
// resolves to 1/2048 in fp16, a quarter pixel on a 480-wide level, which is what separates an fp16 model's decoder from
// the fp32 reference at 1920x1280.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// 16-bit storage <-> float; BF: bfloat16, else fp16
template <bool BF>
__device__ __forceinline__ float ld16(unsigned short bits) {
  if (BF) return __uint_as_float(((unsigned)bits) << 16);
  _Float16 h;
  __builtin_memcpy(&h, &bits, 2);
  return (float)h;
}
template <bool BF>
__device__ __forceinline__ unsigned short st16(float v) {
  if (BF) {
    const unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
  _Float16 h = (_Float16)v;
  unsigned short bits;
  __builtin_memcpy(&bits, &h, 2);
  return bits;
}

template <bool BF>
__global__ __launch_bounds__(256) void query_sine_embed_kernel(const unsigned short* __restrict__ ref,
                                                               const unsigned short* __restrict__ valid_ratios,
                                                               const float* __restrict__ valid_ratios32,
                                                               unsigned short* __restrict__ ref_in, float* __restrict__ ref_in32,
                                                               unsigned short* __restrict__ embed,
                                                               int64_t rows, int Nq, int ref_dim, int L, int F,
                                                               float log2_temperature, int apply_sigmoid) {
  const int chunks = ref_dim * F / 8;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * chunks) return;
  const int c = (int)(i % chunks);
  const int64_t row = i / chunks;  // b*Nq + q
  const int b = (int)(row / Nq);
  float s[4], s32[4];
  for (int k = 0; k < ref_dim; ++k) {
    float v = ld16<BF>(ref[row * ref_dim + k]);
    s32[k] = apply_sigmoid ? 1.0f / (1.0f + expf(-v)) : v;
    if (apply_sigmoid) v = ld16<BF>(st16<BF>(1.0f / (1.0f + __expf(-v))));
    s[k] = v;
  }
  const unsigned short* vr = valid_ratios + (size_t)b * L * 2;
  const float* vr32 = valid_ratios32 ? valid_ratios32 + (size_t)b * L * 2 : nullptr;
  // level rows of ref_in: lanes c = 0..L-1 of this query write one level each
  if (c < L)
    for (int k = 0; k < ref_dim; ++k) {
      ref_in[(row * L + c) * ref_dim + k] = st16<BF>(s[k] * ld16<BF>(vr[c * 2 + (k & 1)]));
      if (ref_in32) ref_in32[(row * L + c) * ref_dim + k] = s32[k] * vr32[c * 2 + (k & 1)];
    }
  const int j = (c * 8) / F;                          // coordinate block of this chunk
  const int coord = j == 0 ? 1 : (j == 1 ? 0 : j);    // (y, x, w, h)
  const float v0 = ref_in32 ? s32[coord] * vr32[coord & 1]
                            : ld16<BF>(st16<BF>(s[coord] * ld16<BF>(vr[coord & 1])));  // ref_in[b, q, 0, coord]
  const float e = v0 * 6.283185307179586f;
  const int ch0 = c * 8 - j * F;
  u16x8 o;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int f = (ch0 >> 1) + p;
    const float rev = e * __builtin_amdgcn_exp2f(-log2_temperature * (2.0f * (float)f / (float)F)) * 0.15915494309189535f;
    o[2 * p] = st16<BF>(__builtin_amdgcn_sinf(rev));  // v_sin_f32 on revolutions (angle <= 2 pi here)
    o[2 * p + 1] = st16<BF>(__builtin_amdgcn_cosf(rev));
  }
  *reinterpret_cast<u16x8*>(embed + row * (int64_t)(ref_dim * F) + c * 8) = o;
}

}  // namespace

namespace {


template <bool BF>
int query_sine_embed_impl(void* stream, const void* ref_dev, const void* valid_ratios_dev,
                                const float* valid_ratios32_dev, int64_t B, int64_t Nq, int ref_dim, int num_levels,
                                int pos_feat, float temperature, int apply_sigmoid, void* ref_in_dev, float* ref_in32_dev,
                                void* embed_dev) {
  if (!ref_dev || !valid_ratios_dev || !ref_in_dev || !embed_dev || B <= 0 || Nq <= 0 || num_levels <= 0 ||
      temperature <= 0.f || (ref_in32_dev && !valid_ratios32_dev))
    return CODETR_E_BADARG;
  if ((ref_dim != 2 && ref_dim != 4) || pos_feat <= 0 || pos_feat % 8 != 0 || num_levels > ref_dim * pos_feat / 8)
    return CODETR_E_UNSUPPORTED;
  if (B * Nq > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  const int64_t threads = B * Nq * (ref_dim * pos_feat / 8);
  hipLaunchKernelGGL(query_sine_embed_kernel<BF>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(ref_dev),
                     static_cast<const unsigned short*>(valid_ratios_dev), valid_ratios32_dev, static_cast<unsigned short*>(ref_in_dev),
                     ref_in32_dev, static_cast<unsigned short*>(embed_dev), B * Nq, (int)Nq, ref_dim, num_levels, pos_feat, log2f(temperature),
                     apply_sigmoid);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int codetr_query_sine_embed_f16(void* stream, const void* ref_dev, const void* valid_ratios_dev,
                                const float* valid_ratios32_dev, int64_t B, int64_t Nq, int ref_dim, int num_levels,
                                int pos_feat, float temperature, int apply_sigmoid, void* ref_in_dev, float* ref_in32_dev,
                                void* embed_dev) {
  return query_sine_embed_impl<false>(stream, ref_dev, valid_ratios_dev, valid_ratios32_dev, B, Nq, ref_dim, num_levels,
                                      pos_feat, temperature, apply_sigmoid, ref_in_dev, ref_in32_dev, embed_dev);
}
int codetr_query_sine_embed_bf16(void* stream, const void* ref_dev, const void* valid_ratios_dev,
                                 const float* valid_ratios32_dev, int64_t B, int64_t Nq, int ref_dim, int num_levels,
                                 int pos_feat, float temperature, int apply_sigmoid, void* ref_in_dev, float* ref_in32_dev,
                                 void* embed_dev) {
  return query_sine_embed_impl<true>(stream, ref_dev, valid_ratios_dev, valid_ratios32_dev, B, Nq, ref_dim, num_levels,
                                     pos_feat, temperature, apply_sigmoid, ref_in_dev, ref_in32_dev, embed_dev);
}

}  // extern "C"
