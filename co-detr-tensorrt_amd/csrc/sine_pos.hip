// Sine positional encoding of one pyramid level, produced directly in the transformer's layout.
//
// Replaces SinePositionalEncoding.forward (reference codetr/positional_encoding.py:58-93: normalise the running
// sums, divide by 128 temperatures, interleaved sin / cos, cat, permute -- ~14 ATen kernels per level) plus the
// flatten / transpose / "+ level_embed" / cat steps of CoDinoTransformer.forward (codetr/transformer.py:508-519):
// one launch writes lvl_pos_embed[b, level_start + y*W + x, :] = [pos_y (num_feats) | pos_x (num_feats)] + level_embed.
//
// Inputs are the running sums of the not-mask along y and x (exact small integers, computed by the host with two
// cumsum calls); everything else happens here in fp32:  e = (cum + offset) / (last + eps) * scale,
// channel 2f -> sin(e / T^(2f/num_feats)), channel 2f+1 -> cos(same).  One lane = 8 output channels (16 B).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

constexpr int kThreads = 256;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// E = _Float16 or __bf16 (storage of the encoding and of level_embed; fp32 arithmetic either way)
template <class E, class V8>
__global__ __launch_bounds__(kThreads) void sine_pos_kernel(const float* __restrict__ ycum, const float* __restrict__ xcum,
                                                            const E* __restrict__ level_embed,
                                                            E* __restrict__ out, int64_t out_batch_stride, int H,
                                                            int W, int num_feats, float log2_temperature, float scale,
                                                            float eps, float offset, int normalize, int64_t total_chunks) {
  const int lanes_per_token = (2 * num_feats) >> 3;  // 16-byte chunks per token
  const int half = lanes_per_token >> 1;             // chunks per axis
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total_chunks; i += (int64_t)gridDim.x * kThreads) {
    const int c = (int)(i % lanes_per_token);
    const int64_t tok = i / lanes_per_token;  // b*H*W + y*W + x
    const int hw = H * W;
    const int b = (int)(tok / hw);
    const int r = (int)(tok - (int64_t)b * hw);
    const int y = r / W, x = r - y * W;
    const bool is_x = c >= half;
    const float* cum = is_x ? xcum : ycum;
    float e = cum[tok];
    if (normalize) {
      const float last = is_x ? xcum[(int64_t)b * hw + y * W + (W - 1)] : ycum[(int64_t)b * hw + (H - 1) * W + x];
      e = (e + offset) / (last + eps) * scale;
    }
    const int ch0 = (is_x ? c - half : c) * 8;  // first channel of this chunk inside its axis block
    V8 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int f = (ch0 >> 1) + p;  // frequency index: channels 2f, 2f+1
      const float inv = __builtin_amdgcn_exp2f(-log2_temperature * (2.0f * (float)f / (float)num_feats));
      // v_sin_f32 / v_cos_f32 take revolutions and reduce the argument themselves (valid to +-256 revolutions; the
      // angle here is <= scale + a little, ~1 revolution); ~1e-6 absolute, far inside the f16 rounding of the result,
      // and ~10x fewer instructions than libm's sinf / cosf with their own range reduction
      const float rev = e * inv * 0.15915494309189535f;
      o[2 * p] = (E)__builtin_amdgcn_sinf(rev);
      o[2 * p + 1] = (E)__builtin_amdgcn_cosf(rev);
    }
    if (level_embed) {
      const V8 le = *reinterpret_cast<const V8*>(level_embed + c * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (E)((float)o[k] + (float)le[k]);  // fp16 + fp16 -> fp16, as the reference
    }
    *reinterpret_cast<V8*>(out + (size_t)b * out_batch_stride + (size_t)r * (2 * num_feats) + c * 8) = o;
  }
}

}  // namespace

namespace {

template <class E, class V8>
int sine_entry(void* stream, const float* ycum_dev, const float* xcum_dev, const void* level_embed_dev,
                               void* out_dev, int64_t out_batch_stride, int64_t B, int64_t H, int64_t W, int num_feats,
                               float temperature, float scale, float eps, float offset, int normalize) {
  if (!ycum_dev || !xcum_dev || !out_dev || B <= 0 || H <= 0 || W <= 0 || num_feats <= 0 || temperature <= 0.f)
    return CODETR_E_BADARG;
  if (num_feats % 8 != 0) return CODETR_E_UNSUPPORTED;
  if (B * H * W > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  const int64_t chunks = B * H * W * ((2 * num_feats) / 8);
  int64_t blocks = (chunks + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL((sine_pos_kernel<E, V8>), dim3((unsigned)blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream), ycum_dev,
                     xcum_dev, static_cast<const E*>(level_embed_dev), static_cast<E*>(out_dev),
                     out_batch_stride, (int)H, (int)W, num_feats, log2f(temperature), scale, eps, offset, normalize, chunks);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int codetr_sine_pos_tokens_f16(void* stream, const float* ycum_dev, const float* xcum_dev, const void* level_embed_dev,
                               void* out_dev, int64_t out_batch_stride, int64_t B, int64_t H, int64_t W, int num_feats,
                               float temperature, float scale, float eps, float offset, int normalize) {
  return sine_entry<_Float16, f16x8>(stream, ycum_dev, xcum_dev, level_embed_dev, out_dev, out_batch_stride, B, H, W, num_feats,
                             temperature, scale, eps, offset, normalize);
}

int codetr_sine_pos_tokens_bf16(void* stream, const float* ycum_dev, const float* xcum_dev, const void* level_embed_dev,
                               void* out_dev, int64_t out_batch_stride, int64_t B, int64_t H, int64_t W, int num_feats,
                               float temperature, float scale, float eps, float offset, int normalize) {
  return sine_entry<__bf16, bf16x8>(stream, ycum_dev, xcum_dev, level_embed_dev, out_dev, out_batch_stride, B, H, W, num_feats,
                             temperature, scale, eps, offset, normalize);
}

}  // extern "C"
