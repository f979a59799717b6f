// Patchify for the Swin stem (mmdet PatchEmbed, reference codetr/swin.py:13, 567; equivalent source
// codetr/transformer_mmcv.py:100-210): the k x k / stride-k convolution is a GEMM over non-overlapping patches, so
// the stem runs as  [gather patches -> codetr_linear_* (K padded to 64) -> codetr_layernorm_*]  and its output is
// token-major [B, H/k * W/k, E] directly.  This file is the gather: NCHW image -> [tokens, kpad] rows with the
// (c, ky, kx) column order of conv.weight.view(E, C*k*k), zero-filled beyond C*k*k and beyond the image
// ("corner" padding of PatchEmbed: zeros to the right / bottom).  Replaces an NHWC implicit-GEMM convolution + two
// layout transposes + the flatten(2).transpose(1, 2) copy (1.4 ms per 8 images at 1920x1280) with one 16-bit copy
// kernel and a short-K X-stationary GEMM.
//
// 256 threads move 64 tokens: reads are lane = token (64 x 8 B contiguous per (c, ky) row), the rows are staged in
// LDS and leave as whole 128-byte lines.  Pure 16-bit data movement: one instantiation serves fp16 and bf16.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

constexpr int kTok = 64;  // tokens per workgroup

typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void patch_im2col_k4_kernel(const unsigned short* __restrict__ x,
                                                              unsigned short* __restrict__ out, int C, int H, int W,
                                                              int Hp, int Wp, long n_tokens) {
  __shared__ __attribute__((aligned(16))) unsigned short rows[kTok][64];
  const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
  const long t0 = (long)blockIdx.x * kTok;
  long t = t0 + lane;
  const bool live = t < n_tokens;
  t = live ? t : n_tokens - 1;
  const int tx = (int)(t % Wp);
  const long rest = t / Wp;
  const int ty = (int)(rest % Hp);
  const long b = rest / Hp;
  const bool fast = (W & 3) == 0;
  for (int p = grp; p < 16; p += 4) {  // piece p = (c, ky): 4 pixels = 8 bytes; pieces >= 4 C are the zero padding
    u16x4 v = {0, 0, 0, 0};
    const int c = p >> 2, ky = p & 3;
    const int y = ty * 4 + ky, x0 = tx * 4;
    if (c < C && y < H) {
      const unsigned short* src = x + ((b * C + c) * (long)H + y) * W + x0;
      if (fast && x0 + 3 < W) {
        v = *reinterpret_cast<const u16x4*>(src);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (x0 + i < W) v[i] = src[i];
      }
    }
    *reinterpret_cast<u16x4*>(&rows[lane][p * 4]) = v;
  }
  __syncthreads();
  // 64 tokens x 128 B = 512 pieces of 16 B, contiguous in `out`
  for (int e = tid; e < kTok * 8; e += 256) {
    const int tk = e >> 3, ch = e & 7;
    if (t0 + tk < n_tokens)
      *reinterpret_cast<u32x4*>(out + (t0 + tk) * 64 + ch * 8) = *reinterpret_cast<const u32x4*>(&rows[tk][ch * 8]);
  }
}

// k x k / stride s / zero padding p patches of a TOKEN-major map [B, H, W, C] -> rows [B * Ho * Wo, k*k*C] with the K
// axis ordered (ky, kx, c): whole C-vectors move as 16-byte pieces (the neck's extra 3x3 / stride-2 level as a GEMM).
__global__ __launch_bounds__(256) void im2col_tokens_kernel(const unsigned short* __restrict__ x,
                                                            unsigned short* __restrict__ out, int H, int W, int C8,
                                                            int Ho, int Wo, int k, int stride, int pad,
                                                            long total_pieces) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_pieces; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    long r = i / C8;
    const int tap = (int)(r % (k * k));
    r /= (k * k);
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const long b = r / Ho;
    const int ky = tap / k, kx = tap - ky * k;
    const int y = oy * stride + ky - pad, xx = ox * stride + kx - pad;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (y >= 0 && y < H && xx >= 0 && xx < W)
      v = *reinterpret_cast<const u32x4*>(x + (((b * H + y) * (long)W + xx) * C8 + c8) * 8);
    *reinterpret_cast<u32x4*>(out + i * 8) = v;
  }
}

}  // namespace

extern "C" {

int codetr_im2col_tokens_b16(void* stream, const void* x_dev, int64_t B, int64_t H, int64_t W, int64_t C, int k,
                             int stride, int pad, void* out_dev) {
  if (!x_dev || !out_dev || B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0)
    return CODETR_E_BADARG;
  if (C % 8 != 0) return CODETR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(out_dev)) & 15) return CODETR_E_BADARG;
  const int64_t Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return CODETR_E_BADARG;
  if (H > 0x7fffffffLL || W > 0x7fffffffLL || C > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  const int64_t pieces = B * Ho * Wo * k * k * (C / 8);
  int64_t blocks = (pieces + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(im2col_tokens_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(x_dev), static_cast<unsigned short*>(out_dev), (int)H, (int)W,
                     (int)(C / 8), (int)Ho, (int)Wo, k, stride, pad, (long)pieces);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

int codetr_patch_im2col_b16(void* stream, const void* x_dev, int64_t B, int C, int64_t H, int64_t W, int k, int kpad,
                            void* out_dev) {
  if (!x_dev || !out_dev || B <= 0 || C <= 0 || H <= 0 || W <= 0) return CODETR_E_BADARG;
  if (k != 4 || kpad != 64 || C * k * k > kpad) return CODETR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x_dev) & 7) || (reinterpret_cast<uintptr_t>(out_dev) & 15)) return CODETR_E_BADARG;
  const int64_t Hp = (H + k - 1) / k, Wp = (W + k - 1) / k;
  const int64_t n = B * Hp * Wp;
  const int64_t blocks = (n + kTok - 1) / kTok;
  if (blocks > 0x7fffffffLL || H > 0x7fffffffLL || W > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(patch_im2col_k4_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(x_dev), static_cast<unsigned short*>(out_dev), C, (int)H,
                     (int)W, (int)Hp, (int)Wp, (long)n);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // extern "C"
