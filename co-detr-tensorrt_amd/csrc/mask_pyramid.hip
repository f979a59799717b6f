// Padding-mask pyramid of one batch: everything the transformer derives from img_masks, in two launches.
//
// Replaces, per pyramid level, the ATen sequence of CoDINOHead.forward / CoDinoTransformer.forward
// (reference codetr/co_dino_head.py:155: F.interpolate(img_masks[None], size=feat.shape[-2:]).to(bool);
//  codetr/positional_encoding.py:78-79: not_mask.cumsum(1), not_mask.cumsum(2);
//  codetr/transformer.py:384-399 get_valid_ratio: sum(~mask[:, :, 0]), sum(~mask[:, 0, :]);
//  codetr/transformer.py:513-520: mask.flatten(1) + cat over levels)
// -- ~14 small kernels per level, two of them ATen's outer-dim scan at 30 us -- with
//   mask_flat[b, start_l + y*W_l + x]   the level masks already concatenated (uint8, 1 = padding)
//   ycum / xcum                         running counts of valid pixels down each column / along each row (fp32,
//                                       exact small integers), level l stored as [B, H_l, W_l] at offset B*start_l
//   valid_counts[b, l, (w, h)]          valid pixels in the first row / first column (fp32)
// Nearest-neighbour source index as ATen's upsample_nearest2d: min(int(floorf(dst * (float)in / out)), in - 1).
//
// Byte work, one thread per token in both passes (a first version that walked each column / row with one thread
// -- 160 dependent iterations at the stride-8 level -- took 267 us):
//   pass 1  level_mask_kernel : mask_flat[b, s] = img[b, src(y), src(x)] != 0
//   pass 2  level_cums_kernel : one wave per level row (ballot + popcount prefix -> xcum) and one per 64-column
//                               strip (running count down the strip -> ycum); a thread-per-token version that
//                               re-summed its column and row (<= 400 byte loads per thread) still took 93 us
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

constexpr int kMaxLevels = 8;
struct Levels {
  int h[kMaxLevels], w[kMaxLevels];
  int64_t start[kMaxLevels];
};

__device__ __forceinline__ int src_index(int dst, float scale, int in) {
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

__device__ __forceinline__ int level_of(const Levels& lv, int L, int64_t s) {
  int l = L - 1;
  while (l > 0 && s < lv.start[l]) --l;
  return l;
}

// EB = bytes per mask element: 1 (bool / uint8), 2 (fp16 / bf16) or 4 (fp32); "non-zero" ignores the sign bit of the
// floating-point forms (-0.0 is zero, NaN is not), i.e. `mask != 0` without a separate comparison kernel
template <int EB>
__global__ __launch_bounds__(256) void level_mask_kernel(const unsigned char* __restrict__ img, int Hi, int Wi, Levels lv,
                                                         int L, int64_t S, int64_t total,
                                                         unsigned char* __restrict__ mask_flat) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / S);
  const int64_t s = i - (int64_t)b * S;
  const int l = level_of(lv, L, s);
  const int r = (int)(s - lv.start[l]);
  const int H = lv.h[l], W = lv.w[l];
  const int y = r / W, x = r - y * W;
  const float sy = (float)Hi / (float)H, sx = (float)Wi / (float)W;
  const size_t e = ((size_t)b * Hi + src_index(y, sy, Hi)) * Wi + src_index(x, sx, Wi);
  if (EB == 1) mask_flat[i] = img[e] != 0;
  else if (EB == 2) mask_flat[i] = (reinterpret_cast<const unsigned short*>(img)[e] & 0x7fffu) != 0;
  else mask_flat[i] = (reinterpret_cast<const unsigned*>(img)[e] & 0x7fffffffu) != 0;
}

// One wave per task.  Tasks of image b: first every row of every level (xcum: ballot + popcount prefix, 64 columns
// per step), then every 64-column strip of every level (ycum: running count down the strip, coalesced 64-byte loads).
__global__ __launch_bounds__(64) void level_cums_kernel(const unsigned char* __restrict__ mask_flat, int B, Levels lv,
                                                        int L, int64_t S, int rows_total, float* __restrict__ ycum,
                                                        float* __restrict__ xcum, float* __restrict__ valid_counts) {
  const int b = blockIdx.y, lane = threadIdx.x;
  int task = blockIdx.x;
  if (task < rows_total) {
    int l = 0;
    while (task >= lv.h[l]) task -= lv.h[l++];
    const int y = task, H = lv.h[l], W = lv.w[l];
    const unsigned char* row = mask_flat + (size_t)b * S + lv.start[l] + (size_t)y * W;
    float* out = xcum + (size_t)B * lv.start[l] + ((size_t)b * H + y) * W;
    int carry = 0;
    for (int x0 = 0; x0 < W; x0 += 64) {
      const int x = x0 + lane;
      const bool valid = x < W && row[x] == 0;
      const unsigned long long bal = __ballot(valid);
      const int incl = __popcll(bal & (~0ull >> (63 - lane)));
      if (x < W) out[x] = (float)(carry + incl);
      carry += __popcll(bal);
    }
    if (y == 0 && lane == 0) valid_counts[((size_t)b * L + l) * 2 + 0] = (float)carry;  // valid columns of the first row
    return;
  }
  task -= rows_total;
  int l = 0;
  while (task >= (lv.w[l] + 63) / 64) task -= (lv.w[l++] + 63) / 64;
  const int H = lv.h[l], W = lv.w[l];
  const int x = task * 64 + lane;
  if (x >= W) return;
  const unsigned char* col = mask_flat + (size_t)b * S + lv.start[l] + x;
  float* out = ycum + (size_t)B * lv.start[l] + (size_t)b * H * W + x;
  int run = 0;
#pragma unroll 8
  for (int y = 0; y < H; ++y) {
    run += col[(size_t)y * W] == 0;
    out[(size_t)y * W] = (float)run;
  }
  if (x == 0) valid_counts[((size_t)b * L + l) * 2 + 1] = (float)run;  // valid rows of the first column
}

}  // namespace

extern "C" {

int codetr_mask_pyramid(void* stream, const void* img_mask_dev, int64_t B, int64_t H_img, int64_t W_img, int num_levels,
                        const int64_t* level_shapes_host, void* mask_flat_dev, float* ycum_dev, float* xcum_dev,
                        float* valid_counts_dev, int mask_elem_bytes) {
  if (mask_elem_bytes != 1 && mask_elem_bytes != 2 && mask_elem_bytes != 4) return CODETR_E_UNSUPPORTED;
  if (!img_mask_dev || !level_shapes_host || !mask_flat_dev || !ycum_dev || !xcum_dev || !valid_counts_dev || B <= 0 ||
      H_img <= 0 || W_img <= 0 || num_levels <= 0)
    return CODETR_E_BADARG;
  if (num_levels > kMaxLevels) return CODETR_E_UNSUPPORTED;
  if (B > 65535 || H_img > 0x7fffffffLL || W_img > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  Levels lv{};
  int64_t S = 0;
  for (int l = 0; l < num_levels; ++l) {
    const int64_t h = level_shapes_host[2 * l], w = level_shapes_host[2 * l + 1];
    if (h <= 0 || w <= 0) return CODETR_E_BADARG;
    if (h > 0x7fffffffLL || w > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
    lv.h[l] = (int)h;
    lv.w[l] = (int)w;
    lv.start[l] = S;
    S += h * w;
  }
  const int64_t total = B * S;
  if ((total + 255) / 256 > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto mk = mask_elem_bytes == 1 ? level_mask_kernel<1> : (mask_elem_bytes == 2 ? level_mask_kernel<2> : level_mask_kernel<4>);
  hipLaunchKernelGGL(mk, grid, block, 0, st, static_cast<const unsigned char*>(img_mask_dev), (int)H_img,
                     (int)W_img, lv, num_levels, S, total, static_cast<unsigned char*>(mask_flat_dev));
  int rows_total = 0, strips_total = 0;
  for (int l = 0; l < num_levels; ++l) {
    rows_total += lv.h[l];
    strips_total += (lv.w[l] + 63) / 64;
  }
  hipLaunchKernelGGL(level_cums_kernel, dim3((unsigned)(rows_total + strips_total), (unsigned)B), dim3(64), 0, st,
                     static_cast<const unsigned char*>(mask_flat_dev), (int)B, lv, num_levels, S, rows_total, ycum_dev,
                     xcum_dev, valid_counts_dev);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // extern "C"
