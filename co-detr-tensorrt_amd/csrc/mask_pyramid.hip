// Padding-mask pyramid of one batch: everything the transformer derives from img_masks, in one launch.
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
// Byte work on a 2.4 MB mask: grid (level, image, 2) -- z = 0 walks columns (mask + ycum), z = 1 walks rows (xcum);
// one thread per column / row, loads independent of the running sum.
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

__global__ __launch_bounds__(256) void mask_pyramid_kernel(const unsigned char* __restrict__ img, int B, int Hi, int Wi,
                                                           Levels lv, int L, int64_t S,
                                                           unsigned char* __restrict__ mask_flat,
                                                           float* __restrict__ ycum, float* __restrict__ xcum,
                                                           float* __restrict__ valid_counts) {
  const int l = blockIdx.x, b = blockIdx.y;
  const int H = lv.h[l], W = lv.w[l];
  const float sy = (float)Hi / (float)H, sx = (float)Wi / (float)W;
  const unsigned char* im = img + (size_t)b * Hi * Wi;
  const size_t base = (size_t)B * lv.start[l] + (size_t)b * H * W;  // level block [B, H, W] of the cum buffers
  if (blockIdx.z == 0) {
    for (int x = threadIdx.x; x < W; x += 256) {
      const int xs = src_index(x, sx, Wi);
      float run = 0.f;
      for (int y = 0; y < H; ++y) {
        const unsigned char m = im[(size_t)src_index(y, sy, Hi) * Wi + xs] != 0;
        run += m ? 0.f : 1.f;
        mask_flat[(size_t)b * S + lv.start[l] + (size_t)y * W + x] = m;
        ycum[base + (size_t)y * W + x] = run;
      }
      if (x == 0) valid_counts[((size_t)b * L + l) * 2 + 1] = run;  // valid rows of the first column
    }
  } else {
    for (int y = threadIdx.x; y < H; y += 256) {
      const unsigned char* row = im + (size_t)src_index(y, sy, Hi) * Wi;
      float run = 0.f;
      for (int x = 0; x < W; ++x) {
        run += row[src_index(x, sx, Wi)] != 0 ? 0.f : 1.f;
        xcum[base + (size_t)y * W + x] = run;
      }
      if (y == 0) valid_counts[((size_t)b * L + l) * 2 + 0] = run;  // valid columns of the first row
    }
  }
}

}  // namespace

extern "C" {

int codetr_mask_pyramid(void* stream, const void* img_mask_dev, int64_t B, int64_t H_img, int64_t W_img, int num_levels,
                        const int64_t* level_shapes_host, void* mask_flat_dev, float* ycum_dev, float* xcum_dev,
                        float* valid_counts_dev) {
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
  hipLaunchKernelGGL(mask_pyramid_kernel, dim3((unsigned)num_levels, (unsigned)B, 2), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned char*>(img_mask_dev), (int)B, (int)H_img,
                     (int)W_img, lv, num_levels, S, static_cast<unsigned char*>(mask_flat_dev), ycum_dev, xcum_dev,
                     valid_counts_dev);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // extern "C"
