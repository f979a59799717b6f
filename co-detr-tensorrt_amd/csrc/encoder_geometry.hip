// Token geometry of the deformable encoder, one launch: reference points, their per-level scaling, the two-stage
// proposals and the keep / drop state of every token.
//
// Replaces (f16 inference path)
//   get_reference_points                     reference codetr/transformer.py:280-305   (~6 ATen launches per level + cat)
//   reference_points[:, :, None] * valid_ratios[:, None]                  codetr/transformer.py:530
//   make_encoder_output_proposals_export     codetr/transformer.py:331-339 (cat, 1 - p, div, log)
//   apply_mask_to_proposal_and_memory        codetr/transformer.py:351-380 (proposal half; the `memory * total_mask`
//                                            half becomes row state 2 of codetr_linear_*'s row mask on enc_output)
// -- about 60 launches of ~5 us, plus two full passes over memory [B, S, 256].
//
// Per token (b, s) of level l at (y, x):
//   ref        = ( f16((x + .5) / f16(vr_w * W_l)),  f16((y + .5) / f16(vr_h * H_l)) )       the reference's fp16 roundings
//   ref_lvl[k] = f16(ref * vr[b, k])                                                         k = 0..L-1
//   prop       = f16(logit(p)),  p = (ref_x, ref_y, w, w),  w = f16(0.05 * 2^l)              logit in fp32
//   keep       = all(-4.6 < prop < 4.6) and not padding
//   proposals  = keep ? prop : (prop finite ? finfo(f16).max : NaN)      == prop * keep + (1 - keep) * finfo.max
//   row_state  = keep ? 0 : 2
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

constexpr int kMaxLevels = 8;
struct Levels {
  int h[kMaxLevels], w[kMaxLevels];
  int64_t start[kMaxLevels];
};
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// 16-bit storage <-> float; BF: bfloat16 (the bf16 model's instantiations: same formulas, that type's roundings and its
// finfo.max), else fp16
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
__device__ __forceinline__ float h_round(float v) { return ld16<BF>(st16<BF>(v)); }

template <bool BF>
__global__ __launch_bounds__(256) void encoder_geometry_kernel(const unsigned short* __restrict__ valid_ratios,
                                                               const unsigned char* __restrict__ mask_flat, Levels lv,
                                                               int L, int64_t S, int64_t total,
                                                               unsigned short* __restrict__ ref,
                                                               unsigned short* __restrict__ ref_lvl,
                                                               unsigned short* __restrict__ proposals,
                                                               unsigned char* __restrict__ row_state) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / S);
  const int64_t s = i - (int64_t)b * S;
  int l = L - 1;
  while (l > 0 && s < lv.start[l]) --l;
  const int r = (int)(s - lv.start[l]);
  const int W = lv.w[l], H = lv.h[l];
  const int y = r / W, x = r - y * W;
  const unsigned short* vr = valid_ratios + (size_t)b * L * 2;
  const float rx = h_round<BF>(((float)x + 0.5f) / h_round<BF>(ld16<BF>(vr[2 * l]) * (float)W));
  const float ry = h_round<BF>(((float)y + 0.5f) / h_round<BF>(ld16<BF>(vr[2 * l + 1]) * (float)H));
  *reinterpret_cast<u16x2*>(ref + i * 2) = u16x2{st16<BF>(rx), st16<BF>(ry)};
  for (int k = 0; k < L; ++k)
    *reinterpret_cast<u16x2*>(ref_lvl + (i * L + k) * 2) =
        u16x2{st16<BF>(rx * ld16<BF>(vr[2 * k])), st16<BF>(ry * ld16<BF>(vr[2 * k + 1]))};
  const float wl = h_round<BF>(0.05f * (float)(1 << l));
  const float p[4] = {rx, ry, wl, wl};
  const float lo = h_round<BF>(-4.6f), hi = h_round<BF>(4.6f);
  bool keep = mask_flat[i] == 0, finite = true;
  float of[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float qf = h_round<BF>(logf(p[k] / (1.0f - p[k])));
    keep = keep && qf > lo && qf < hi;  // NaN compares false, as in the reference
    finite = finite && (qf - qf == 0.0f);
    of[k] = qf;
  }
  if (!keep) {
    // prop * 0 + 1 * finfo.max, element-wise: finite -> max, inf / NaN -> NaN
    const float fmax_t = BF ? __uint_as_float(0x7f7f0000u) : 65504.0f;   // finfo(bf16).max / finfo(f16).max
#pragma unroll
    for (int k = 0; k < 4; ++k) of[k] = (of[k] - of[k] == 0.0f) ? fmax_t : __builtin_nanf("");
  }
  *reinterpret_cast<u16x4*>(proposals + i * 4) = u16x4{st16<BF>(of[0]), st16<BF>(of[1]), st16<BF>(of[2]), st16<BF>(of[3])};
  row_state[i] = keep ? 0 : 2;
}

// out[r] = max over the C columns of x[r, :] (NaN wins, as torch.max): the two-stage ranking score of a token
template <bool BF>
__global__ __launch_bounds__(256) void row_max_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ out,
                                                      int64_t rows, int C) {
  // 16 lanes per row, 8 columns per lane per step
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  float m = -__builtin_inff();
  bool nan = false;
  if (row < rows) {
    const unsigned short* xr = x + row * C;
    if ((C & 7) == 0) {
      for (int c = sub * 8; c < C; c += 128) {
        const u16x8 v = *reinterpret_cast<const u16x8*>(xr + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = ld16<BF>(v[e]);
          nan = nan || f != f;
          m = f > m ? f : m;
        }
      }
    } else {
      for (int c = sub; c < C; c += 16) {
        const float f = ld16<BF>(xr[c]);
        nan = nan || f != f;
        m = f > m ? f : m;
      }
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o);
    const int n2 = __shfl_xor((int)nan, o);
    m = m2 > m ? m2 : m;
    nan = nan || n2;
  }
  if (row < rows && sub == 0) out[row] = st16<BF>(nan ? __builtin_nanf("") : m);
}

}  // namespace

namespace {


template <bool BF>
int encoder_geometry_impl(void* stream, const void* valid_ratios_dev, const void* mask_flat_dev, int64_t B,
                                int num_levels, const int64_t* level_shapes_host, void* reference_points_dev,
                                void* reference_by_level_dev, void* proposals_dev, void* row_state_dev) {
  if (!valid_ratios_dev || !mask_flat_dev || !level_shapes_host || !reference_points_dev || !reference_by_level_dev ||
      !proposals_dev || !row_state_dev || B <= 0 || num_levels <= 0)
    return CODETR_E_BADARG;
  if (num_levels > kMaxLevels) return CODETR_E_UNSUPPORTED;
  Levels lv{};
  int64_t S = 0;
  for (int l = 0; l < num_levels; ++l) {
    const int64_t h = level_shapes_host[2 * l], w = level_shapes_host[2 * l + 1];
    if (h <= 0 || w <= 0) return CODETR_E_BADARG;
    if (h > 0x7fffffffLL || w > 0x7fffffffLL || h * w > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
    lv.h[l] = (int)h;
    lv.w[l] = (int)w;
    lv.start[l] = S;
    S += h * w;
  }
  const int64_t total = B * S;
  if ((total + 255) / 256 > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(encoder_geometry_kernel<BF>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(valid_ratios_dev),
                     static_cast<const unsigned char*>(mask_flat_dev), lv, num_levels, S, total,
                     static_cast<unsigned short*>(reference_points_dev), static_cast<unsigned short*>(reference_by_level_dev),
                     static_cast<unsigned short*>(proposals_dev), static_cast<unsigned char*>(row_state_dev));
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

template <bool BF>
int row_max_impl(void* stream, const void* x_dev, void* out_dev, int64_t rows, int64_t C) {
  if (!x_dev || !out_dev || rows <= 0 || C <= 0) return CODETR_E_BADARG;
  if (C > 0x7fffffffLL || (rows * 16 + 255) / 256 > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(row_max_kernel<BF>, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(x_dev), static_cast<unsigned short*>(out_dev),
                     rows, (int)C);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int codetr_encoder_geometry_f16(void* stream, const void* valid_ratios_dev, const void* mask_flat_dev, int64_t B,
                                int num_levels, const int64_t* level_shapes_host, void* reference_points_dev,
                                void* reference_by_level_dev, void* proposals_dev, void* row_state_dev) {
  return encoder_geometry_impl<false>(stream, valid_ratios_dev, mask_flat_dev, B, num_levels, level_shapes_host,
                                      reference_points_dev, reference_by_level_dev, proposals_dev, row_state_dev);
}
int codetr_encoder_geometry_bf16(void* stream, const void* valid_ratios_dev, const void* mask_flat_dev, int64_t B,
                                 int num_levels, const int64_t* level_shapes_host, void* reference_points_dev,
                                 void* reference_by_level_dev, void* proposals_dev, void* row_state_dev) {
  return encoder_geometry_impl<true>(stream, valid_ratios_dev, mask_flat_dev, B, num_levels, level_shapes_host,
                                     reference_points_dev, reference_by_level_dev, proposals_dev, row_state_dev);
}
int codetr_row_max_f16(void* stream, const void* x_dev, void* out_dev, int64_t rows, int64_t C) {
  return row_max_impl<false>(stream, x_dev, out_dev, rows, C);
}
int codetr_row_max_bf16(void* stream, const void* x_dev, void* out_dev, int64_t rows, int64_t C) {
  return row_max_impl<true>(stream, x_dev, out_dev, rows, C);
}

}  // extern "C"
