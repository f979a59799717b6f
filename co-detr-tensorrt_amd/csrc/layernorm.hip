// LayerNorm over the last dimension for MI355X (gfx950): y = (x - mean) / sqrt(var + eps) * gamma + beta.
//
// Every nn.LayerNorm of the hot path (Swin norm1/norm2/stage norms/patch-merging norms, encoder and
// decoder norms, enc_output_norm: reference codetr/swin.py:331,345,627; codetr/transformer_mmcv.py:647;
// codetr/transformer.py:153,452) with C in {192, 256, 384, 768, 1536, 3072}.  Pure HBM streaming:
// one read and one write of the activation, fp32 statistics.
//
// Mapping: a row is served by G lanes (G = 32 when C*2 bytes <= 512, else 64), each lane owning
// 16-byte chunks (8 halves) strided by G; a 256-thread workgroup handles 256/G rows at a time and
// grid-strides.  The row stays in registers between the mean and the variance pass (two-pass,
// no E[x^2]-E[x]^2 cancellation); the cross-lane sums are wave shuffles (DPP / permlane), no LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

constexpr int kThreads = 256;
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct LnHalf {
  __device__ static float up(short b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
  }
  __device__ static short down(float v) {
    _Float16 h = (_Float16)v;
    short b;
    __builtin_memcpy(&b, &h, 2);
    return b;
  }
};
struct LnBf16 {
  __device__ static float up(short b) { return __uint_as_float(((unsigned)(unsigned short)b) << 16); }
  __device__ static short down(float v) {   // v_cvt_pk_bf16_f32 (round to nearest even, quiet NaN)
    const __bf16 h = (__bf16)v;
    return __builtin_bit_cast(short, h);
  }
};

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// G lanes per row, NCH chunks of 8 elements per lane (G * NCH * 8 >= C)
// MERGE: the input row is gathered on the fly from the 2 x 2 neighbourhood of a token map [B, H, W, Cs] -- Swin's
// PatchMerging (reference codetr/transformer_mmcv.py:213-316: nn.Unfold(2, stride 2) + LayerNorm(4 Cs)) with the 4 Cs axis
// ordered (ky, kx, c) -- so the merged map is never written un-normalised: row r = (b, y2, x2), chunk ch -> sub-pixel
// k = ch / (Cs/8) = ky*2 + kx, source token (2 y2 + ky, 2 x2 + kx), zeros beyond an odd H / W (F.pad).
struct MergeGeom {
  int H, W, Cs, H2, W2;
};

template <class T, int G, int NCH, bool MERGE = false>
__global__ __launch_bounds__(kThreads) void layernorm_kernel(const short* __restrict__ x, const short* __restrict__ gamma,
                                                             const short* __restrict__ beta, short* __restrict__ y,
                                                             int64_t rows, int C, float eps, MergeGeom mg = MergeGeom{}) {
  constexpr int ROWS_PER_BLOCK = kThreads / G;
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nchunks = C >> 3;
  const float inv_c = 1.0f / (float)C;
  // gamma / beta chunks of this lane are row-invariant: keep them in registers
  s16x8 gw[NCH], gb[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = sub + c * G;
    if (ch < nchunks) {
      gw[c] = *reinterpret_cast<const s16x8*>(gamma + ch * 8);
      gb[c] = *reinterpret_cast<const s16x8*>(beta + ch * 8);
    }
  }
  for (int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + rloc; row < rows; row += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
    const short* xr = x + row * C;
    int mb = 0, my = 0, mx = 0;
    if (MERGE) {
      mb = (int)(row / ((int64_t)mg.H2 * mg.W2));
      const int rr = (int)(row - (int64_t)mb * mg.H2 * mg.W2);
      my = rr / mg.W2;
      mx = rr - my * mg.W2;
    }
    s16x8 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = sub + c * G;
      if (ch < nchunks) {
        if (MERGE) {
          const int cpc = mg.Cs >> 3, k = ch / cpc, within = ch - k * cpc;
          const int sy = 2 * my + (k >> 1), sx = 2 * mx + (k & 1);
          v[c] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
          if (sy < mg.H && sx < mg.W)
            v[c] = *reinterpret_cast<const s16x8*>(x + (((int64_t)mb * mg.H + sy) * mg.W + sx) * mg.Cs + within * 8);
        } else {
          v[c] = *reinterpret_cast<const s16x8*>(xr + ch * 8);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += T::up(v[c][e]);
      }
    }
    const float mean = group_sum<G>(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = sub + c * G;
      if (ch < nchunks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = T::up(v[c][e]) - mean;
          q = fmaf(d, d, q);
        }
      }
    }
    const float rstd = rsqrtf(group_sum<G>(q) * inv_c + eps);
    short* yr = y + row * C;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = sub + c * G;
      if (ch < nchunks) {
        s16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = T::down(fmaf((T::up(v[c][e]) - mean) * rstd, T::up(gw[c][e]), T::up(gb[c][e])));
        *reinterpret_cast<s16x8*>(yr + ch * 8) = o;
      }
    }
  }
}

template <class T, int G, int NCH, bool MERGE = false>
int launch_cfg(hipStream_t st, const void* x, const void* g, const void* b, void* y, int64_t rows, int C, float eps,
               MergeGeom mg = MergeGeom{}) {
  constexpr int RPB = kThreads / G;
  int64_t blocks = (rows + RPB - 1) / RPB;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 workgroups per CU
  hipLaunchKernelGGL((layernorm_kernel<T, G, NCH, MERGE>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                     static_cast<const short*>(x), static_cast<const short*>(g), static_cast<const short*>(b),
                     static_cast<short*>(y), rows, C, eps, mg);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

template <class T>
int launch_merge(hipStream_t st, const void* x, const void* g, const void* b, void* y, int64_t B, int64_t H, int64_t W,
                 int64_t Cs, float eps) {
  if (!x || !g || !b || !y || B <= 0 || H <= 0 || W <= 0 || Cs <= 0) return CODETR_E_BADARG;
  const int64_t C = 4 * Cs;
  if (Cs % 8 != 0 || C > 4096) return CODETR_E_UNSUPPORTED;
  if (H > 0x7fffffffLL || W > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  MergeGeom mg{(int)H, (int)W, (int)Cs, (int)((H + 1) / 2), (int)((W + 1) / 2)};
  const int64_t rows = B * mg.H2 * mg.W2;
  const int nch = (int)(C / 8);
  if (nch <= 32) return launch_cfg<T, 32, 1, true>(st, x, g, b, y, rows, (int)C, eps, mg);
  if (nch <= 64) return launch_cfg<T, 64, 1, true>(st, x, g, b, y, rows, (int)C, eps, mg);
  if (nch <= 128) return launch_cfg<T, 64, 2, true>(st, x, g, b, y, rows, (int)C, eps, mg);
  if (nch <= 192) return launch_cfg<T, 64, 3, true>(st, x, g, b, y, rows, (int)C, eps, mg);
  if (nch <= 256) return launch_cfg<T, 64, 4, true>(st, x, g, b, y, rows, (int)C, eps, mg);
  if (nch <= 384) return launch_cfg<T, 64, 6, true>(st, x, g, b, y, rows, (int)C, eps, mg);
  return launch_cfg<T, 64, 8, true>(st, x, g, b, y, rows, (int)C, eps, mg);
}

template <class T>
int launch(hipStream_t st, const void* x, const void* g, const void* b, void* y, int64_t rows, int64_t C, float eps) {
  if (!x || !g || !b || !y || rows <= 0 || C <= 0) return CODETR_E_BADARG;
  if (C % 8 != 0 || C > 4096) return CODETR_E_UNSUPPORTED;
  const int nch = (int)(C / 8);
  if (nch <= 32) return launch_cfg<T, 32, 1>(st, x, g, b, y, rows, (int)C, eps);
  if (nch <= 64) return launch_cfg<T, 64, 1>(st, x, g, b, y, rows, (int)C, eps);
  if (nch <= 128) return launch_cfg<T, 64, 2>(st, x, g, b, y, rows, (int)C, eps);
  if (nch <= 192) return launch_cfg<T, 64, 3>(st, x, g, b, y, rows, (int)C, eps);
  if (nch <= 256) return launch_cfg<T, 64, 4>(st, x, g, b, y, rows, (int)C, eps);
  if (nch <= 384) return launch_cfg<T, 64, 6>(st, x, g, b, y, rows, (int)C, eps);
  return launch_cfg<T, 64, 8>(st, x, g, b, y, rows, (int)C, eps);
}

}  // namespace

extern "C" {

int codetr_layernorm_f16(void* stream, const void* x_dev, const void* gamma_dev, const void* beta_dev, void* y_dev,
                         int64_t rows, int64_t C, float eps) {
  return launch<LnHalf>(static_cast<hipStream_t>(stream), x_dev, gamma_dev, beta_dev, y_dev, rows, C, eps);
}

int codetr_layernorm_bf16(void* stream, const void* x_dev, const void* gamma_dev, const void* beta_dev, void* y_dev,
                          int64_t rows, int64_t C, float eps) {
  return launch<LnBf16>(static_cast<hipStream_t>(stream), x_dev, gamma_dev, beta_dev, y_dev, rows, C, eps);
}

int codetr_patch_merge_layernorm_f16(void* stream, const void* x_dev, const void* gamma_dev, const void* beta_dev,
                                     void* y_dev, int64_t B, int64_t H, int64_t W, int64_t C, float eps) {
  return launch_merge<LnHalf>(static_cast<hipStream_t>(stream), x_dev, gamma_dev, beta_dev, y_dev, B, H, W, C, eps);
}

int codetr_patch_merge_layernorm_bf16(void* stream, const void* x_dev, const void* gamma_dev, const void* beta_dev,
                                      void* y_dev, int64_t B, int64_t H, int64_t W, int64_t C, float eps) {
  return launch_merge<LnBf16>(static_cast<hipStream_t>(stream), x_dev, gamma_dev, beta_dev, y_dev, B, H, W, C, eps);
}

}  // extern "C"
