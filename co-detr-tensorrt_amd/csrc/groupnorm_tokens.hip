// GroupNorm over token-major activations [B, HW, C] for MI355X (gfx950): the ChannelMapper neck's
// GN(32, 256) (mmdet ChannelMapper built at reference codetr/codetr.py:53-54 from configs lsj:40-47) applied
// where the data already lives -- the output of the 1x1 "conv" run as a GEMM over tokens -- and written
// straight into the level's slice of the flattened multi-level feature map [B, S, C] that the deformable
// encoder consumes (reference transformer.py:508-519 builds that tensor with flatten/transpose/cat copies).
//
// 8 channels per group = one 16-byte chunk per lane: 32 lanes cover a 256-channel token.
//   pass 1  gn_partial_kernel : per workgroup, fp32 (sum, sum of squares) of each group over a slab of rows
//   pass 2  gn_finalize_kernel: fp64 reduction of the partials -> (mean, rstd) per (image, group)
//   pass 3  gn_apply_kernel   : y = (x - mean) * rstd * gamma + beta, 16 B in / 16 B out per lane
// HBM traffic: x is read twice and written once (x of one level fits the 256 MB Infinity Cache, so the second
// read is served on-die at 1920x1280).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerSlab = 128;  // rows reduced by one workgroup in pass 1 (600 workgroups at the stride-8 level)
typedef short s16x8 __attribute__((ext_vector_type(8)));

// BF = false: fp16 storage, true: bf16 storage (fp32 statistics and arithmetic either way)
template <bool BF>
__device__ __forceinline__ float h2f(short b) {
  if (BF) return __uint_as_float(((unsigned)(unsigned short)b) << 16);
  _Float16 h;
  __builtin_memcpy(&h, &b, 2);
  return (float)h;
}
template <bool BF>
__device__ __forceinline__ short f2h(float v) {
  short b;
  if (BF) {
    const __bf16 h = (__bf16)v;  // round-to-nearest-even
    __builtin_memcpy(&b, &h, 2);
    return b;
  }
  const _Float16 h = (_Float16)v;
  __builtin_memcpy(&b, &h, 2);
  return b;
}

// grid (nslab, B); partial[b][slab][group][2]
template <bool BF>
__global__ __launch_bounds__(kThreads) void gn_partial_kernel(const short* __restrict__ x, float* __restrict__ partial,
                                                              int HW, int C, int nslab) {
  const int groups = C >> 3;                    // lanes per row
  const int rows_par = kThreads / groups;       // rows handled in parallel
  const int gidx = threadIdx.x % groups, rpar = threadIdx.x / groups;
  const int b = blockIdx.y, slab = blockIdx.x;
  const int r0 = slab * kRowsPerSlab;
  const int r1 = min(r0 + kRowsPerSlab, HW);
  const short* xb = x + (size_t)b * HW * C;
  float s = 0.f, q = 0.f;
  if (threadIdx.x < rows_par * groups) {
    for (int r = r0 + rpar; r < r1; r += rows_par) {
      const s16x8 v = *reinterpret_cast<const s16x8*>(xb + (size_t)r * C + gidx * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = h2f<BF>(v[e]);
        s += f;
        q = fmaf(f, f, q);
      }
    }
  }
  __shared__ float red[kThreads * 2];
  red[threadIdx.x * 2] = s;
  red[threadIdx.x * 2 + 1] = q;
  __syncthreads();
  if (threadIdx.x < groups) {
    float ts = 0.f, tq = 0.f;
    for (int p = 0; p < rows_par; ++p) {
      ts += red[(p * groups + threadIdx.x) * 2];
      tq += red[(p * groups + threadIdx.x) * 2 + 1];
    }
    float* dst = partial + (((size_t)b * nslab + slab) * groups + threadIdx.x) * 2;
    dst[0] = ts;
    dst[1] = tq;
  }
}

// one wave per (b, group): lanes stride over the slabs, fp64 butterfly at the end (a single thread walking 600
// dependent loads cost 22 us per level)
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats,
                                                         int B, int groups, int nslab, double count, float eps) {
  const int i = blockIdx.x;
  const int b = i / groups, g = i % groups;
  double s = 0.0, q = 0.0;
  for (int sl = threadIdx.x; sl < nslab; sl += 64) {
    const float* p = partial + (((size_t)b * nslab + sl) * groups + g) * 2;
    s += (double)p[0];
    q += (double)p[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  if (threadIdx.x == 0) {
    const double mean = s / count;
    double var = q / count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

template <bool BF>
__global__ __launch_bounds__(kThreads) void gn_apply_kernel(const short* __restrict__ x, const float* __restrict__ stats,
                                                            const short* __restrict__ gamma, const short* __restrict__ beta,
                                                            short* __restrict__ out, int64_t out_batch_stride, int HW,
                                                            int C, int64_t total_chunks) {
  const int groups = C >> 3;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total_chunks; i += (int64_t)gridDim.x * kThreads) {
    const int g = (int)(i % groups);
    const int64_t row = i / groups;  // b*HW + r
    const int b = (int)(row / HW);
    const int r = (int)(row - (int64_t)b * HW);
    const float mean = stats[2 * (b * groups + g)], rstd = stats[2 * (b * groups + g) + 1];
    const s16x8 v = *reinterpret_cast<const s16x8*>(x + row * C + g * 8);
    const s16x8 gw = *reinterpret_cast<const s16x8*>(gamma + g * 8);
    const s16x8 gb = *reinterpret_cast<const s16x8*>(beta + g * 8);
    s16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2h<BF>(fmaf((h2f<BF>(v[e]) - mean) * rstd, h2f<BF>(gw[e]), h2f<BF>(gb[e])));
    *reinterpret_cast<s16x8*>(out + (size_t)b * out_batch_stride + (size_t)r * C + g * 8) = o;
  }
}

}  // namespace

namespace {

template <bool BF>
int gn_entry(void* stream, const void* x_dev, const void* gamma_dev, const void* beta_dev,
                                void* out_dev, int64_t out_batch_stride, void* workspace_dev, int64_t B, int64_t HW,
                                int64_t C, int groups, float eps) {
  if (!x_dev || !gamma_dev || !beta_dev || !out_dev || !workspace_dev || B <= 0 || HW <= 0 || C <= 0)
    return CODETR_E_BADARG;
  if (groups <= 0 || C != (int64_t)groups * 8 || groups > kThreads || kThreads % groups != 0) return CODETR_E_UNSUPPORTED;
  if (HW > 0x7fffffffLL || B > 65535) return CODETR_E_TOO_LARGE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nslab = (int)((HW + kRowsPerSlab - 1) / kRowsPerSlab);
  float* partial = static_cast<float*>(workspace_dev);
  float* stats = partial + (size_t)B * nslab * groups * 2;
  hipLaunchKernelGGL(gn_partial_kernel<BF>, dim3(nslab, (unsigned)B), dim3(kThreads), 0, st,
                     static_cast<const short*>(x_dev), partial, (int)HW, (int)C, nslab);
  const int nstat = (int)(B * groups);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)nstat), dim3(64), 0, st, partial, stats, (int)B, groups,
                     nslab, (double)HW * 8.0, eps);
  const int64_t chunks = B * HW * groups;
  int64_t blocks = (chunks + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gn_apply_kernel<BF>, dim3((unsigned)blocks), dim3(kThreads), 0, st, static_cast<const short*>(x_dev),
                     stats, static_cast<const short*>(gamma_dev), static_cast<const short*>(beta_dev),
                     static_cast<short*>(out_dev), out_batch_stride, (int)HW, (int)C, chunks);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int64_t codetr_groupnorm_tokens_workspace_bytes(int64_t B, int64_t HW, int64_t C) {
  if (B <= 0 || HW <= 0 || C <= 0) return 0;
  const int64_t groups = C / 8, nslab = (HW + kRowsPerSlab - 1) / kRowsPerSlab;
  return (B * nslab * groups * 2 + B * groups * 2) * (int64_t)sizeof(float);
}

int codetr_groupnorm_tokens_f16(void* stream, const void* x_dev, const void* gamma_dev, const void* beta_dev,
                                void* out_dev, int64_t out_batch_stride, void* workspace_dev, int64_t B, int64_t HW,
                                int64_t C, int groups, float eps) {
  return gn_entry<false>(stream, x_dev, gamma_dev, beta_dev, out_dev, out_batch_stride, workspace_dev, B, HW, C, groups,
                       eps);
}

int codetr_groupnorm_tokens_bf16(void* stream, const void* x_dev, const void* gamma_dev, const void* beta_dev,
                                void* out_dev, int64_t out_batch_stride, void* workspace_dev, int64_t B, int64_t HW,
                                int64_t C, int groups, float eps) {
  return gn_entry<true>(stream, x_dev, gamma_dev, beta_dev, out_dev, out_batch_stride, workspace_dev, B, HW, C, groups,
                       eps);
}

}  // extern "C"
