// Pre- and post-processing either side of CoDETR.forward, on the GPU.
//
// preprocess_kernel   replaces the reference Inferencer's CPU pipeline per image (codetr/inferencer.py:439-452 ->
//                     mmdet Resize(keep_ratio) = cv2.resize(INTER_LINEAR) on uint8, mmdet Pad(size, pad_val),
//                     DetDataPreprocessor (x - mean) / std in fp32, cast to the model dtype) and the mask loop of
//                     run_inference (codetr/inferencer.py:354-358): uint8 HWC RGB in, normalised CHW + mask out.
//                     The resize is OpenCV's 8-bit arithmetic (11-bit fixed-point coefficients, cvRound, edge clamp,
//                     (.. + 2^21) >> 22), so the result is the integer image cv2 would produce, then one fp32
//                     subtract and one IEEE divide per value.  Byte work: one thread per output pixel.
// batched_nms_kernel  replaces torchvision.ops.batched_nms in postprocess_predictions (codetr/inferencer.py:388-398)
//                     for the <= 300 detections of an image: greedy, per class, candidates in descending score order
//                     (the caller passes them sorted), IoU > thr suppresses; fp32 arithmetic on the given boxes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

struct ResizeAxis {
  float scale;  // src / dst
  int src, dst;
};

// source index and the weight of the NEXT sample, scaled by 2048 (cv2: cvRound(f * INTER_RESIZE_COEF_SCALE))
__device__ __forceinline__ void coeff(int d, ResizeAxis ax, int& s, int& a1) {
  // cv2: scale = 1. / (dsize / ssize) in double; fx = (float)((dx + 0.5) * scale - 0.5); sx = cvFloor(fx); fx -= sx
  const double scale = 1.0 / ((double)ax.dst / (double)ax.src);
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int i = (int)floorf(f);
  f -= (float)i;
  if (i < 0) {
    i = 0;
    f = 0.f;
  }
  if (i >= ax.src - 1) {
    i = ax.src - 1;
    f = 0.f;
  }
  s = i;
  a1 = (int)rintf(f * 2048.0f);  // round half to even, as cvRound
}

template <class OutT>
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* __restrict__ src, int Hs, int Ws, int Hr,
                                                         int Wr, int Hp, int Wp, float m0, float m1, float m2, float s0,
                                                         float s1, float s2, int p0, int p1, int p2,
                                                         OutT* __restrict__ dst, OutT* __restrict__ mask) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= Wp) return;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
  int v[3] = {p0, p1, p2};
  const bool inside = y < Hr && x < Wr;
  if (inside) {
    int sy, b1, sx, a1;
    coeff(y, ResizeAxis{0.f, Hs, Hr}, sy, b1);
    coeff(x, ResizeAxis{0.f, Ws, Wr}, sx, a1);
    const int sy1 = min(sy + 1, Hs - 1), sx1 = min(sx + 1, Ws - 1);
    const int a0 = 2048 - a1, b0 = 2048 - b1;
    const unsigned char* r0 = src + ((size_t)sy * Ws) * 3;
    const unsigned char* r1 = src + ((size_t)sy1 * Ws) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int top = a0 * r0[sx * 3 + c] + a1 * r0[sx1 * 3 + c];
      const int bot = a0 * r1[sx * 3 + c] + a1 * r1[sx1 * 3 + c];
      const long long acc = (long long)b0 * top + (long long)b1 * bot + (1 << 21);
      int q = (int)(acc >> 22);
      v[c] = q < 0 ? 0 : (q > 255 ? 255 : q);
    }
  }
  const size_t plane = (size_t)Hp * Wp, o = (size_t)y * Wp + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) dst[c * plane + o] = (OutT)(((float)v[c] - mean[c]) / stdv[c]);
  if (mask) mask[o] = inside ? (OutT)0.f : (OutT)1.f;
}

// one workgroup; boxes / labels in descending score order.  keep[i] = 1 if i survives.
__global__ __launch_bounds__(1024) void batched_nms_kernel(const float* __restrict__ boxes,
                                                           const int64_t* __restrict__ labels, int N, float thr,
                                                           unsigned char* __restrict__ keep) {
  extern __shared__ unsigned char s_keep[];
  for (int j = threadIdx.x; j < N; j += blockDim.x) s_keep[j] = 1;
  __syncthreads();
  for (int i = 0; i < N; ++i) {
    if (s_keep[i]) {  // uniform across the workgroup (read after the barrier below)
      const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
      const float ia = (ix2 - ix1) * (iy2 - iy1);
      const int64_t il = labels[i];
      for (int j = i + 1 + threadIdx.x; j < N; j += blockDim.x) {
        if (!s_keep[j] || labels[j] != il) continue;
        const float jx1 = boxes[4 * j], jy1 = boxes[4 * j + 1], jx2 = boxes[4 * j + 2], jy2 = boxes[4 * j + 3];
        const float w = fmaxf(0.f, fminf(ix2, jx2) - fmaxf(ix1, jx1));
        const float h = fmaxf(0.f, fminf(iy2, jy2) - fmaxf(iy1, jy1));
        const float inter = w * h;
        const float iou = inter / (ia + (jx2 - jx1) * (jy2 - jy1) - inter);
        if (iou > thr) s_keep[j] = 0;
      }
    }
    __syncthreads();
  }
  for (int j = threadIdx.x; j < N; j += blockDim.x) keep[j] = s_keep[j];
}

template <class OutT>
int launch_pre(void* stream, const void* src, int64_t Hs, int64_t Ws, int64_t Hr, int64_t Wr, int64_t Hp, int64_t Wp,
               const float* mean, const float* stdv, const int* pad, void* dst, void* mask) {
  if (!src || !dst || !mean || !stdv || !pad || Hs <= 0 || Ws <= 0 || Hr <= 0 || Wr <= 0 || Hp < Hr || Wp < Wr)
    return CODETR_E_BADARG;
  if (Hs > 32767 || Ws > 32767 || Hp > 65535 || Wp > 0x7fffffffLL) return CODETR_E_TOO_LARGE;  // 2048 * 255 * 2048 fits int64; indices int
  for (int c = 0; c < 3; ++c)
    if (stdv[c] == 0.f || pad[c] < 0 || pad[c] > 255) return CODETR_E_BADARG;
  hipLaunchKernelGGL((preprocess_kernel<OutT>), dim3((unsigned)((Wp + 255) / 256), (unsigned)Hp), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned char*>(src), (int)Hs, (int)Ws, (int)Hr,
                     (int)Wr, (int)Hp, (int)Wp, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], pad[0], pad[1],
                     pad[2], static_cast<OutT*>(dst), static_cast<OutT*>(mask));
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int codetr_preprocess_u8_f16(void* stream, const void* src_dev, int64_t H_src, int64_t W_src, int64_t H_resized,
                             int64_t W_resized, int64_t H_pad, int64_t W_pad, const float* mean_host,
                             const float* std_host, const int* pad_value_host, void* dst_dev, void* mask_dev) {
  return launch_pre<_Float16>(stream, src_dev, H_src, W_src, H_resized, W_resized, H_pad, W_pad, mean_host, std_host,
                              pad_value_host, dst_dev, mask_dev);
}

int codetr_preprocess_u8_f32(void* stream, const void* src_dev, int64_t H_src, int64_t W_src, int64_t H_resized,
                             int64_t W_resized, int64_t H_pad, int64_t W_pad, const float* mean_host,
                             const float* std_host, const int* pad_value_host, void* dst_dev, void* mask_dev) {
  return launch_pre<float>(stream, src_dev, H_src, W_src, H_resized, W_resized, H_pad, W_pad, mean_host, std_host,
                           pad_value_host, dst_dev, mask_dev);
}

int codetr_batched_nms_f32(void* stream, const float* boxes_sorted_dev, const int64_t* labels_sorted_dev, int64_t N,
                           float iou_threshold, void* keep_dev) {
  if (N == 0) return 0;
  if (!boxes_sorted_dev || !labels_sorted_dev || !keep_dev || N < 0) return CODETR_E_BADARG;
  if (N > 60000) return CODETR_E_TOO_LARGE;  // keep flags live in LDS
  const int threads = N >= 1024 ? 1024 : (int)((N + 63) / 64 * 64);
  hipLaunchKernelGGL(batched_nms_kernel, dim3(1), dim3(threads), (size_t)((N + 15) / 16 * 16),
                     static_cast<hipStream_t>(stream), boxes_sorted_dev, labels_sorted_dev, (int)N, iou_threshold,
                     static_cast<unsigned char*>(keep_dev));
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // extern "C"
