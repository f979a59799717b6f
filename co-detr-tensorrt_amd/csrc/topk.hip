// Row-wise top-k for the two selections of the detection head: the two-stage proposal choice (top 900 of the S = 204 600
// per-token best class logits, reference codetr/transformer.py:560-561) and the final detections (top 300 of the
// 900 x 80 sigmoid scores, reference codetr/co_dino_head.py:183-186).  Replaces torch.topk (a radix sort + gather chain
// of rocPRIM kernels) with one launch: one 1024-thread workgroup per row.
//
// Order: descending value, ties by ascending index, NaN first (torch.topk's NaN rule; its tie order is unspecified).
// Every element gets a unique 40-bit composite  C = key16 << 24 | (2^24 - 1 - index), key16 = the order-preserving
// integer image of the 16-bit float (NaN -> 0xFFFF): the k largest composites ARE the answer, there is no tie to
// break.  MSD radix select over the 5 bytes of C (256-bin LDS histograms, wave-aggregated atomics because scores
// cluster in a few bins), compaction of the k winners, bitonic sort of <= 1024 composites in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kMaxK = 1024;

template <bool BF>
__device__ __forceinline__ unsigned key16(unsigned short bits) {
  const unsigned mag = bits & 0x7fffu;
  if (mag > (BF ? 0x7f80u : 0x7c00u)) return 0xffffu;  // NaN sorts above everything
  return (bits & 0x8000u) ? (~(unsigned)bits & 0xffffu) : ((unsigned)bits | 0x8000u);
}

template <bool BF>
__device__ __forceinline__ unsigned long long composite(unsigned short bits, unsigned idx) {
  return ((unsigned long long)key16<BF>(bits) << 24) | (unsigned long long)(0xffffffu - idx);
}

// hist[bin] += 1 for every active lane, one LDS atomic per distinct bin of the wave
__device__ __forceinline__ void wave_hist_add(unsigned* hist, unsigned bin, bool active) {
  unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
  const unsigned lane = threadIdx.x & 63u;
  while (todo) {
    const int leader = __builtin_ctzll(todo);
    const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
    const unsigned long long same = __builtin_amdgcn_ballot_w64(active && bin == b);
    if (lane == (unsigned)leader) atomicAdd(&hist[b], (unsigned)__builtin_popcountll(same));
    todo &= ~same;
  }
}

template <bool BF>
__global__ __launch_bounds__(kThreads) void topk_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ values,
                                                        int64_t* __restrict__ indices, int n, int k) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long prefix_s;
  __shared__ unsigned need_s, count_s;
  __shared__ unsigned long long buf[kMaxK];
  const int tid = threadIdx.x;
  const unsigned short* row = x + (size_t)blockIdx.x * n;

  // ---- MSD radix select: after byte `pass` the top (pass + 1) bytes of the k-th largest composite are known ----
  unsigned long long prefix = 0;  // known high bytes of the threshold, right-aligned
  unsigned need = (unsigned)k;    // how many of the elements matching the prefix are still wanted
  for (int pass = 0; pass < 5; ++pass) {
    const int shift = 32 - 8 * pass;  // this byte = bits [shift, shift + 8)
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += kThreads) {
      const int i = i0 + tid;
      bool act = i < n;
      unsigned bin = 0;
      if (act) {
        const unsigned long long c = composite<BF>(row[i], (unsigned)i);
        act = pass == 0 || (c >> (shift + 8)) == prefix;
        bin = (unsigned)(c >> shift) & 0xffu;
      }
      wave_hist_add(hist, bin, act);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned acc = 0;
      int b = 255;
      for (; b > 0; --b) {
        if (acc + hist[b] >= need) break;
        acc += hist[b];
      }
      prefix_s = (prefix << 8) | (unsigned)b;
      need_s = need - acc;  // elements wanted inside bin b
    }
    __syncthreads();
    prefix = prefix_s;
    need = need_s;
    __syncthreads();
  }
  // prefix is now the complete 40-bit composite of the k-th largest element: winners are exactly { C >= prefix }
  if (tid == 0) count_s = 0;
  __syncthreads();
  for (int i = tid; i < n; i += kThreads) {
    const unsigned long long c = composite<BF>(row[i], (unsigned)i);
    if (c >= prefix) {
      const unsigned slot = atomicAdd(&count_s, 1u);
      if (slot < (unsigned)kMaxK) buf[slot] = c;
    }
  }
  __syncthreads();
  for (int i = (int)count_s + tid; i < kMaxK; i += kThreads) buf[i] = 0;  // pad (count_s == k)
  __syncthreads();
  // ---- bitonic sort, descending, 1024 composites, one element pair per thread and step ----
  for (int size = 2; size <= kMaxK; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (tid < kMaxK / 2) {
        const int lo = 2 * tid - (tid & (stride - 1));  // index of the lower element of this thread's pair
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long a = buf[lo], b = buf[hi];
        if ((a < b) == desc) {
          buf[lo] = b;
          buf[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  if (tid < k) {
    const unsigned long long c = buf[tid];
    const unsigned idx = 0xffffffu - (unsigned)(c & 0xffffffu);
    indices[(size_t)blockIdx.x * k + tid] = (int64_t)idx;
    if (values) values[(size_t)blockIdx.x * k + tid] = row[idx];
  }
}

template <bool BF>
int topk_entry(void* stream, const void* x, int64_t rows, int64_t n, int k, void* values, int64_t* indices) {
  if (!x || !indices || rows <= 0 || n <= 0 || k <= 0) return CODETR_E_BADARG;
  if (k > kMaxK || k > n || n >= (1 << 24)) return CODETR_E_UNSUPPORTED;
  if (rows > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  hipLaunchKernelGGL(topk_kernel<BF>, dim3((unsigned)rows), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(x), static_cast<unsigned short*>(values), indices, (int)n, k);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int codetr_topk_f16(void* stream, const void* x_dev, int64_t rows, int64_t n, int k, void* values_dev,
                    int64_t* indices_dev) {
  return topk_entry<false>(stream, x_dev, rows, n, k, values_dev, indices_dev);
}

int codetr_topk_bf16(void* stream, const void* x_dev, int64_t rows, int64_t n, int k, void* values_dev,
                     int64_t* indices_dev) {
  return topk_entry<true>(stream, x_dev, rows, n, k, values_dev, indices_dev);
}

}  // extern "C"
