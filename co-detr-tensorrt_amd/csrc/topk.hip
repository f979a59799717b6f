// Row-wise top-k for the two selections of the detection head: the two-stage proposal choice (top 900 of the S = 204 600
// per-token best class logits, reference codetr/transformer.py:560-561) and the final detections (top 300 of the
// 900 x 80 sigmoid scores, reference codetr/co_dino_head.py:183-186).  Replaces torch.topk (a radix sort + gather chain
// of rocPRIM kernels) with one launch: one 1024-thread workgroup per row.
//
// Order: descending value, ties by ascending index, NaN first (torch.topk's NaN rule; its tie order is unspecified).
// Every element gets a unique 40-bit composite  C = key16 << 24 | (2^24 - 1 - index), key16 = the order-preserving
// integer image of the 16-bit float (NaN -> 0xFFFF): the k largest composites ARE the answer, there is no tie to
// break.
//   * rows of up to 204 800 elements (the model's sizes): every thread owns a contiguous share of the row and sweeps
//     it four times with 16-byte loads (the row stays in L2): two 256-bin histogram rounds give the 16-bit threshold
//     value (run-length flushing: scores cluster in a few bins), a block scan over (greater, tied) counts places the
//     winners -- ties in index order -- without atomics, a bitonic sort of the <= 1024 composites orders them.
//   * longer rows: MSD radix select over the 5 bytes of C with one sweep of the row per byte (256-bin LDS histograms,
//     wave-aggregated atomics), compaction of the k winners, the same sort.
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

// block-wide exclusive scan of one value per thread (1024 threads = 16 waves): shuffle scan inside each wave, the 16
// wave totals through `scratch` (>= 16 u32), two barriers
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* scratch, unsigned* total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  unsigned before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) {
    const unsigned t = scratch[w];
    before += w < wave ? t : 0u;
    all += t;
  }
  if (total) *total = all;
  __syncthreads();
  return before + incl - v;
}

// bitonic sort of buf[0 .. 1023], descending, 1024 threads.  Pairs at distance <= 64 stay inside the 128-element
// segment of the thread's own wave (LDS operations of a wave retire in order), so only the strides >= 128 need a
// workgroup barrier: 6 barriers instead of 55.
__device__ __forceinline__ void bitonic_sort_desc(unsigned long long* buf) {
  const int tid = threadIdx.x;
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
      if (stride > 64 || (stride == 1 && size >= 128)) {
        __syncthreads();  // the next step crosses wave segments (or this was the last wave-local step before one)
      } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  __syncthreads();
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
  bitonic_sort_desc(buf);
  if (tid < k) {
    const unsigned long long c = buf[tid];
    const unsigned idx = 0xffffffu - (unsigned)(c & 0xffffffu);
    indices[(size_t)blockIdx.x * k + tid] = (int64_t)idx;
    if (values) values[(size_t)blockIdx.x * k + tid] = row[idx];
  }
}

constexpr int kRegVecs = 25;  // 16-byte vectors per thread held in registers: rows up to 1024 * 25 * 8 elements

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short elem16(const u32x4& v, int e) {
  const unsigned w = v[e >> 1];
  return (unsigned short)((e & 1) ? (w >> 16) : (w & 0xffffu));
}

// Rows of up to kRegVecs * 8 * 1024 elements: thread t owns the contiguous share [t * 8 nvec, (t + 1) * 8 nvec) and
// sweeps it four times with 16-byte loads, five in flight (the row stays in L2): two histogram rounds, the count
// round, the placement round.
template <bool BF>
__global__ __launch_bounds__(kThreads) void topk_reg_kernel(const unsigned short* __restrict__ x,
                                                            unsigned short* __restrict__ values,
                                                            int64_t* __restrict__ indices, int n, int k, int nvec,
                                                            int chunks, const int64_t* __restrict__ index_map) {
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_s[2];
  __shared__ unsigned long long buf[kMaxK];
  unsigned* scratch = reinterpret_cast<unsigned*>(buf);  // the scans finish before buf is filled
  const int tid = threadIdx.x;
  const unsigned short* row = x + (size_t)blockIdx.x * n;
  const int base = tid * nvec * 8;

  constexpr int kBatch = 5;
  // visit(f): f(key, index) for every element of this thread's share, in index order; keys of positions past the row
  // are 0 (below every real key, which is >= 0x0400)
  auto sweep = [&](auto&& f) {
    for (int j0 = 0; j0 < nvec; j0 += kBatch) {
      u32x4 v[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        v[u] = u32x4{0u, 0u, 0u, 0u};
        const int i0 = base + (j0 + u) * 8;
        if (j0 + u < nvec && i0 < n) {
          if (i0 + 8 <= n) {
            v[u] = *reinterpret_cast<const u32x4*>(row + i0);
          } else {
            for (int e = 0; e < 8; ++e)
              if (i0 + e < n) v[u][e >> 1] |= (unsigned)row[i0 + e] << (16 * (e & 1));
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        if (j0 + u < nvec) {
          const int i0 = base + (j0 + u) * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) f(i0 + e < n ? key16<BF>(elem16(v[u], e)) : 0u, (unsigned)(i0 + e));
        }
      }
    }
  };

  // ---- threshold value: two 8-bit histogram rounds over the 16-bit keys ----
  unsigned prefix = 0, need = (unsigned)k;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    unsigned cur = 0, cnt = 0;  // run-length: one LDS atomic per run of equal bins (scores cluster in a few bins)
    sweep([&](unsigned key, unsigned) {
      const bool in = key != 0u && (pass == 0 || (key >> 8) == prefix);
      const unsigned bin = pass == 0 ? key >> 8 : key & 0xffu;
      if (in) {
        if (cnt && bin == cur) {
          ++cnt;
        } else {
          if (cnt) atomicAdd(&hist[cur], cnt);
          cur = bin;
          cnt = 1;
        }
      }
    });
    if (cnt) atomicAdd(&hist[cur], cnt);
    __syncthreads();
    if (tid < 64) {
      // threshold bin, searched by one wave: lane l owns bins 4l .. 4l+3, a suffix sum over the lanes finds the lane in
      // which the count from the top crosses `need`, that lane walks its four bins
      const unsigned h[4] = {hist[4 * tid], hist[4 * tid + 1], hist[4 * tid + 2], hist[4 * tid + 3]};
      const unsigned own = h[0] + h[1] + h[2] + h[3];
      unsigned suf = own;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = __shfl_down(suf, off, 64);
        if (tid + off < 64) suf += up;
      }
      const unsigned above = suf - own;  // elements in the bins of higher lanes
      if (above < need && need <= above + own) {
        unsigned acc = above;
        int j = 3;
        for (; j > 0; --j) {
          if (acc + h[j] >= need) break;
          acc += h[j];
        }
        sel_s[0] = (prefix << 8) | (unsigned)(4 * tid + j);
        sel_s[1] = need - acc;
      }
    }
    __syncthreads();
    prefix = sel_s[0];
    need = sel_s[1];
    __syncthreads();
  }
  const unsigned T = prefix;          // 16-bit key of the k-th largest element
  const unsigned ties_wanted = need;  // how many elements equal to T belong to the result (lowest indices first)
  const unsigned n_greater = (unsigned)k - ties_wanted;

  // ---- placement without atomics: ascending index = (thread, position inside the thread's share) ----
  unsigned my_gt = 0, my_tie = 0;
  sweep([&](unsigned key, unsigned) {
    my_gt += key > T ? 1u : 0u;
    my_tie += key == T ? 1u : 0u;
  });
  unsigned gt_slot = block_exclusive_scan(my_gt, scratch, nullptr);
  unsigned tie_seen = block_exclusive_scan(my_tie, scratch, nullptr);
  for (int i = tid; i < kMaxK; i += kThreads) buf[i] = 0;  // pad
  __syncthreads();
  sweep([&](unsigned key, unsigned idx) {
    const unsigned long long c = ((unsigned long long)key << 24) | (unsigned long long)(0xffffffu - idx);
    if (key > T) {
      buf[gt_slot++] = c;
    } else if (key == T) {
      if (tie_seen < ties_wanted) buf[n_greater + tie_seen] = c;
      ++tie_seen;
    }
  });
  __syncthreads();
  bitonic_sort_desc(buf);
  if (tid < k) {
    const unsigned long long c = buf[tid];
    const unsigned idx = 0xffffffu - (unsigned)(c & 0xffffffu);
    // chunks > 1: this "row" is chunk (blockIdx.x % chunks) of a longer row -> report the index inside that row;
    // index_map: the elements are candidates of an earlier selection -> report the index each one came from
    int64_t o = (int64_t)idx + (int64_t)(blockIdx.x % (unsigned)chunks) * n;
    if (index_map) o = index_map[(size_t)blockIdx.x * n + idx];
    indices[(size_t)blockIdx.x * k + tid] = o;
    if (values) values[(size_t)blockIdx.x * k + tid] = row[idx];
  }
}

template <bool BF>
int topk_entry(void* stream, const void* x, int64_t rows, int64_t n, int k, void* values, int64_t* indices,
               int chunks = 1, const int64_t* index_map = nullptr) {
  if (!x || !indices || rows <= 0 || n <= 0 || k <= 0 || chunks <= 0) return CODETR_E_BADARG;
  if (k > kMaxK || k > n || n >= (1 << 24)) return CODETR_E_UNSUPPORTED;
  if (rows > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  const int nvec = (int)((n + kThreads * 8 - 1) / (kThreads * 8));  // 16-byte vectors per thread
  // (a single row may have any length: its start is aligned and the sweep reads the last partial vector element by element;
  // 608x608 has 30 785 encoder tokens and took the one-sweep-per-round kernel below at 106 us instead of 25)
  if (nvec <= kRegVecs && (n % 8 == 0 || rows == 1) && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    hipLaunchKernelGGL(topk_reg_kernel<BF>, dim3((unsigned)rows), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned short*>(x), static_cast<unsigned short*>(values), indices, (int)n, k,
                       nvec, chunks, index_map);
  } else {
    if (chunks != 1 || index_map) return CODETR_E_UNSUPPORTED;
    hipLaunchKernelGGL(topk_kernel<BF>, dim3((unsigned)rows), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned short*>(x), static_cast<unsigned short*>(values), indices, (int)n, k);
  }
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

// Long rows over several workgroups: pass 1 selects the top k of each of `chunks` equal pieces of every row (rows *
// chunks workgroups), pass 2 selects the top k of the chunks * k candidates and maps them back.  Candidates of one row
// are laid out chunk-major and sorted inside a chunk, so position order among equal values is index order: the result
// is the same stable order as the one-pass kernel's.  Workspace: rows * chunks * k * (2 + 8) bytes, caller-owned.
template <bool BF>
int topk_two_pass(void* stream, const void* x, int64_t rows, int64_t n, int k, int chunks, void* values,
                  int64_t* indices, void* workspace, int64_t workspace_bytes) {
  if (!workspace || chunks < 2 || n % chunks != 0) return CODETR_E_BADARG;
  const int k1 = (k + 7) & ~7;  // candidates kept per piece: k rounded up so that the candidate rows stay 16-byte multiples
  const int64_t cn = n / chunks, cand = (int64_t)chunks * k1;
  if (cn % 8 != 0 || cn < k1 || k1 > kMaxK || workspace_bytes < rows * cand * 10 ||
      (reinterpret_cast<uintptr_t>(workspace) & 15))
    return CODETR_E_BADARG;
  int64_t* cand_idx = static_cast<int64_t*>(workspace);
  unsigned short* cand_val = reinterpret_cast<unsigned short*>(cand_idx + rows * cand);
  int rc = topk_entry<BF>(stream, x, rows * chunks, cn, k1, cand_val, cand_idx, chunks, nullptr);
  if (rc) return rc;
  return topk_entry<BF>(stream, cand_val, rows, cand, k, values, indices, 1, cand_idx);
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

int64_t codetr_topk_chunks(int64_t n, int k, int64_t rows, int64_t* workspace_bytes) {
  // how many pieces codetr_topk_chunked_* should cut a row into (1 = call codetr_topk_*): the largest divisor <= 32 of
  // n whose pieces are 16-byte multiples of at least k elements; only worth it for rows beyond one sweep's comfort
  if (workspace_bytes) *workspace_bytes = 0;
  const int k1 = (k + 7) & ~7;
  if (n < 32768 || k <= 0 || k1 > kMaxK) return 1;
  for (int c = 32; c >= 2; --c) {
    if (n % c == 0 && (n / c) % 8 == 0 && n / c >= k1) {
      if (workspace_bytes) *workspace_bytes = rows * c * k1 * 10;
      return c;
    }
  }
  return 1;
}

int codetr_topk_chunked_f16(void* stream, const void* x_dev, int64_t rows, int64_t n, int k, int chunks,
                            void* values_dev, int64_t* indices_dev, void* workspace_dev, int64_t workspace_bytes) {
  return topk_two_pass<false>(stream, x_dev, rows, n, k, chunks, values_dev, indices_dev, workspace_dev,
                              workspace_bytes);
}

int codetr_topk_chunked_bf16(void* stream, const void* x_dev, int64_t rows, int64_t n, int k, int chunks,
                             void* values_dev, int64_t* indices_dev, void* workspace_dev, int64_t workspace_bytes) {
  return topk_two_pass<true>(stream, x_dev, rows, n, k, chunks, values_dev, indices_dev, workspace_dev,
                             workspace_bytes);
}

}  // extern "C"
