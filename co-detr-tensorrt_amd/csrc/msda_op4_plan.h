// Device-side plan of the windowed form of the PUBLIC multi-scale deformable attention op (csrc/msda_op4.hip), shared with
// the general kernel of csrc/msda_forward.hip.
//
// The op's pyramid shapes arrive as a DEVICE tensor (reference codetr/csrc/deformable_attention_torch.cpp:16-31: value_spatial_
// shapes, value_level_start_index), so the host cannot size a (region, head) grid or choose windows without a device-to-host
// copy.  Instead BOTH kernels are launched, and every workgroup of both evaluates this plan from the device-side shapes with
// scalar arithmetic: where it applies the windowed kernel does the work and the general kernel's workgroups return at once,
// where it does not it is the other way round.  Nothing here depends on sampling locations: it is a function of
// (shapes, level starts, S) alone, so both kernels always agree.
#pragma once
#include <stdint.h>

namespace codetr_op4 {

constexpr int kL = 5, kP = 4;
constexpr int kRegion = 16;          // a workgroup's region: 16 x 16 pixels of the finest level
constexpr int kThreads = 512;
constexpr int kPairs = kThreads / 4; // (query, head) pairs per workgroup iteration
constexpr int kMaxIt = 3;            // iterations per wave
constexpr int kWinPixels = 1200;     // staged pixels (64 B each) a pass may use: 75 KiB of the 80 KiB a workgroup may hold (4 KiB: the fix-up queue)
constexpr int kMarginCap = 12;       // pixels

struct Plan {
  int W[kL], H[kL], start[kL];
  int RX, RY;                        // regions along x / y
  int mg[kL];                        // window margin of each level, pixels of that level
};

// floor(a / b) for 0 <= a < 2^22, 0 < b.  Device: one float reciprocal + an exact fix-up (a 32-bit integer division is ~40
// instructions, and the general kernel evaluates the verdict below in every workgroup of its skipped launch).
__host__ __device__ inline int idiv(int a, int b) {
#ifdef __HIP_DEVICE_COMPILE__
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
  const int r = a - q * b;
  q += r >= b ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
#else
  return a / b;
#endif
}
__host__ __device__ inline int cdiv(int a, int b) { return idiv(a + b - 1, b); }

// first level of the pass a level belongs to: passes {0}, {1, 2}, {3, 4}
__host__ __device__ inline int pass_first(int l) { return l == 0 ? 0 : l <= 2 ? 1 : 3; }

// Pixels a pass (levels first .. last) stages at margin mg, an upper bound over the regions.  Staged columns of a region on
// level l: floor(r n / R - 1/2 - mg) .. ceil((r + 1) n / R - 1/2 + mg), clamped to the level plus its zero border
// -> at most min(ceil(n / R) + 2 mg + 3, n + 2).
__host__ __device__ inline int pass_pixels(const Plan& p, const int (&cw)[kL], const int (&ch)[kL], int first, int last, int mg) {
  int px = 0;
  for (int l = first; l <= last; ++l) {
    int w = cw[l] + 2 * mg + 3, h = ch[l] + 2 * mg + 3;
    w = w < p.W[l] + 2 ? w : p.W[l] + 2;
    h = h < p.H[l] + 2 ? h : p.H[l] + 2;
    px += w * h;
  }
  return px;
}

// Returns true where the windowed kernel serves the call, and fills the plan (margins only with `margins`: the general
// kernel's skip test needs the verdict alone).  Conditions (speed assumptions of the windowed kernel -- any call they exclude
// is served by the general kernel, with identical results):
//   the level starts are the prefix sums of the level sizes and the sizes sum to S (a dense pyramid);
//   level 0 is the largest level; every side <= 4096; a region's queries fit kPairs * kMaxIt slots;
//   B * M * regions < 2^22 (tile numbers go through a reciprocal-based division);
//   every pass fits kWinPixels with a margin of at least 0.
__host__ __device__ inline bool make_plan(const int64_t* __restrict__ ss, const int64_t* __restrict__ ls, int64_t S, int64_t BM,
                                          Plan& p, bool margins = true) {
  bool ok = true;
  int64_t sum = 0;
  for (int l = 0; l < kL; ++l) {
    const int64_t h = ss[2 * l], w = ss[2 * l + 1];
    ok = ok && h > 0 && w > 0 && h <= 4096 && w <= 4096 && ls[l] == sum;
    p.H[l] = (int)h;
    p.W[l] = (int)w;
    p.start[l] = (int)sum;
    sum += h * w;
  }
  ok = ok && sum == S;
  if (!ok) return false;
  for (int l = 1; l < kL; ++l) ok = ok && p.H[l] * p.W[l] <= p.H[0] * p.W[0];
  p.RX = cdiv(p.W[0], kRegion);
  p.RY = cdiv(p.H[0], kRegion);
  int cw[kL], ch[kL], slots = 0;
  for (int l = 0; l < kL; ++l) {
    cw[l] = cdiv(p.W[l], p.RX);                     // (a region's share of a level is at most the ceiling)
    ch[l] = cdiv(p.H[l], p.RY);
    slots += cw[l] * ch[l];
  }
  ok = ok && slots <= kPairs * kMaxIt && BM * p.RX * p.RY < ((int64_t)1 << 22);
  ok = ok && pass_pixels(p, cw, ch, 0, 0, 0) <= kWinPixels && pass_pixels(p, cw, ch, 1, 2, 0) <= kWinPixels &&
       pass_pixels(p, cw, ch, 3, 4, 0) <= kWinPixels;
  if (!ok || !margins) return ok;
  // per pass: the largest uniform margin (<= kMarginCap) whose windows fit (0 does, see above)
  for (int first = 0; first < kL; first = first == 0 ? 1 : first + 2) {
    const int last = first == 0 ? 0 : first + 1;
    int mg = kMarginCap;
    while (mg > 0 && pass_pixels(p, cw, ch, first, last, mg) > kWinPixels) --mg;
    for (int l = first; l <= last; ++l) p.mg[l] = mg;
  }
  return true;
}

}  // namespace codetr_op4
