// Persistent GEMM for the large linears of the hot path (Swin stages 0-3: reference codetr/swin.py:92-112 qkv / proj,
// :331-352 the MLP):  Y[M,N] = act(X[M,K] . W[N,K]^T + bias[N]) (+ R[M,N]),  fp16 / bf16 storage, fp32 accumulation on
// v_mfma_f32_16x16x32_{f16,bf16}.  Built from what profiles/r03_gemm256_ablation.txt measured on the 256-tile kernel of
// gemm_f16.hip (a burst of fragment reads behind every barrier, the first loads of every tile exposed, 12 % tile-count
// quantisation); what each step was worth, and the forms that lost, is in profiles/r04_gemm_sk.txt.
//
//   * one 512-thread workgroup per CU, resident for the whole launch; 8 waves (two per SIMD) as 2 (m) x 4 (n), each wave
//     a 128 x 64 piece of a 256 x 256 output tile = 8 x 4 MFMA tiles, 128 accumulator registers.  (A 4-wave / 128 x 128 /
//     AGPR form of this pipeline was built first and measured 2 x slower: with one wave per SIMD nothing covers the
//     100+ cycles an LDS-DMA instruction takes to issue.)
//   * operands move global -> LDS by LDS-DMA (16 B per lane, scalar base + per-thread offset) in HALF-STAGES of 32 k:
//     W[256 rows][64 B] + X[256 rows][64 B] = 32 KiB, a ring of NS = 4 of them (5 measured the same).  In phase p a
//     wave multiplies the fragments of half-stage p (already in registers), reads those of p + 1 into the other register
//     set and issues its 4 DMA pieces of half-stage p + NS into the slot p just left; ONE counted wait + barrier per
//     phase (half-stage p + 2 has landed for everybody; everybody is done reading slot p + 1).  No MFMA ever waits for
//     an LDS read of its own phase.
//   * the half-stage stream does not stop at tile boundaries: the producer side runs NS half-stages ahead through the
//     workgroup's whole list of tiles, so the first loads of a tile fly under the previous tile's last phases and
//     epilogue (one workgroup per tile -- flag 0x20 -- measured 7-20 % slower).
//   * work list: floor(T / G) rounds of whole tiles on the G resident workgroups (XCD-aware order: the 32 workgroups of
//     an XCD walk 32 consecutive tiles, n fastest) and the T mod G left-over tiles as whole tiles of the first workgroups.
//     Stream-K (flag 0x40) cuts the left-over tiles along K into equal ranges of 64-deep k-tiles over up to G workgroups
//     (at most 4 parts per tile): a part stores its fp32 accumulators to a per-(workgroup, wave) slab and takes a ticket
//     on the tile's per-wave counter; the wave that draws the last ticket adds the other parts and runs the epilogue of
//     its piece -- no workgroup ever waits for another one (placement- and dispatch-order independent; write-through
//     slab stores + s_waitcnt vmcnt(0) + relaxed agent atomic, the last arriver takes an agent-scope acquire;
//     MI355X_MICROARCH.md, inter-workgroup visibility).  It is correct (tools/micro/gemm_sk_bench, tests) but NOT the
//     default: at this model's K <= 3072 a 256 x 256 fp32 partial (256 KiB out, 256 KiB back in) costs as much as the
//     tile's own operand traffic, and the split measured 10-50 % slower than the quantisation it removes.
//   * MFMA roles: A = X rows, B = W rows, so a lane's accumulator column is an output column; the weight rows are staged
//     in a permuted order (on the DMA source side: tile i, column c <-> column 4 c + i of the wave's 64) so that the four
//     n-tiles give lane c the 4 consecutive output columns 4 c .. 4 c + 3 of a row: 16 adjacent lanes store one whole
//     128-byte line, 4 rows per store instruction, straight from registers -- no LDS staging (the ring keeps running).
//     Stores and residual loads are buffer instructions on a wave-uniform descriptor with ONE 32-bit lane offset and the
//     row in the scalar offset: per-lane 64-bit addresses were hoisted and spilled by the compiler, and a scratch access
//     is a vector-memory operation that waits for every LDS-DMA piece in flight (vmcnt retires in order) -- that alone
//     was 40 % of the first version's time.  For the same reason the bias sits in LDS behind the ring, and the fragments
//     of the next tile's first half-stage are dropped over the epilogue and re-read behind it (one barrier).
//
// Requirements: K % 64 == 0, K >= 128, N % 8 == 0, N <= 16384, dense row-major operands, 16-byte aligned bases; M, N
// otherwise arbitrary (edge tiles clamp their loads and mask their stores).  No row mask / head-major output
// (gemm_f16.hip serves those).  The workspace (codetr_linear_sk_workspace_bytes) must be zero-filled once; every launch
// leaves its counters at zero.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"
#include "gemm_elem.h"

using namespace codetr_gemm;

namespace {

// diagnostic builds only (tools/micro: -DCODETR_SK_ABL=mask gives WRONG results by construction): 1 = no LDS-DMA inside the
// main loop, 2 = no MFMAs, 4 = no fragment reads, 8 = no wait + barrier per phase, 16 = no output stores
#ifndef CODETR_SK_ABL
#define CODETR_SK_ABL 0
#endif
constexpr int kAbl = CODETR_SK_ABL;
typedef unsigned u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned))));

constexpr int kSlot = 32768;     // one half-stage: W[256][64 B] then X[256][64 B]
constexpr int kOpBytes = 16384;
constexpr int kMaxParts = 4;     // stream-K: most parts a tile is cut into
constexpr int kWaves = 8;
constexpr int kBiasBytes = 32768;     // bias in LDS: N <= 16384
constexpr int kSlabFloats = 8192;    // one wave's 128 x 64 fp32 accumulators

struct SkArgs {
  const unsigned char* X;
  const unsigned char* W;
  const unsigned short* bias;
  const unsigned short* R;
  unsigned short* Y;
  float* slabs;        // [G][2][8 waves][8192] fp32: slab 0 = a part that starts inside its tile, slab 1 = one that starts it
  unsigned* counters;  // [stream-K tiles][8 waves]
  int M, N, K;
  int tiles_n, T, nk;  // nk = K / 64
  int G;               // workgroups (a multiple of 8)
  int D;               // tiles of the data-parallel rounds (a multiple of G); tiles D .. T-1 are the stream-K region
  int S;               // stream-K units (64-deep k-tiles): (T - D) * nk
  int Gs;              // workgroups that take part in the stream-K region (0 .. Gs-1)
  int np;              // 1: one workgroup per tile (G == T == D), XCD-aware order
};

// swizzle key of a 64-byte LDS row (4 chunks of 16 B): {0, 3, 2, 1}[(row >> 2) & 3] -- conflict-free under
// ds_read_b128's lane groups (tests/test_lds_bank_model.py)
__device__ __forceinline__ int key64(int row) {
  const int q = (row >> 2) & 3;
  return q ^ ((q & 1) << 1);
}

// one LDS-DMA piece: 16 B per lane from (wave-uniform 64-bit base in SGPRs) + (per-thread 32-bit byte offset) to the
// wave-uniform LDS address `lds_addr` + lane * 16.  M0 is written in the statement that reads it.
__device__ __forceinline__ void lds_dma16(const unsigned char* src, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds_addr)
               : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- the work list of one workgroup: stream-K range first, then the data-parallel rounds ----
struct Cursor {
  int mode;  // 0: stream-K region, 1: data-parallel rounds, 2: finished
  int u1;    // stream-K: end of this workgroup's unit range
  int r;     // data-parallel: round
  int tile, kb, ke;  // current item: tile and its range of 64-deep k-tiles
};

__device__ __forceinline__ void dp_item(Cursor& c, const SkArgs& a, int w) {
  if (c.r * a.G < a.D) {
    c.mode = 1;
    if (a.np) {   // every XCD (workgroup id mod 8) walks a contiguous run of tiles
      const int q = a.T >> 3, r = a.T & 7, x = w & 7;
      c.tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (w >> 3);
    } else {
      c.tile = (w & 7) * (a.D >> 3) + c.r * (a.G >> 3) + (w >> 3);
    }
    c.kb = 0;
    c.ke = a.nk;
  } else {
    c.mode = 2;
  }
}

__device__ __forceinline__ int sk_begin(const SkArgs& a, int w) {  // first unit of workgroup w (w <= Gs)
  return (int)((long long)w * a.S / a.Gs);
}

__device__ __forceinline__ void cursor_init(Cursor& c, const SkArgs& a, int w) {
  c.r = 0;
  c.u1 = 0;
  if (w < a.Gs) {
    const int u0 = sk_begin(a, w);
    c.u1 = sk_begin(a, w + 1);
    if (u0 < c.u1) {
      c.mode = 0;
      const int t = u0 / a.nk;
      c.tile = a.D + t;
      c.kb = u0 - t * a.nk;
      const int e = c.u1 - t * a.nk;
      c.ke = e < a.nk ? e : a.nk;
      return;
    }
  }
  dp_item(c, a, w);
}

__device__ __forceinline__ void cursor_next(Cursor& c, const SkArgs& a, int w) {
  if (c.mode == 0) {
    const int t = c.tile - a.D;
    if (t * a.nk + c.ke < c.u1) {
      c.tile++;
      c.kb = 0;
      const int e = c.u1 - (t + 1) * a.nk;
      c.ke = e < a.nk ? e : a.nk;
      return;
    }
    c.r = 0;
    dp_item(c, a, w);
    return;
  }
  if (c.mode == 1) {
    c.r++;
    dp_item(c, a, w);
  }
}

template <class T>
struct Frags {
  typename T::frag a[4];  // W rows (MFMA B operand): 4 n-tiles
  typename T::frag b[8];  // X rows (MFMA A operand): 8 m-tiles
};

template <class T, int ACT, bool HAS_BIAS, bool HAS_RES, int NS, bool SK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void linear_sk_kernel(const SkArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NS * kSlot + (HAS_BIAS ? kBiasBytes : 0)];   // ring, then the bias
  using frag = typename T::frag;
  constexpr int PP = 4;               // DMA pieces per wave and half-stage (2 of W, 2 of X)
  constexpr int VMN = PP * (NS - 2);  // pieces that may stay in flight at the end of a phase

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;   // 2 x 4 waves: rows wm*128, columns wn*64
  const int wg = blockIdx.x;
  const int K = a.K, K2 = K * 2;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);

  Cursor cc;   // consumer side
  cursor_init(cc, a, wg);
  if (cc.mode == 2) return;   // nothing to do (fewer k-tiles in the problem than workgroups)
  Cursor pc = cc;             // producer side: NS half-stages ahead

  // ---- producer: this wave's 2 + 2 pieces of a half-stage ----
  // piece P = wave + 8 q covers LDS rows P*16 .. P*16+15 (64 B each): lane -> row P*16 + (lane >> 2), position lane & 3,
  // which holds source chunk (lane & 3) ^ key64(row).  W rows are permuted: LDS row q*64 + i*16 + c holds weight row
  // q*64 + 4 c + i of the tile (see the epilogue).
  unsigned voffW[2], voffX[2];
  const unsigned char* Wp = a.W;
  const unsigned char* Xp = a.X;
  int pk2 = 0;   // next half-stage (32-deep) of the producer's item
  auto prod_set_tile = [&]() {
    const int tm = pc.tile / a.tiles_n, tn = pc.tile - tm * a.tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int nmax = a.N - 1 - n0, mmax = a.M - 1 - m0;   // edge tiles: the last row again (its outputs are never stored)
    // (recomputed from the lane id at every tile switch: kept in registers across the main loop these constants were
    // spilled, and a scratch reload is a vector-memory operation that queues behind every LDS-DMA piece in flight)
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = (wave + 8 * q) * 16 + (ln >> 2);
      const int rw = (r & 192) + 4 * (r & 15) + ((r >> 4) & 3);
      const unsigned co = (unsigned)((((ln & 3) ^ key64(r)) * 16));
      const int rn = rw < nmax ? rw : nmax, rm = r < mmax ? r : mmax;
      voffW[q] = (unsigned)rn * (unsigned)K2 + co;
      voffX[q] = (unsigned)rm * (unsigned)K2 + co;
    }
    Wp = a.W + ((size_t)n0 * K + (size_t)pc.kb * 64) * 2;
    Xp = a.X + ((size_t)m0 * K + (size_t)pc.kb * 64) * 2;
    pk2 = 2 * pc.kb;
  };
  prod_set_tile();
  // one piece (g = 0, 1: W; 2, 3: X) of the producer's current half-stage into ring slot `slot`
  auto produce_piece = [&](int g, int slot) {
    const unsigned dst = lds0 + (unsigned)slot * kSlot + (unsigned)wave * 1024u;
    if (g < 2) lds_dma16(Wp, voffW[g], dst + (unsigned)g * 8192u);
    else lds_dma16(Xp, voffX[g - 2], dst + kOpBytes + (unsigned)(g - 2) * 8192u);
  };
  // past the end of the list the producer re-fetches its last half-stage (nobody reads it): the counted waits stay uniform
  auto produce_advance = [&]() {
    if (pc.mode == 2) return;
    ++pk2;
    if (pk2 < 2 * pc.ke) {
      Wp += 64;
      Xp += 64;
      return;
    }
    cursor_next(pc, a, wg);
    if (pc.mode != 2) prod_set_tile();
  };

  // ---- consumer: fragment addresses ----
  // fragment i of an operand = LDS rows base + i*16 + (lane & 15), chunk (lane >> 4) ^ key64(row)
  const int fa = lane & 15, fc = lane >> 4;
  const unsigned offA = (unsigned)((wn * 64 + fa) * 64 + ((fc ^ key64(fa)) * 16));
  const unsigned offB = (unsigned)(kOpBytes + (wm * 128 + fa) * 64 + ((fc ^ key64(fa)) * 16));

  f32x4 acc[4][8];   // [n-tile][m-tile]
  Frags<T> F0 = {}, F1 = {};

  if (HAS_BIAS) {   // visible to everybody behind the prologue's barriers
    for (int i = tid; i * 8 < a.N; i += 512)
      *reinterpret_cast<u32x4*>(lds + NS * kSlot + i * 16) = *reinterpret_cast<const u32x4*>(a.bias + i * 8);
  }
  // ---- prologue: NS half-stages in flight, fragments of half-stage 0 in F0 ----
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int g = 0; g < PP; ++g) produce_piece(g, s);
    produce_advance();
  }
  wait_vm<PP * (NS - 1)>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < 4) F0.a[j] = *reinterpret_cast<const frag*>(lds + offA + j * 1024);
    F0.b[j] = *reinterpret_cast<const frag*>(lds + offB + j * 1024);
  }
  wait_vm<PP * (NS - 2)>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int post = 0;   // phases left in which the previous epilogue's stores may stay in flight
  int ws = 0;   // ring slot of the current half-stage p (free for the DMA of p + NS); p + 1 is read from ws + 1

  // one phase: 32 MFMAs on CUR in 4 groups of 8 (one n-tile each), 3 fragment reads of the next half-stage into NXT and
  // one DMA piece per group, then wait + barrier
#define SK_PHASE(CUR, NXT, FIRSTK)                                                                                        \
  {                                                                                                                  \
    const int rs = ws + 1 == NS ? 0 : ws + 1;                                                                        \
    const unsigned char* rbase = lds + rs * kSlot;                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                  \
      if (!(kAbl & 4)) {                                                                                 \
        NXT.a[i] = *reinterpret_cast<const frag*>(rbase + offA + i * 1024);                                          \
        NXT.b[2 * i] = *reinterpret_cast<const frag*>(rbase + offB + (2 * i) * 1024);                                \
        NXT.b[2 * i + 1] = *reinterpret_cast<const frag*>(rbase + offB + (2 * i + 1) * 1024);                        \
      }                                                                                                              \
      if (!(kAbl & 1)) {                                                                                             \
        produce_piece(i, ws);                                                                                        \
      }                                                                                                              \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                \
        if (kAbl & 2) asm volatile("" ::"v"(CUR.a[i]), "v"(CUR.b[j]));                                               \
        else if (FIRSTK) acc[i][j] = T::mfma(CUR.b[j], CUR.a[i], f32x4{0.f, 0.f, 0.f, 0.f});                         \
        else acc[i][j] = T::mfma(CUR.b[j], CUR.a[i], acc[i][j]);                                                     \
      }                                                                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
    }                                                                                                                \
    produce_advance();                                                                                               \
    if (!(kAbl & 8)) {                                                                                               \
      /* the NS - 2 phases behind an epilogue leave its 32 output stores out of the count (vmcnt retires in order) */ \
      if (post > 0) {                                                                                                \
        wait_vm<VMN + 32>();                                                                                         \
        --post;                                                                                                      \
      } else {                                                                                                       \
        wait_vm<VMN>();                                                                                              \
      }                                                                                                              \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
      __builtin_amdgcn_s_barrier();                                                                                  \
    }                                                                                                                \
    ws = rs;                                                                                                         \
  }

  while (cc.mode != 2) {
    // ---- main loop of the item: 2 phases per 64-deep k-tile, the first k-tile starts the accumulators at zero ----
    SK_PHASE(F0, F1, true)
    SK_PHASE(F1, F0, false)
    for (int kk = cc.kb + 1; kk < cc.ke; ++kk) {
      SK_PHASE(F0, F1, false)
      SK_PHASE(F1, F0, false)
    }

    const int tm = cc.tile / a.tiles_n, tn = cc.tile - tm * a.tiles_n;
    const int m0 = tm * 256 + wm * 128, n0 = tn * 256 + wn * 64;   // this wave's corner
    const int g = lane >> 4, nl = 4 * fa;
    const int mleft = a.M - m0, nleft = a.N - n0;   // rows / columns of the piece that exist
    const unsigned rowb = (unsigned)a.N * 2u;
    const unsigned span = (mleft > 0 && nleft > 0) ? (unsigned)(mleft < 128 ? mleft : 128) * rowb : 0u;
    // lanes whose columns do not exist get an offset outside every descriptor (loads return 0, stores are dropped)
    const unsigned voff = nl < nleft ? (unsigned)(4 * g) * rowb + (unsigned)nl * 2u : 0x80000000u;
    const int mrem = mleft - 4 * g;   // row j*16 + r of this lane exists iff j*16 + r < mrem (a constant against one register)
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.R)) + ((size_t)m0 * a.N + n0) * 2, 0,
        HAS_RES ? span : 0u, 0x00020000);
    uint2 rr[8][4];
    auto load_res = [&](int j0, int j1) {
#pragma unroll
      for (int j = j0; j < j1; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const u32x2v t2 = __builtin_amdgcn_raw_buffer_load_b64(rres, j * 16 + r < mrem ? voff : 0x80000000u,
                                                                   (unsigned)(j * 16 + r) * rowb, 0);
          rr[j][r] = uint2{t2[0], t2[1]};
        }
    };
    bool finish = true;
    if (SK && (cc.kb != 0 || cc.ke != a.nk)) {
      // ---- a part of a stream-K tile: publish the partial sums, the last arriver (per wave) continues ----
      // slab 0 of a workgroup: its part that starts inside a tile (the first item of its range); slab 1: its part that
      // starts a tile it does not finish (the last item) -- both can be pending at once
      const int t = cc.tile - a.D;
      float* slab = a.slabs + (((size_t)wg * 2 + (cc.kb == 0 ? 1 : 0)) * kWaves + wave) * kSlabFloats;
      {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, kSlabFloats * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            u32x4 v;
            __builtin_memcpy(&v, &acc[i][j], 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, (i * 8 + j) * 1024, 16 /* sc1: write-through */);
          }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slab has left the CU (and its L2)
      // the workgroups that hold a part of this tile: w_a .. w_b (owner of unit u = ceil((u + 1) Gs / S) - 1)
      const long long ua = (long long)t * a.nk, ub = ua + a.nk - 1;
      const int w_a = (int)(((ua + 1) * a.Gs + a.S - 1) / a.S) - 1, w_b = (int)(((ub + 1) * a.Gs + a.S - 1) / a.S) - 1;
      int parts = 0;
      for (int w2 = w_a; w2 <= w_b; ++w2) parts += sk_begin(a, w2) < sk_begin(a, w2 + 1) ? 1 : 0;
      unsigned ticket = 0;
      if (lane == 0)
        ticket = __hip_atomic_fetch_add(a.counters + t * kWaves + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
      finish = (int)ticket == parts - 1;
      if (finish) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int w2 = w_a; w2 <= w_b; ++w2) {
          const int b2 = sk_begin(a, w2);
          if (w2 == wg || b2 >= sk_begin(a, w2 + 1)) continue;
          const float* other = a.slabs + (((size_t)w2 * 2 + (b2 > (int)ua ? 0 : 1)) * kWaves + wave) * kSlabFloats;
          const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(other), 0, kSlabFloats * 4, 0x00020000);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jh = 0; jh < 8; jh += 4) {   // 4 loads in flight, then their 4 adds (more spilled the accumulators)
              u32x4 v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j)
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(ro, lane * 16, (i * 8 + jh + j) * 1024, 0);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                f32x4 f;
                __builtin_memcpy(&f, &v[j], 16);
                acc[i][jh + j] += f;
              }
              __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (lane == 0) __hip_atomic_store(a.counters + t * kWaves + wave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // wave-uniform: did this wave issue the epilogue's 32 output stores?  (a piece wholly outside the matrix -- e.g. waves 6, 7
    // of every tile at N = 192 -- and a non-finishing stream-K part store nothing)
    const bool stored = finish && a.M > m0 && a.N > n0;
    if (stored) {
      // ---- epilogue of this wave's 128 x 64 piece, straight from the accumulators ----
      // acc[i][j][r]: output row m0 + j*16 + 4 (lane >> 4) + r, column n0 + 4 (lane & 15) + i: the four n-tiles give the
      // lane 4 consecutive columns = 8 bytes, lanes 0-15 one 128-byte line, a store instruction 4 whole lines
      float bias4[4] = {0.f, 0.f, 0.f, 0.f};
      if (HAS_BIAS) {
        const int nb = n0 + nl < a.N ? n0 + nl : 0;
        const uint2 b2 = *reinterpret_cast<const uint2*>(lds + NS * kSlot + nb * 2);
        bias4[0] = T::to_f32((unsigned short)(b2.x & 0xffffu));
        bias4[1] = T::to_f32((unsigned short)(b2.x >> 16));
        bias4[2] = T::to_f32((unsigned short)(b2.y & 0xffffu));
        bias4[3] = T::to_f32((unsigned short)(b2.y >> 16));
      }
      // buffer descriptors on the piece's corner (wave-uniform) + ONE 32-bit lane offset; the row of a store goes into the
      // scalar offset.  (Per-lane 64-bit addresses were hoisted and spilled by the compiler: 32 stores, each behind a
      // scratch reload and `s_waitcnt vmcnt(0)` -- every store waited for the one before and for the DMA pieces in flight.)
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<unsigned char*>(a.Y) + ((size_t)m0 * a.N + n0) * 2, 0, span, 0x00020000);
      if (HAS_RES) load_res(0, 4);   // residual rows: four row sets in flight, the other four requested two row sets later
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (HAS_RES && j == 2) load_res(4, 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) x[i] = acc[i][j][r] + bias4[i];
          if (ACT == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = x[i] < 0.f ? 0.f : x[i];   // NaN-propagating, like torch.relu
          }
          if (ACT == 2) {
            const f32x2 g01 = gelu_erf2(f32x2{x[0], x[1]}), g23 = gelu_erf2(f32x2{x[2], x[3]});
            x[0] = g01.x; x[1] = g01.y; x[2] = g23.x; x[3] = g23.y;
          }
          uint2 o = {T::pack2(x[0], x[1]), T::pack2(x[2], x[3])};
          if (HAS_RES) {
            // fp16(fp16(linear) + residual): the two roundings of `identity + linear(x)` in the reference's fp16 path
            const uint2 q = rr[j][r];
            const float y0 = T::to_f32((unsigned short)(o.x & 0xffffu)) + T::to_f32((unsigned short)(q.x & 0xffffu));
            const float y1 = T::to_f32((unsigned short)(o.x >> 16)) + T::to_f32((unsigned short)(q.x >> 16));
            const float y2 = T::to_f32((unsigned short)(o.y & 0xffffu)) + T::to_f32((unsigned short)(q.y & 0xffffu));
            const float y3 = T::to_f32((unsigned short)(o.y >> 16)) + T::to_f32((unsigned short)(q.y >> 16));
            o = uint2{T::pack2(y0, y1), T::pack2(y2, y3)};
          }
          if (!(kAbl & 16))
            __builtin_amdgcn_raw_buffer_store_b64(u32x2v{o.x, o.y}, ry, j * 16 + r < mrem ? voff : 0x80000000u,
                                                  (unsigned)(j * 16 + r) * rowb, 0);
        }
      }
    }
    // The fragments of the next item's first half-stage (read in the last phase) are dropped over the epilogue, which needs
    // their registers (kept live, the compiler spilled around every store -- and a scratch access is a vector-memory
    // operation that waits for every LDS-DMA piece in flight).  They are read again here; the barrier keeps a wave that is
    // already in the next phase from issuing its DMA piece into that slot before everybody has re-read it.
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < 4) F0.a[j] = *reinterpret_cast<const frag*>(lds + ws * kSlot + offA + j * 1024);
      F0.b[j] = *reinterpret_cast<const frag*>(lds + ws * kSlot + offB + j * 1024);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the relaxed wait (VMN + 32) is only sound behind 32 stores that were really issued: a wave without them has nothing
    // but LDS-DMA pieces in its queue, and vmcnt(VMN + 32) would let it past pieces the next phases read
    post = stored && !(kAbl & 16) ? NS - 2 : 0;
    cursor_next(cc, a, wg);
  }
#undef SK_PHASE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the producer's redundant fetches past the end
}

// ---- host side ----
int device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

bool sk_supported(int64_t M, int64_t N, int64_t K) {
  return M > 0 && N > 0 && K >= 128 && K % 64 == 0 && N % 8 == 0 && M <= 0x7fffffffLL && N <= 16384 &&
         255 * K * 2 + 64 < 0x7fffffffLL;
}

// plan of one launch; returns false when the problem has no work
bool sk_plan(int64_t M, int64_t N, int64_t K, int G, int flags, SkArgs& a) {
  const int64_t tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256, T = tiles_m * tiles_n;
  if (T > 0x3fffffff) return false;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_n = (int)tiles_n; a.T = (int)T; a.nk = (int)(K / 64);
  a.G = G;
  a.np = 0;
  if (flags & 0x20) {   // one workgroup per tile
    a.G = (int)T; a.D = (int)T; a.S = 0; a.Gs = 0; a.np = 1;
    return true;
  }
  const bool no_sk = (flags & 0x40) == 0;   // stream-K split of the left-over tiles only on request (see the header)
  int rounds = (int)(T / G);
  int rem = (int)(T - (int64_t)rounds * G);
  if (no_sk && rem > 0) {   // data-parallel only (A/B): the left-over tiles are whole items of the first `rem` workgroups
    a.D = rounds * G; a.S = rem * a.nk; a.Gs = rem;
    return true;
  }
  a.D = rounds * G;
  a.S = rem * a.nk;
  // workgroups of the stream-K region: every workgroup when that leaves each tile in <= kMaxParts parts, else rem * kMaxParts
  int gs = G;
  if ((int64_t)rem * kMaxParts < gs) gs = rem * kMaxParts;
  if (gs > a.S) gs = a.S;
  a.Gs = gs;
  return true;
}

int64_t sk_slab_bytes(int G) { return (int64_t)G * 2 * kWaves * kSlabFloats * 4; }
int64_t sk_workspace_bytes(int G) { return sk_slab_bytes(G) + (int64_t)G * kWaves * 4; }

template <class T, int ACT, int NS, bool SK>
int launch_sk_act(hipStream_t st, const SkArgs& a, bool has_bias, bool has_res) {
  const dim3 grid((unsigned)a.G), block(512);
#define CODETR_SK(HB, HR) hipLaunchKernelGGL((linear_sk_kernel<T, ACT, HB, HR, NS, SK>), grid, block, 0, st, a)
  if (has_bias && has_res) CODETR_SK(true, true);
  else if (has_bias) CODETR_SK(true, false);
  else if (has_res) CODETR_SK(false, true);
  else CODETR_SK(false, false);
#undef CODETR_SK
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

template <class T>
int launch_sk(hipStream_t st, const void* X, const void* W, const void* bias, const void* R, void* Y, int64_t M, int64_t N,
              int64_t K, int act, void* ws, int64_t ws_bytes, int flags) {
  if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return CODETR_E_BADARG;
  if (act < 0 || act > 2 || !sk_supported(M, N, K)) return CODETR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Y) |
       reinterpret_cast<uintptr_t>(R) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(ws)) & 15)
    return CODETR_E_BADARG;
  const int G = device_cus() / 8 * 8;
  if (G <= 0) return CODETR_E_BADARG;
  // the workspace is only touched by the stream-K split
  if ((flags & 0x40) && (!ws || ws_bytes < sk_workspace_bytes(G))) return CODETR_E_BADARG;
  SkArgs a;
  if (!sk_plan(M, N, K, G, flags, a)) return CODETR_E_TOO_LARGE;
  a.X = static_cast<const unsigned char*>(X);
  a.W = static_cast<const unsigned char*>(W);
  a.bias = static_cast<const unsigned short*>(bias);
  a.R = static_cast<const unsigned short*>(R);
  a.Y = static_cast<unsigned short*>(Y);
  a.slabs = static_cast<float*>(ws);
  a.counters = ws ? reinterpret_cast<unsigned*>(static_cast<unsigned char*>(ws) + sk_slab_bytes(G)) : nullptr;
  const bool hb = bias != nullptr, hr = R != nullptr;
  // a plan whose left-over tiles are whole items (no split) runs the instantiation without the partial-tile path
  const bool split = a.Gs > 0 && a.S != a.Gs * a.nk;
#define CODETR_SK_NS(ACT) return split ? launch_sk_act<T, ACT, 4, true>(st, a, hb, hr) : launch_sk_act<T, ACT, 4, false>(st, a, hb, hr)
  switch (act) {
    case 0: CODETR_SK_NS(0);
    case 1: CODETR_SK_NS(1);
    default: CODETR_SK_NS(2);
  }
#undef CODETR_SK_NS
}

}  // namespace

extern "C" {

int64_t codetr_linear_sk_workspace_bytes(void) { return sk_workspace_bytes(device_cus() / 8 * 8); }

int codetr_linear_sk_supported(int64_t M, int64_t N, int64_t K) { return sk_supported(M, N, K) ? 1 : 0; }

// Where the persistent kernel measured faster than the 256-tile kernel of gemm_f16.hip (tools/micro/gemm_sk_bench on the
// 4- and 8-image Swin-L shapes, profiles/r04_gemm_sk.txt): K < 1536 with at least a tile per CU: -2 ... -25 %; at K = 1536 it
// is level; the long-K
// layers (fc2 of stages 2 / 3) are level to 5 % slower and stay on the old kernel.
int codetr_linear_sk_preferred(int64_t M, int64_t N, int64_t K, int act, int has_residual) {
  (void)act;
  (void)has_residual;
  if (!sk_supported(M, N, K) || K >= 1536) return 0;
  const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
  const int64_t tn = (N + 255) / 256;
  // little of the 256-wide tile wasted (the rule of the 256-tile kernel)
  return tiles >= device_cus() && (N * 8 >= tn * 256 * 7 || (tn == 1 && N * 4 >= 256 * 3)) ? 1 : 0;
}

int codetr_linear_sk_f16(void* stream, const void* x_dev, const void* w_dev, const void* bias_dev, const void* residual_dev,
                         void* y_dev, int64_t M, int64_t N, int64_t K, int act, void* workspace_dev, int64_t workspace_bytes,
                         int flags) {
  return launch_sk<HalfT>(static_cast<hipStream_t>(stream), x_dev, w_dev, bias_dev, residual_dev, y_dev, M, N, K, act,
                          workspace_dev, workspace_bytes, flags);
}

int codetr_linear_sk_bf16(void* stream, const void* x_dev, const void* w_dev, const void* bias_dev, const void* residual_dev,
                          void* y_dev, int64_t M, int64_t N, int64_t K, int act, void* workspace_dev,
                          int64_t workspace_bytes, int flags) {
#if CODETR_SK_ABL
  return CODETR_E_UNSUPPORTED;   // diagnostic builds carry the fp16 instantiations only
#endif
  return launch_sk<BFloatT>(static_cast<hipStream_t>(stream), x_dev, w_dev, bias_dev, residual_dev, y_dev, M, N, K, act,
                            workspace_dev, workspace_bytes, flags);
}

}  // extern "C"
