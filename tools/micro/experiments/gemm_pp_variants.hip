// EXPERIMENT, not part of libcodetr_hip.so (round 6): csrc/gemm_pp.hip with every main-loop variant that was built and
// measured beside the one that ships (linear_pp_kernel, half-stage image, one LOAD + one MFMA phase per half-stage):
//   flags 2        linear_pp_kernel PH = 2   the half-stage in two phases of 16 MFMAs (4 barriers per half-stage): level
//   flags 0        linear_pp2_kernel         K-TILE image: 128-byte LDS rows, a DMA piece = 8 whole cache lines, W ring of 2 /
//                                            X ring of 3 tiles: 3-7 % ahead at K >= 1536 on two boxes, 3-10 % behind on a third
//   flags 3        linear_pp3_kernel         ONE barrier per half-stage (the groups run LOAD / MFMA in opposite order, ring
//                                            of 5): level -- a wave's own LOAD + MFMA chain (450-650 + 540 cycles), not the
//                                            second barrier, is what the period follows
//   flags 4        linear_pp2_kernel + alt   odd strips issue their DMA pieces before their fragment reads: 3-6 % slower
// (the register-staged form is in gemm_pp_regstage.hip).  Numbers and in-kernel stamps: profiles/r06_gemm_pp.txt.
// Build for the harness:  hipcc ... -shared tools/micro/experiments/gemm_pp_variants.hip  (same flags as csrc/gemm_pp.o)
// Ping-pong GEMM for the long-K linears of the hot path (Swin stages 2-3: reference codetr/swin.py:92-112 qkv / proj,
// :331-352 the MLP):  Y[M,N] = act(X[M,K] . W[N,K]^T + bias[N]) (+ R[M,N]),  fp16 / bf16 storage, fp32 accumulation on
// v_mfma_f32_16x16x32_{f16,bf16}.
//
// Why a third main loop (round 6).  The ablations of the two earlier ones (profiles/r03_gemm256_ablation.txt,
// profiles/r04_gemm_sk.txt) say the same thing: the bare MFMA stream runs at the matrix pipe's ideal, and the LDS fragment
// reads (+45 %) and the LDS-DMA issue (+25 %) ADD to it instead of hiding under it.  Both kernels run the two waves of a
// SIMD through the same instruction mix at the same time (one barrier per phase keeps them in lockstep): when the LDS
// queue is full both stall, in order, in front of their MFMAs.  Here the two waves of a SIMD take turns instead
// (cdna_hip_programming.md section 5, the 8-phase template; MI355X_MICROARCH.md "Two waves per SIMD" item 9):
//
//   * 512 threads = 8 waves as 2 GROUPS (wave >> 2 = the m half of the 256 x 256 tile: one wave of each group on every
//     SIMD) x 4 (wave & 3 = a 64-column strip); a wave owns 128 x 64 outputs = 8 x 4 MFMA tiles, 128 accumulators.
//   * a phase of a wave is a LOAD segment (12 ds_read_b128 of the fragments of one 32-deep half-stage, its 4 LDS-DMA pieces
//     of a later half-stage, the counted waits) and an MFMA segment (the 32 MFMAs of that half-stage, nothing else, at
//     raised priority), each closed by a workgroup barrier.  Group 1 runs ONE barrier behind group 0, so in every interval
//     between two barriers one wave of each SIMD feeds the matrix pipe while the other one loads; no MFMA is ever issued
//     behind a memory instruction of its own wave.  (PH = 2: the half-stage in two such phases of 16 MFMAs.)
//   * operands: the ring of NS = 4 half-stages of csrc/gemm_sk.hip (W[256 rows][64 B] + X[256 rows][64 B] = 32 KiB per
//     slot, LDS-DMA with a scalar base + per-thread offset, swizzle on the source address), filled NS - 1 half-stages
//     ahead of the reads; the stream does not stop at tile boundaries (persistent workgroups, one per CU).
//       WAR: half-stage p is read in L(p) -- group 0 in interval 2p, group 1 in 2p + 1 -- and every wave waits for its own
//            reads (lgkmcnt(0)) BEFORE the barrier that closes its LOAD segment; slot p is refilled from L(p + 1) on
//            (interval 2p + 2 at the earliest).
//       RAW: the pieces of half-stage q are issued in L(q - NS + 1); every wave retires its own at the end of L(q - 1)
//            with vmcnt(4 (NS - 2)) (vmcnt retires in order) -- group 1 in interval 2q - 1, one barrier before the first
//            read of q (group 0, interval 2q).
//   * tile boundary: group 0 waits one barrier before its epilogue so that both groups run their epilogues together
//     (back to back they would serialise: a wave's epilogue takes ten MFMA segments), group 1 waits one barrier at the
//     start of a tile to fall behind again: 2 n + 1 barriers per tile for both (n = phases of the tile).
//   * epilogue, work list, buffer stores: as csrc/gemm_sk.hip (permuted weight rows: 16 adjacent lanes write one whole
//     128-byte line straight from the accumulators), without its stream-K split.
//
// Requirements: K % 64 == 0, K >= 128, N % 8 == 0, N <= 16384, dense row-major operands, 16-byte aligned bases; M, N
// otherwise arbitrary (edge tiles clamp their loads and mask their stores).  No row mask / head-major output.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "codetr_hip.h"
#include "gemm_elem.h"

using namespace codetr_gemm;

namespace {

// diagnostic builds only (-DCODETR_PP_ABL=mask gives WRONG results by construction): 1 = no LDS-DMA inside the main loop,
// 2 = no MFMAs, 4 = no fragment reads, 16 = no output stores
#ifndef CODETR_PP_ABL
#define CODETR_PP_ABL 0
#endif
constexpr int kAbl = CODETR_PP_ABL;
// diagnostic build only (tools/micro/pp_stamps.hip): where every wave's cycles go -- per wave, sums over the main loops of
// [0] LOAD segment until everything is issued, [1] waiting for the staged data (vmcnt), [2] waiting for its LDS operations,
// [3] at the barrier behind the LOAD segment, [4] MFMA segment, [5] at the barrier behind it, [6] epilogue, [7] whole kernel
#ifdef CODETR_PP_STAMPS
__device__ unsigned long long* g_pp_stamps = nullptr;
#define PP_T(i) const unsigned long long pp_t##i = __builtin_readcyclecounter()
#define PP_ACC(k, a, b) pp_acc[k] += pp_t##b - pp_t##a
#define PP_STAMPS_OUT()                                                                    \
  if (lane == 0 && g_pp_stamps) {                                                          \
    pp_acc[7] = __builtin_readcyclecounter() - pp_k0;                                      \
    unsigned long long* o = g_pp_stamps + ((size_t)blockIdx.x * 8 + wave) * 8;             \
    for (int i = 0; i < 8; ++i) o[i] = pp_acc[i];                                          \
  }
#else
#define PP_T(i)
#define PP_ACC(k, a, b)
#define PP_STAMPS_OUT()
#endif
typedef unsigned u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned))));

constexpr int kSlot = 32768;     // one half-stage: W[256][64 B] then X[256][64 B]
constexpr int kOpBytes = 16384;
constexpr int kBiasBytes = 32768;     // bias in LDS: N <= 16384

struct PpArgs {
  const unsigned char* X;
  const unsigned char* W;
  const unsigned short* bias;
  const unsigned short* R;
  unsigned short* Y;
  int M, N, K;
  int tiles_n, T, nk;  // nk = K / 64
  int G;               // workgroups (a multiple of 8)
  int rounds;          // whole rounds of G tiles
  int rem;             // T - rounds * G left-over tiles: one more item of workgroups 0 .. rem-1 (taken first)
  int alt;             // 1: the odd strips (wave & 1) issue their LDS-DMA pieces BEFORE their fragment reads, the even ones after
};

// swizzle key of a 64-byte LDS row (4 chunks of 16 B): conflict-free under ds_read_b128's lane groups
// (tests/test_lds_bank_model.py)
__device__ __forceinline__ int key64(int row) {
  const int q = (row >> 2) & 3;
  return q ^ ((q & 1) << 1);
}

__device__ __forceinline__ void lds_dma16(const unsigned char* src, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds_addr)
               : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// a workgroup barrier that nothing is scheduled across (MFMAs are register-only: the scheduler would otherwise move them
// past the barrier into the other group's segment)
__device__ __forceinline__ void seg_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// tile of item `idx` of workgroup w, -1 past the end.  Left-over tiles first (whole items of workgroups 0 .. rem-1), then
// the rounds: the 32 workgroups of an XCD (w mod 8) walk consecutive tiles, n fastest.
__device__ __forceinline__ int item_tile(const PpArgs& a, int w, int idx) {
  const int D = a.rounds * a.G;
  if (w < a.rem) {
    if (idx == 0) return D + w;
    --idx;
  }
  if (idx >= a.rounds) return -1;
  return (w & 7) * (D >> 3) + idx * (a.G >> 3) + (w >> 3);
}

// ---- epilogue of a wave's 128 x 64 piece of tile `tile`, straight from the accumulators (shared by both main loops) ----
// Returns whether this wave issued the 32 output stores (wave-uniform: a piece wholly outside the matrix stores nothing).
template <class T, int ACT, bool HAS_BIAS, bool HAS_RES, bool BIAS_LDS = true>
__device__ __forceinline__ bool pp_epilogue(f32x4 (&acc)[4][8], const PpArgs& a, int tile, int grp, int wn, int lane,
                                            const unsigned char* bias_lds) {
  const int fa = lane & 15;
  const int tm = tile / a.tiles_n, tn = tile - tm * a.tiles_n;
  const int m0 = tm * 256 + grp * 128, n0 = tn * 256 + wn * 64;   // this wave's corner
  const int g = lane >> 4, nl = 4 * fa;
  const int mleft = a.M - m0, nleft = a.N - n0;   // rows / columns of the piece that exist
  const unsigned rowb = (unsigned)a.N * 2u;
  const unsigned span = (mleft > 0 && nleft > 0) ? (unsigned)(mleft < 128 ? mleft : 128) * rowb : 0u;
  // lanes whose columns do not exist get an offset outside every descriptor (loads return 0, stores are dropped)
  const unsigned voff = nl < nleft ? (unsigned)(4 * g) * rowb + (unsigned)nl * 2u : 0x80000000u;
  const int mrem = mleft - 4 * g;   // row j*16 + r of this lane exists iff j*16 + r < mrem
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
  // wave-uniform: does this wave issue the epilogue's 32 output stores?  (a piece wholly outside the matrix stores nothing)
  const bool stored = a.M > m0 && a.N > n0;
  if (stored) {
    // ---- epilogue of this wave's 128 x 64 piece, straight from the accumulators ----
    // acc[i][j][r]: output row m0 + j*16 + 4 (lane >> 4) + r, column n0 + 4 (lane & 15) + i: the four n-tiles give the
    // lane 4 consecutive columns = 8 bytes, lanes 0-15 one 128-byte line, a store instruction 4 whole lines
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (HAS_BIAS) {
      const int nb = n0 + nl < a.N ? n0 + nl : 0;
      const uint2 b2 = BIAS_LDS ? *reinterpret_cast<const uint2*>(bias_lds + nb * 2)
                                : *reinterpret_cast<const uint2*>(a.bias + nb);   // (N % 8 == 0: 8-byte aligned, in range)
      bias4[0] = T::to_f32((unsigned short)(b2.x & 0xffffu));
      bias4[1] = T::to_f32((unsigned short)(b2.x >> 16));
      bias4[2] = T::to_f32((unsigned short)(b2.y & 0xffffu));
      bias4[3] = T::to_f32((unsigned short)(b2.y >> 16));
    }
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
  return stored;
}

template <class T, int ACT, bool HAS_BIAS, bool HAS_RES, int NS, int PH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void linear_pp_kernel(const PpArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NS * kSlot + (HAS_BIAS ? kBiasBytes : 0)];   // ring, then the bias
  using frag = typename T::frag;
  constexpr int PP = 4;               // DMA pieces per wave and half-stage (2 of W, 2 of X)
  constexpr int VMN = PP * (NS - 2);  // pieces that may stay in flight at the end of a LOAD segment

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;   // group = m half (rows grp*128), strip = columns wn*64
  const int wg = blockIdx.x;
  const int K = a.K, K2 = K * 2;
  const int nh = 2 * a.nk;                    // half-stages (phases) of a tile
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);

#ifdef CODETR_PP_STAMPS
  unsigned long long pp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pp_k0 = __builtin_readcyclecounter();
#endif
  int c_idx = 0;
  int c_tile = item_tile(a, wg, 0);
  if (c_tile < 0) return;   // nothing to do (wave-uniform for the whole workgroup: no barrier has been executed)

  // ---- producer: this wave's 2 + 2 pieces of a half-stage, NS - 1 half-stages ahead of the reads ----
  // piece P = wave + 8 q covers LDS rows P*16 .. P*16+15 (64 B each): lane -> row P*16 + (lane >> 2), position lane & 3,
  // which holds source chunk (lane & 3) ^ key64(row).  W rows are permuted: LDS row q*64 + i*16 + c holds weight row
  // q*64 + 4 c + i of the tile (see the epilogue).
  unsigned voffW[2], voffX[2];
  const unsigned char* Wp = a.W;
  const unsigned char* Xp = a.X;
  int p_idx = 0, p_tile = c_tile, p_h = 0;   // producer's item, its tile (-1: past the end), next half-stage of it
  auto prod_set_tile = [&]() {
    const int tm = p_tile / a.tiles_n, tn = p_tile - tm * a.tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int nmax = a.N - 1 - n0, mmax = a.M - 1 - m0;   // edge tiles: the last row again (its outputs are never stored)
    int ln = lane;
    asm volatile("" : "+v"(ln));   // recomputed at every tile switch instead of kept (and spilled) across the main loop
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = (wave + 8 * q) * 16 + (ln >> 2);
      const int rw = (r & 192) + 4 * (r & 15) + ((r >> 4) & 3);
      const unsigned co = (unsigned)((((ln & 3) ^ key64(r)) * 16));
      const int rn = rw < nmax ? rw : nmax, rm = r < mmax ? r : mmax;
      voffW[q] = (unsigned)rn * (unsigned)K2 + co;
      voffX[q] = (unsigned)rm * (unsigned)K2 + co;
    }
    Wp = a.W + (size_t)n0 * K * 2;
    Xp = a.X + (size_t)m0 * K * 2;
    p_h = 0;
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
    if (p_tile < 0) return;
    ++p_h;
    if (p_h < nh) {
      Wp += 64;
      Xp += 64;
      return;
    }
    p_tile = item_tile(a, wg, ++p_idx);
    if (p_tile >= 0) prod_set_tile();
  };

  // ---- consumer: fragment addresses ----
  // fragment i of an operand = LDS rows base + i*16 + (lane & 15), chunk (lane >> 4) ^ key64(row)
  const int fa = lane & 15, fc = lane >> 4;
  const unsigned offA = (unsigned)((wn * 64 + fa) * 64 + ((fc ^ key64(fa)) * 16));
  const unsigned offB = (unsigned)(kOpBytes + (grp * 128 + fa) * 64 + ((fc ^ key64(fa)) * 16));

  f32x4 acc[4][8];   // [n-tile][m-tile]
  frag fw[4], fx[8]; // W rows (MFMA B operand): 4 n-tiles; X rows (MFMA A operand): 8 m-tiles

  if (HAS_BIAS) {   // visible to everybody behind the prologue's barrier
    for (int i = tid; i * 8 < a.N; i += 512)
      *reinterpret_cast<u32x4*>(lds + NS * kSlot + i * 16) = *reinterpret_cast<const u32x4*>(a.bias + i * 8);
  }
  // ---- prologue: half-stages 0 .. NS-2 in flight, half-stage 0 landed for everybody ----
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
#pragma unroll
    for (int g = 0; g < PP; ++g) produce_piece(g, s);
    produce_advance();
  }
  wait_vm<VMN>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the bias rows written above
  seg_barrier();
  int post = 0;   // LOAD segments left in which the previous epilogue's stores may stay in flight
  int ws = 0;     // ring slot of the current half-stage p; slot ws - 1 is free for the DMA of p + NS - 1

  // LOAD segment, part `part` of PH: the fragment reads of this part, PP / PH pieces of half-stage p + NS - 1 into the slot
  // p - 1 left, and (last part) the counted wait that retires half-stage p + 1; every wave waits for its own reads before
  // it arrives at the barrier.
  auto load_seg = [&](int part) {
    PP_T(0);
    const unsigned char* rbase = lds + ws * kSlot;
    const int fs = ws == 0 ? NS - 1 : ws - 1;
    if (!(kAbl & 4)) {
      if (part == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const frag*>(rbase + offA + i * 1024);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (PH == 1 || (j >> 2) == part) fx[j] = *reinterpret_cast<const frag*>(rbase + offB + j * 1024);
    }
    if (!(kAbl & 1)) {
#pragma unroll
      for (int g = 0; g < PP; ++g)
        if (PH == 1 || (g >> 1) == part) produce_piece(g, fs);
    }
    if (part == PH - 1) produce_advance();
    PP_T(1);
    if (part == PH - 1) {
      // the NS - 2 LOAD segments behind an epilogue leave its 32 output stores out of the count (vmcnt retires in order)
      if (post > 0) {
        wait_vm<VMN + 32>();
        --post;
      } else {
        wait_vm<VMN>();
      }
    }
    PP_T(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_T(3);
    seg_barrier();
    PP_T(4);
    PP_ACC(0, 0, 1); PP_ACC(1, 1, 2); PP_ACC(2, 2, 3); PP_ACC(3, 3, 4);
  };
  // MFMA segment, part `part` of PH
  auto mfma_seg = [&](int part, bool firstk) {
    PP_T(5);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (PH == 2 && (j >> 2) != part) continue;
        if (kAbl & 2) asm volatile("" ::"v"(fw[i]), "v"(fx[j]));
        else if (firstk) acc[i][j] = T::mfma(fx[j], fw[i], f32x4{0.f, 0.f, 0.f, 0.f});
        else acc[i][j] = T::mfma(fx[j], fw[i], acc[i][j]);
      }
    __builtin_amdgcn_s_setprio(0);
    PP_T(6);
    seg_barrier();
    PP_T(7);
    PP_ACC(4, 5, 6); PP_ACC(5, 6, 7);
  };

  while (c_tile >= 0) {
    if (grp == 1) seg_barrier();   // group 1 falls one barrier behind
    // ---- main loop of the tile: the first half-stage starts the accumulators at zero ----
#pragma unroll
    for (int part = 0; part < PH; ++part) {
      load_seg(part);
      mfma_seg(part, true);
    }
    ws = ws + 1 == NS ? 0 : ws + 1;
    for (int h = 1; h < nh; ++h) {
#pragma unroll
      for (int part = 0; part < PH; ++part) {
        load_seg(part);
        mfma_seg(part, false);
      }
      ws = ws + 1 == NS ? 0 : ws + 1;
    }
    if (grp == 0) seg_barrier();   // group 0 waits for group 1's last MFMA segment: both epilogues run together

    PP_T(8);
    const bool stored = pp_epilogue<T, ACT, HAS_BIAS, HAS_RES>(acc, a, c_tile, grp, wn, lane, lds + NS * kSlot);
    PP_T(9);
    PP_ACC(6, 8, 9);
    // the relaxed wait (VMN + 32) is only sound behind 32 stores that were really issued: a wave without them has nothing
    // but LDS-DMA pieces in its queue
    post = stored && !(kAbl & 16) ? NS - 2 : 0;
    c_tile = item_tile(a, wg, ++c_idx);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the producer's redundant fetches past the end
  PP_STAMPS_OUT();
}

// ---------------------------------------------------------------------------------------------------------------------
// K-TILE IMAGE form of the ping-pong loop (the default).  What the in-kernel stamps of the half-stage form say
// (profiles/r06_gemm_pp.txt, swin2.fc2): an MFMA segment takes 505-540 cycles, a LOAD segment 450-520 -- 240 for the 12
// fragment reads and ~100 for EACH of the four LDS-DMA pieces: a piece of the half-stage image is 16 rows x 64 bytes, sixteen
// half cache lines per instruction.  Here an operand tile is 64 deep: LDS rows of 128 bytes, a piece = 8 rows x one WHOLE
// 128-byte line (cdna_hip_programming.md section 5: "x through LDS in full 128-B lines"), the swizzle of the 256-tile kernel
// (chunk position = chunk ^ ((row >> 1) & 7)).  Two phases per k-tile (k-step 0, k-step 1), each 12 reads + 4 pieces + 32
// MFMAs, the same barriers and the same group stagger as above.  Rings: W two tiles (W(t + 1) issued in L(t, 0), two LOAD
// periods ahead of its first read), X three tiles (X(t + 2) issued in L(t, 1), three periods ahead): 5 x 32 KiB = the whole
// LDS, so the bias is read from global memory in the epilogue.
//   WAR: W slot (t + 1) & 1 held W(t - 1), X slot (t + 2) % 3 held X(t - 1); their last reads are group 1's L(t - 1, 1) in
//        interval 4t - 1, retired by lgkmcnt(0) before barrier 4t; the earliest refill is group 0's L(t, 0) in interval 4t.
//   RAW: every wave waits vmcnt(4) at the end of L(t, 1) -- only its 4 pieces of X(t + 2) may stay in flight, so W(t + 1)
//        and X(t + 1) have landed -- group 1 in interval 4t + 3, one barrier before group 0 reads them in interval 4t + 4.
//        (An epilogue's 32 output stores sit in the same queue in front of the next tile's pieces: they have had two LOAD
//        periods by the time that wait asks for them.)
constexpr int kTileOp = 32768;   // one operand tile: 256 rows x 128 B

template <class T, int ACT, bool HAS_BIAS, bool HAS_RES>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void linear_pp2_kernel(const PpArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[5 * kTileOp];   // [W0 W1][X0 X1 X2]
  using frag = typename T::frag;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;   // group = m half (rows grp*128), strip = columns wn*64
  const int wg = blockIdx.x;
  const int K = a.K, K2 = K * 2;
  const int nk = a.nk;                        // k-tiles of an output tile
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);

#ifdef CODETR_PP_STAMPS
  unsigned long long pp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pp_k0 = __builtin_readcyclecounter();
#endif
  int c_idx = 0;
  int c_tile = item_tile(a, wg, 0);
  if (c_tile < 0) return;   // nothing to do (uniform for the whole workgroup: no barrier has been executed)

  // ---- producers: W one k-tile ahead of the reads, X two.  Piece P = wave + 8 q (q = 0 .. 3) of an operand tile covers LDS
  // rows P*8 .. P*8+7: lane -> row P*8 + (lane >> 3), position lane & 7, which holds source chunk (lane & 7) ^ ((row >> 1) & 7).
  // W rows are permuted: LDS row q*64 + i*16 + c holds weight row q*64 + 4 c + i of the tile (see the epilogue).
  struct Prod {
    unsigned voff[4];
    const unsigned char* base;   // first k-tile not yet issued, row 0 of the tile
    int idx, tile, kt;
  };
  Prod pw, px;
  auto prod_set_tile = [&](Prod& p, bool is_w) {
    const int tm = p.tile / a.tiles_n, tn = p.tile - tm * a.tiles_n;
    const int r0 = is_w ? tn * 256 : tm * 256;
    const int rmax = (is_w ? a.N : a.M) - 1 - r0;   // edge tiles: the last row again (its outputs are never stored)
    int ln = lane;
    asm volatile("" : "+v"(ln));   // recomputed at every tile switch instead of kept (and spilled) across the main loop
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = (wave + 8 * q) * 8 + (ln >> 3);
      const int rs = is_w ? (r & 192) + 4 * (r & 15) + ((r >> 4) & 3) : r;
      const unsigned co = (unsigned)((((ln & 7) ^ ((r >> 1) & 7)) * 16));
      p.voff[q] = (unsigned)(rs < rmax ? rs : rmax) * (unsigned)K2 + co;
    }
    p.base = (is_w ? a.W : a.X) + (size_t)r0 * K * 2;
    p.kt = 0;
  };
  auto prod_init = [&](Prod& p, bool is_w) {
    p.idx = 0;
    p.tile = c_tile;
    prod_set_tile(p, is_w);
  };
  // the 4 pieces of the producer's current k-tile into the operand slot at LDS byte address `slot`, then on to the next
  // k-tile (past the end of the list: the last one again -- nobody reads it; the counted waits stay uniform)
  auto produce = [&](Prod& p, bool is_w, unsigned slot) {
    if (!(kAbl & 1)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) lds_dma16(p.base, p.voff[q], slot + (unsigned)(wave + 8 * q) * 1024u);
    }
    if (p.tile < 0) return;
    if (++p.kt < nk) {
      p.base += 128;
      return;
    }
    p.tile = item_tile(a, wg, ++p.idx);
    if (p.tile >= 0) prod_set_tile(p, is_w);
  };
  prod_init(pw, true);
  prod_init(px, false);

  // ---- consumer: fragment addresses.  Fragment i of an operand = LDS rows base + i*16 + (lane & 15), k-chunk
  // ks*4 + (lane >> 4) at position chunk ^ ((row >> 1) & 7): the two k-steps differ in bit 2 of the position, an XOR of 64
  const int fa = lane & 15, fc = lane >> 4;
  const unsigned offA = (unsigned)((wn * 64 + fa) * 128 + ((fc ^ ((fa >> 1) & 7)) * 16));
  const unsigned offB = (unsigned)((grp * 128 + fa) * 128 + ((fc ^ ((fa >> 1) & 7)) * 16));

  f32x4 acc[4][8];   // [n-tile][m-tile]
  frag fw[4], fx[8]; // W rows (MFMA B operand): 4 n-tiles; X rows (MFMA A operand): 8 m-tiles

  // ---- prologue: W(0), X(0) landed for everybody, X(1) in flight ----
  produce(pw, true, lds0);
  produce(px, false, lds0 + 2 * kTileOp);
  produce(px, false, lds0 + 3 * kTileOp);
  wait_vm<4>();
  seg_barrier();
  int wsl = 0, xsl = 0;   // slots of W(t), X(t)

  // The four LOAD-segment waves of an interval (one per SIMD) would all ask the LDS for their 12 KiB of fragments first and
  // the texture path for their 4 KiB of pieces afterwards -- each unit saturated in turn, idle otherwise.  With a.alt the
  // odd strips issue their pieces first: the two units work side by side.
  const bool dma_first = a.alt && (wn & 1);
  auto load_seg = [&](int ks) {
    PP_T(0);
    const unsigned char* wb = lds + wsl * kTileOp;
    const unsigned char* xb = lds + (2 + xsl) * kTileOp;
    auto reads = [&]() {
      if (kAbl & 4) return;
#pragma unroll
      for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const frag*>(wb + ((offA + i * 2048) ^ (unsigned)(ks * 64)));
#pragma unroll
      for (int j = 0; j < 8; ++j) fx[j] = *reinterpret_cast<const frag*>(xb + ((offB + j * 2048) ^ (unsigned)(ks * 64)));
    };
    auto pieces = [&]() {
      if (ks == 0) {
        produce(pw, true, lds0 + (unsigned)(wsl ^ 1) * kTileOp);                         // W(t + 1) -> the slot W(t - 1) left
      } else {
        const int x2 = xsl == 0 ? 2 : xsl - 1;                                            // (xsl + 2) % 3
        produce(px, false, lds0 + (unsigned)(2 + x2) * kTileOp);                          // X(t + 2) -> the slot X(t - 1) left
      }
    };
    if (dma_first) {
      pieces();
      __builtin_amdgcn_sched_barrier(0);
      reads();
    } else {
      reads();
      __builtin_amdgcn_sched_barrier(0);
      pieces();
    }
    PP_T(1);
    if (ks == 1) wait_vm<4>();
    PP_T(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_T(3);
    seg_barrier();
    PP_T(4);
    PP_ACC(0, 0, 1); PP_ACC(1, 1, 2); PP_ACC(2, 2, 3); PP_ACC(3, 3, 4);
  };
  auto mfma_seg = [&](bool firstk) {
    PP_T(5);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (kAbl & 2) asm volatile("" ::"v"(fw[i]), "v"(fx[j]));
        else if (firstk) acc[i][j] = T::mfma(fx[j], fw[i], f32x4{0.f, 0.f, 0.f, 0.f});
        else acc[i][j] = T::mfma(fx[j], fw[i], acc[i][j]);
      }
    __builtin_amdgcn_s_setprio(0);
    PP_T(6);
    seg_barrier();
    PP_T(7);
    PP_ACC(4, 5, 6); PP_ACC(5, 6, 7);
  };

  while (c_tile >= 0) {
    if (grp == 1) seg_barrier();   // group 1 falls one barrier behind
    load_seg(0);
    mfma_seg(true);
    load_seg(1);
    mfma_seg(false);
    wsl ^= 1;
    xsl = xsl == 2 ? 0 : xsl + 1;
    for (int t = 1; t < nk; ++t) {
      load_seg(0);
      mfma_seg(false);
      load_seg(1);
      mfma_seg(false);
      wsl ^= 1;
      xsl = xsl == 2 ? 0 : xsl + 1;
    }
    if (grp == 0) seg_barrier();   // group 0 waits for group 1's last MFMA segment: both epilogues run together
    PP_T(8);
    pp_epilogue<T, ACT, HAS_BIAS, HAS_RES, false>(acc, a, c_tile, grp, wn, lane, nullptr);
    PP_T(9);
    PP_ACC(6, 8, 9);
    c_tile = item_tile(a, wg, ++c_idx);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the producers' redundant fetches past the end
  PP_STAMPS_OUT();
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE BARRIER PER HALF-STAGE (flags 3).  The stamps of the two forms above (profiles/r06_gemm_pp.txt) put ~60-190 cycles of
// every 1 375-cycle half-stage at each of its two barriers: the barrier behind the MFMA segment orders nothing -- it only
// keeps the two groups alternating.  Here the groups run the two segments in OPPOSITE ORDER between two barriers instead:
//     epoch k (between barriers k and k + 1):   group 0:  M(k)  L(k + 1)        group 1:  L(k)  M(k)
// so one wave of every SIMD multiplies while the other loads in both halves of the epoch, with one barrier per half-stage.
// Ring of NS = 5 half-stages (the whole LDS: the bias comes from global memory); both groups issue their pieces of
// half-stage k + 4 in epoch k, into the slot half-stage k - 1 left:
//   WAR: half-stage k - 1 is read by group 0 in epoch k - 2 and by group 1 in epoch k - 1; every wave retires its reads
//        (lgkmcnt(0)) before the barrier that closes the epoch it read in, so slot (k - 1) % 5 is free from barrier k on.
//   RAW: epoch k + 1 reads half-stages k + 1 (group 1) and k + 2 (group 0): before barrier k + 1 every wave waits
//        vmcnt(8) -- only its pieces of k + 3 and k + 4 may stay in flight -- so everything up to k + 2 has landed for
//        everybody.  A half-stage is issued four epochs before its first read: two full epochs of lead, as in the forms above.
// Tile boundary: in the last epoch of a tile group 0 only issues its pieces (the next tile's first fragments would have to
// live across the epilogue) and reads them after the epilogue; both groups leave the last barrier together, so the two
// epilogues run side by side without the extra barriers of the two-barrier forms.
template <class T, int ACT, bool HAS_BIAS, bool HAS_RES>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void linear_pp3_kernel(const PpArgs a) {
  constexpr int NS = 5, PP = 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NS * kSlot];
  using frag = typename T::frag;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;   // group = m half (rows grp*128), strip = columns wn*64
  const int wg = blockIdx.x;
  const int K = a.K, K2 = K * 2;
  const int nh = 2 * a.nk;                    // half-stages of a tile
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);

#ifdef CODETR_PP_STAMPS
  unsigned long long pp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pp_k0 = __builtin_readcyclecounter();
#endif
  int c_idx = 0;
  int c_tile = item_tile(a, wg, 0);
  if (c_tile < 0) return;   // nothing to do (uniform for the whole workgroup: no barrier has been executed)

  // ---- producer (as linear_pp_kernel): this wave's 2 + 2 pieces of a half-stage, NS - 1 half-stages ahead ----
  unsigned voffW[2], voffX[2];
  const unsigned char* Wp = a.W;
  const unsigned char* Xp = a.X;
  int p_idx = 0, p_tile = c_tile, p_h = 0;
  auto prod_set_tile = [&]() {
    const int tm = p_tile / a.tiles_n, tn = p_tile - tm * a.tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int nmax = a.N - 1 - n0, mmax = a.M - 1 - m0;
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
    Wp = a.W + (size_t)n0 * K * 2;
    Xp = a.X + (size_t)m0 * K * 2;
    p_h = 0;
  };
  prod_set_tile();
  auto produce = [&](int slot) {   // the 4 pieces of the producer's half-stage into `slot`, then on to the next one
    if (!(kAbl & 1)) {
      const unsigned dst = lds0 + (unsigned)slot * kSlot + (unsigned)wave * 1024u;
      lds_dma16(Wp, voffW[0], dst);
      lds_dma16(Wp, voffW[1], dst + 8192u);
      lds_dma16(Xp, voffX[0], dst + kOpBytes);
      lds_dma16(Xp, voffX[1], dst + kOpBytes + 8192u);
    }
    if (p_tile < 0) return;   // past the end: the same half-stage again (nobody reads it; the counted waits stay uniform)
    if (++p_h < nh) {
      Wp += 64;
      Xp += 64;
      return;
    }
    p_tile = item_tile(a, wg, ++p_idx);
    if (p_tile >= 0) prod_set_tile();
  };

  const int fa = lane & 15, fc = lane >> 4;
  const unsigned offA = (unsigned)((wn * 64 + fa) * 64 + ((fc ^ key64(fa)) * 16));
  const unsigned offB = (unsigned)(kOpBytes + (grp * 128 + fa) * 64 + ((fc ^ key64(fa)) * 16));

  f32x4 acc[4][8];   // [n-tile][m-tile]
  frag fw[4], fx[8];

  // ---- prologue: half-stages 0 .. 3 issued, 0 and 1 landed for everybody ----
#pragma unroll
  for (int s_ = 0; s_ < NS - 1; ++s_) produce(s_);
  wait_vm<2 * PP>();
  seg_barrier();
  int rs = 0;   // ring slot of the half-stage the next LOAD segment of this wave reads

  auto read_frags = [&]() {
    const unsigned char* rbase = lds + rs * kSlot;
    if (!(kAbl & 4)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const frag*>(rbase + offA + i * 1024);
#pragma unroll
      for (int j = 0; j < 8; ++j) fx[j] = *reinterpret_cast<const frag*>(rbase + offB + j * 1024);
    }
    rs = rs + 1 == NS ? 0 : rs + 1;
  };
  // LOAD segment of half-stage p: fragments of p (when `reads`), pieces of the epoch's new half-stage into slot `fs`
  auto load_seg = [&](bool reads, int fs) {
    PP_T(0);
    if (reads) read_frags();
    produce(fs);
    PP_T(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_T(2);
    PP_ACC(0, 0, 1); PP_ACC(2, 1, 2);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma_seg = [&](bool firstk) {
    PP_T(5);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (kAbl & 2) asm volatile("" ::"v"(fw[i]), "v"(fx[j]));
        else if (firstk) acc[i][j] = T::mfma(fx[j], fw[i], f32x4{0.f, 0.f, 0.f, 0.f});
        else acc[i][j] = T::mfma(fx[j], fw[i], acc[i][j]);
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    PP_T(6);
    PP_ACC(4, 5, 6);
  };
  auto close_epoch = [&]() {
    PP_T(3);
    wait_vm<2 * PP>();
    PP_T(4);
    seg_barrier();
    PP_T(7);
    PP_ACC(1, 3, 4); PP_ACC(3, 4, 7);
  };

  int es = NS - 1;   // slot the current epoch's new half-stage goes to: (k - 1) % NS
  while (c_tile >= 0) {
    if (grp == 0) {
      // group 0:  [reads of half-stage 0]  then per epoch  M(k)  L(k + 1)
      read_frags();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      mfma_seg(true);
      load_seg(nh > 1, es);
      close_epoch();
      es = es + 1 == NS ? 0 : es + 1;
      for (int k = 1; k < nh; ++k) {
        mfma_seg(false);
        load_seg(k + 1 < nh, es);
        close_epoch();
        es = es + 1 == NS ? 0 : es + 1;
      }
    } else {
      // group 1:  per epoch  L(k)  M(k)
      load_seg(true, es);
      mfma_seg(true);
      close_epoch();
      es = es + 1 == NS ? 0 : es + 1;
      for (int k = 1; k < nh; ++k) {
        load_seg(true, es);
        mfma_seg(false);
        close_epoch();
        es = es + 1 == NS ? 0 : es + 1;
      }
    }
    PP_T(8);
    pp_epilogue<T, ACT, HAS_BIAS, HAS_RES, false>(acc, a, c_tile, grp, wn, lane, nullptr);
    PP_T(9);
    PP_ACC(6, 8, 9);
    c_tile = item_tile(a, wg, ++c_idx);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the producer's redundant fetches past the end
  PP_STAMPS_OUT();
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

bool pp_supported(int64_t M, int64_t N, int64_t K) {
  return M > 0 && N > 0 && K >= 128 && K % 64 == 0 && N % 8 == 0 && M <= 0x7fffffffLL && N <= 16384 &&
         255 * K * 2 + 64 < 0x7fffffffLL;
}

template <class T, int ACT, int MODE>   // MODE 0: K-tile image; 1 / 2: half-stage image in one / two phases
int launch_pp_act(hipStream_t st, const PpArgs& a, bool has_bias, bool has_res) {
  const dim3 grid((unsigned)a.G), block(512);
#define CODETR_PP(HB, HR)                                                                                   \
  do {                                                                                                      \
    if (MODE == 0) hipLaunchKernelGGL((linear_pp2_kernel<T, ACT, HB, HR>), grid, block, 0, st, a);          \
    else if (MODE == 3) hipLaunchKernelGGL((linear_pp3_kernel<T, ACT, HB, HR>), grid, block, 0, st, a);     \
    else hipLaunchKernelGGL((linear_pp_kernel<T, ACT, HB, HR, 4, MODE>), grid, block, 0, st, a);            \
  } while (0)
  if (has_bias && has_res) CODETR_PP(true, true);
  else if (has_bias) CODETR_PP(true, false);
  else if (has_res) CODETR_PP(false, true);
  else CODETR_PP(false, false);
#undef CODETR_PP
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

template <class T>
int launch_pp(hipStream_t st, const void* X, const void* W, const void* bias, const void* R, void* Y, int64_t M, int64_t N,
              int64_t K, int act, int flags) {
  if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return CODETR_E_BADARG;
  if (act < 0 || act > 2 || !pp_supported(M, N, K)) return CODETR_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Y) |
       reinterpret_cast<uintptr_t>(R) | reinterpret_cast<uintptr_t>(bias)) & 15)
    return CODETR_E_BADARG;
  const int64_t tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256, T_ = tiles_m * tiles_n;
  if (T_ > 0x3fffffff) return CODETR_E_TOO_LARGE;
  int G = device_cus() / 8 * 8;
  if (G <= 0) return CODETR_E_BADARG;
  PpArgs a;
  a.X = static_cast<const unsigned char*>(X);
  a.W = static_cast<const unsigned char*>(W);
  a.bias = static_cast<const unsigned short*>(bias);
  a.R = static_cast<const unsigned short*>(R);
  a.Y = static_cast<unsigned short*>(Y);
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_n = (int)tiles_n; a.T = (int)T_; a.nk = (int)(K / 64);
  a.G = G;
  a.rounds = (int)(T_ / G);
  a.rem = (int)(T_ - (int64_t)a.rounds * G);
  a.alt = (flags & 4) ? 1 : 0;
  const bool hb = bias != nullptr, hr = R != nullptr;
  if ((int64_t)256 * K * 2 > 0x7fffffffLL) return CODETR_E_TOO_LARGE;   // 32-bit offsets inside a tile's rows
  const int mode = flags & 3;   // 0: K-tile image (default); 1 / 2: the half-stage image in one / two phases (A/B only)
#define CODETR_PP_ACT(ACT)                                                    \
  switch (mode) {                                                             \
    case 0: return launch_pp_act<T, ACT, 0>(st, a, hb, hr);                   \
    case 1: return launch_pp_act<T, ACT, 1>(st, a, hb, hr);                   \
    case 2: return launch_pp_act<T, ACT, 2>(st, a, hb, hr);                   \
    default: return launch_pp_act<T, ACT, 3>(st, a, hb, hr);                  \
  }
  switch (act) {
    case 0: CODETR_PP_ACT(0)
    case 1: CODETR_PP_ACT(1)
    default: CODETR_PP_ACT(2)
  }
#undef CODETR_PP_ACT
}

}  // namespace

extern "C" {

int codetr_linear_pp_supported(int64_t M, int64_t N, int64_t K) { return pp_supported(M, N, K) ? 1 : 0; }

// Where the ping-pong kernel measured faster than both older kernels (tools/micro/gemm_sk_bench on the 4- and 8-image Swin-L
// shapes, profiles/r06_gemm_pp.txt).  Until measured: nowhere.
int codetr_linear_pp_preferred(int64_t M, int64_t N, int64_t K, int act, int has_residual) {
  (void)act;
  (void)has_residual;
  (void)M; (void)N; (void)K;
  return 0;
}

int codetr_linear_pp_f16(void* stream, const void* x_dev, const void* w_dev, const void* bias_dev, const void* residual_dev,
                         void* y_dev, int64_t M, int64_t N, int64_t K, int act, int flags) {
  return launch_pp<HalfT>(static_cast<hipStream_t>(stream), x_dev, w_dev, bias_dev, residual_dev, y_dev, M, N, K, act, flags);
}

int codetr_linear_pp_bf16(void* stream, const void* x_dev, const void* w_dev, const void* bias_dev, const void* residual_dev,
                          void* y_dev, int64_t M, int64_t N, int64_t K, int act, int flags) {
#if CODETR_PP_ABL
  return CODETR_E_UNSUPPORTED;   // diagnostic builds carry the fp16 instantiations only
#endif
  return launch_pp<BFloatT>(static_cast<hipStream_t>(stream), x_dev, w_dev, bias_dev, residual_dev, y_dev, M, N, K, act, flags);
}

}  // extern "C"
