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
//
// What this buys and what bounds it (profiles/r06_gemm_pp.txt, in-kernel stamps of tools/micro/pp_stamps.hip): -4 ... -10 %
// against the faster of the two older kernels on every Swin stage 2-3 shape.  Per half-stage a wave spends 505-540 cycles in
// its MFMA segment (32 x 16 + the barrier's skew), 450-520 in its LOAD segment -- 240 for the 12 fragment reads (the four
// loading waves of an interval ask the LDS for 48 KiB: 192 cycles at 256 B/clk) and ~100 for EACH of the four DMA pieces (the
// texture path takes a 1-KiB piece from each of four waves at 64 B/clk) -- and 60-190 at each barrier: 1 375 cycles per
// half-stage against 1 024 of MFMA work for the two waves of a SIMD; with the MFMAs alone the same skeleton runs at 1 118.
// Built, measured and NOT kept (tools/micro/experiments/gemm_pp_variants.hip, gemm_pp_regstage.hip): register staging
// (global_load -> VGPR -> ds_write_b128: +15 ... +25 %, the ds_write_b128 stretch the partner's MFMA segment from 512 to
// 749 cycles), a K-tile image with whole-line DMA pieces (box-dependent: -7 ... +10 %), the half-stage in two phases (level),
// one barrier per half-stage with the groups in opposite segment order (level), DMA-first order in the odd strips (+3 ... +6 %).
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

template <class T, int ACT>
int launch_pp_act(hipStream_t st, const PpArgs& a, bool has_bias, bool has_res) {
  const dim3 grid((unsigned)a.G), block(512);
#define CODETR_PP(HB, HR) hipLaunchKernelGGL((linear_pp_kernel<T, ACT, HB, HR, 4, 1>), grid, block, 0, st, a)
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
  const bool hb = bias != nullptr, hr = R != nullptr;
  if ((int64_t)256 * K * 2 > 0x7fffffffLL) return CODETR_E_TOO_LARGE;   // 32-bit offsets inside a tile's rows
  if (flags != 0) return CODETR_E_UNSUPPORTED;   // (the measured variants live in tools/micro/experiments/gemm_pp_variants.hip)
  switch (act) {
    case 0: return launch_pp_act<T, 0>(st, a, hb, hr);
    case 1: return launch_pp_act<T, 1>(st, a, hb, hr);
    default: return launch_pp_act<T, 2>(st, a, hb, hr);
  }
}

}  // namespace

extern "C" {

int codetr_linear_pp_supported(int64_t M, int64_t N, int64_t K) { return pp_supported(M, N, K) ? 1 : 0; }

// Where the ping-pong kernel measured faster than both older kernels (tools/micro/gemm_sk_bench on the 4- and 8-image Swin-L
// shapes over three boxes, profiles/r06_gemm_pp.txt): K >= 768 on problems of at least 200 tiles that waste little of a
// 256-wide tile -- every linear of Swin stages 2 and 3 and stage 1's fc2 (-4 ... -10 %).  The K = 384 layers are level (stay on
// the persistent kernel), N = 192 (stage 0's fc2) is 3 % behind.
int codetr_linear_pp_preferred(int64_t M, int64_t N, int64_t K, int act, int has_residual) {
  (void)act;
  (void)has_residual;
  if (!pp_supported(M, N, K) || K < 768 || N < 384) return 0;
  const int64_t tn = (N + 255) / 256, tiles = ((M + 255) / 256) * tn;
  return tiles >= 200 && N * 8 >= tn * 256 * 7 ? 1 : 0;
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
