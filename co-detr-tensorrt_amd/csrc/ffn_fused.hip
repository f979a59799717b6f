// Fused transformer FFN for MI355X (gfx950):  Y = X + relu(X . W1^T + b1) . W2^T + b2
// (reference codetr/transformer_mmcv.py:484-500: Linear -> ReLU -> Linear + identity; encoder / decoder FFN of
// Co-DINO: C = 256, hidden = 2048 -- 37 % of the model's flops at 1920x1280).
//
// As two GEMMs the hidden activation [M, 2048] (838 MB in fp16 at M = 204 600) is written to HBM by the first and
// read back by the second, and each GEMM re-reads its activation tile from L2 once per 128-column output tile.  Here
// the hidden activation never leaves the CU:
//   * persistent 256-thread workgroups (one per CU) walk the 128-row tiles; each wave keeps its 32 rows of X as MFMA
//     B-fragments in registers (2 m-tiles x 8 k-steps) and its 32 x 256 slice of Y in accumulators;
//   * the hidden dimension is walked in chunks of 64: H^T[h][m] = W1c . X^T  (K = 256), bias folded into the
//     accumulator init, ReLU, fp16 pack -- and the packed accumulator IS the B operand of the second product
//     (cdna_hip_programming.md section 3 "accumulator tile as the next MFMA's operand": the k-slot permutation
//     8g+j <-> rows {4g..4g+3} of two 16-row tiles is baked into W2 once, by codetr_ffn_pack_w2_f16, so that the
//     matching A fragment is one ds_read_b128);
//     Y^T[n][m] += W2c . relu(H)^T  (K = 64);
//   * W1 / W2 chunks (32 KiB each) stream through two 2-stage LDS rings by LDS-DMA, XOR-swizzled on the source
//     address so that the ds_read_b128 fragment reads are conflict-free, and keep streaming across tiles.  One barrier
//     in front of each product, with a counted wait (vmcnt(8)) for pieces issued half a chunk earlier; with one wave
//     per SIMD nothing else hides latency or issue cost, so the schedule is spelled out: fragments are read one step
//     ahead of their MFMAs (behind the first two MFMAs of a group), one LDS-DMA piece of the next chunk per MFMA
//     group; the LDS-DMA is inline assembly (scalar base + per-thread offset: no vector arithmetic per piece, and the
//     compiler keeps emitting counted lgkmcnt waits); b1 / b2 / the LayerNorm parameters sit in LDS;
//   * the epilogue works out of the accumulators: lanes 16 apart swap halves so that every row access is 16 bytes per
//     lane, the identity comes from the operand registers of another lane group, LayerNorm statistics are 64 in-lane
//     values + two shuffles; no LDS staging, no barrier, no second read of X; the next tile's rows arrive meanwhile.
// Algorithmic HBM traffic: X once in, Y once out (2 x M x 256 x 2 B) + 2 MB of weights re-read from L2 per tile.
// History and the in-kernel timeline behind this shape: DESIGN.md section 4, profiles/r02_ffn_stamps.txt.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "codetr_hip.h"

// diagnostic builds only (tools/micro/ffn_ablate.hip -DCODETR_FFN_ABL=mask; WRONG results by construction, never shipped):
// 1 = no LDS-DMA inside the chunk loop, 2 = no MFMAs, 4 = no W fragment reads inside the chunk loop, 8 = no waits / barriers in the
// chunk loop, 16 = no ReLU / pack between the products (B operand of product 2 = stale registers); OPROJ form: 32 = no identity
// loads, 64 = no accumulator -> operand transform (operand = the attention rows), 128 = no product-0 MFMAs, 256 = no Wo staging
#ifdef CODETR_FFN_ABL
#define CODETR_FFN_ABL_MASK CODETR_FFN_ABL
#else
#define CODETR_FFN_ABL_MASK 0
#endif
namespace {

constexpr int C = 256;         // model width (K of the first product, N of the second)
constexpr int BH = 64;         // hidden units per chunk
constexpr int kThreads = 256;
constexpr int kW1Bytes = BH * C * 2;   // 32 KiB: [64 h][256 k]
constexpr int kW2Bytes = C * BH * 2;   // 32 KiB: [256 n][64 h]
constexpr int kStageBytes = kW1Bytes + kW2Bytes;
constexpr int kMaxHidden = 8192;       // b1 lives in LDS behind the two stages (16 KiB)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// element types: fp16 / bf16 storage, fp32 accumulation on the matrix cores either way
struct F16E {
  using e = _Float16;
  using v8 = f16x8;
  using v4 = f16x4;
  __device__ static f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  // v_pk_max_f16: one op per two values
  __device__ static v8 relu(v8 x) {
    const v8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_elementwise_max(x, z);
  }
};
struct BF16E {
  using e = __bf16;
  using v8 = bf16x8;
  using v4 = bf16x4;
  __device__ static f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  // sign-magnitude 16-bit floats order like int16 on the non-negative side: v_pk_max_i16(x, 0) is ReLU (-0 -> +0)
  __device__ static v8 relu(v8 x) {
    s16x8 i;
    __builtin_memcpy(&i, &x, 16);
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    i = __builtin_elementwise_max(i, z);
    __builtin_memcpy(&x, &i, 16);
    return x;
  }
};

// one LDS-DMA piece: 256 threads x 16 B = 4 KiB.  `src` is wave-uniform (kernel argument + scalar offsets), `voff` the
// thread's byte offset, `dst` the wave's uniform LDS destination.  Inline assembly, not __builtin_amdgcn_global_load_lds:
// the compiler's wait-count pass files the builtin with out-of-order LDS traffic and from then on turns every wait for a
// ds_read into lgkmcnt(0), which voids the fragment read-ahead; the instruction only counts in vmcnt, which this kernel
// waits on by hand (see ffn_fp8.hip).
__device__ __forceinline__ void dma16(const unsigned char* src, unsigned voff, unsigned char* dst) {
  const unsigned lds_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)dst);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds_addr) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Persistent: gridDim.x workgroups (one per CU) walk the 128-row tiles blockIdx.x, + gridDim.x, ...; the W ring keeps
// streaming across tiles (every tile reads the same W), the next tile's rows are requested at the start of the
// epilogue, and the epilogue works out of the accumulators (no LDS staging, no barrier) while the next tile's first W
// chunks arrive.
// MTT = 16-row tiles per wave: 2 (128 rows per workgroup, the product shape) or 1 (64 rows per workgroup: a tile takes
// half the MFMA time for the same weight stream -- used for a left-over partial round, see ffn_entry).
//
// OPROJ (the encoder layer's attention output projection folded in -- reference transformer_mmcv.py post-norm layer:
// x1 = norm1(identity + output_proj(attn)), y = norm2(x1 + ffn(x1))): X holds the ATTENTION OUTPUT rows, and a tile starts with
// a "product 0"  T^T[n][m] = Wo . A^T  (4 chunks of 64 output channels through the W1 ring: 256 MFMAs per wave on top of the
// tile's 4096), + bo -> E, + identity row (ID, read here) -> E, LayerNorm (ln_in) -> E: the first product's operand.  That
// operand is produced in the EPILOGUE's lane layout (a lane owns 8 consecutive channels 32 j + cbase ..), which is the MFMA
// B layout with lane groups 1 and 2 exchanged; the exchange is baked into W1's columns once on the host
// (codetr_ffn_oproj_w1_index), so no data moves between lanes and the epilogue's identity is the operand register itself.
// What it replaces: a GEMM launch that reads the attention output and the identity and writes x0 = identity +
// output_proj(attn) (1.26 GB at four 1920x1280 images), and this kernel's read of x0 (0.42 GB).
template <class ET, int MTT = 2, bool OPROJ = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_fused_kernel(
    const unsigned short* __restrict__ X, const unsigned short* __restrict__ W1, const unsigned short* __restrict__ b1,
    const unsigned short* __restrict__ W2, const unsigned short* __restrict__ b2, unsigned short* __restrict__ Y, int M,
    int Hd, const unsigned short* __restrict__ ln_g, const unsigned short* __restrict__ ln_b, float ln_eps,
    const unsigned short* __restrict__ pos, unsigned short* __restrict__ Y2,
    const unsigned short* __restrict__ lnin_g, const unsigned short* __restrict__ lnin_b, float lnin_eps, int ntiles,
    const unsigned short* __restrict__ Wo = nullptr, const unsigned short* __restrict__ bo = nullptr,
    const unsigned short* __restrict__ ID = nullptr) {
  // [W1 stage 0 | W1 stage 1 | W2 stage 0 | W2 stage 1 | b1 | LayerNorm gamma, beta | b2 | bo]: 146 KiB, one object
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kStageBytes + kMaxHidden * 2 + 2048];
  using E = typename ET::e;
  using V8 = typename ET::v8;
  using V4 = typename ET::v4;
  constexpr int MT = MTT, WR = 16 * MT, TR = 4 * WR;   // 16-row tiles / rows per wave / rows per workgroup
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, grp = lane >> 4;
  const int nchunks = Hd / BH;
  unsigned char* ringA = lds;                   // W1 chunks [64 h][256 k]
  unsigned char* ringB = lds + 2 * kW1Bytes;    // packed W2 chunks [256 n][64 h]
  unsigned short* sB1 = reinterpret_cast<unsigned short*>(lds + 2 * kStageBytes);
  unsigned short* sLn = reinterpret_cast<unsigned short*>(lds + 2 * kStageBytes + kMaxHidden * 2);  // gamma[256], beta[256]
  unsigned short* sB2 = sLn + 512;
  unsigned short* sBo = sB2 + 256;

  // LDS-DMA geometry (the images of the previous version of this kernel: tests/test_lds_bank_model.py).
  // W1 chunk: 64 rows x 32 chunks of 16 B; piece p (0..7) = rows 8 p + (tid >> 5), position tid & 31 of row r holds
  // source chunk (tid & 31) ^ (r & 15): the key depends on the piece's parity only.  W2 chunk: 256 rows x 8 chunks;
  // piece q = rows 32 q + (tid >> 3), position tid & 7 holds chunk (tid & 7) ^ ((row >> 1) & 7): piece-independent.
  unsigned w1_voff[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    w1_voff[par] = (unsigned)((tid >> 5) * (C * 2) + (((tid & 31) ^ ((par * 8 + (tid >> 5)) & 15)) * 16));
  const unsigned w2_voff = (unsigned)((tid >> 3) * Hd * 2 + (((tid & 7) ^ ((tid >> 4) & 7)) * 16));
  const unsigned char* W1b = reinterpret_cast<const unsigned char*>(W1);
  const unsigned char* W2b = reinterpret_cast<const unsigned char*>(W2);
  const unsigned char* Wob = reinterpret_cast<const unsigned char*>(Wo);
  auto stage_w1 = [&](int p, int c, unsigned char* dst) {
    dma16(W1b + (size_t)c * kW1Bytes + p * 4096, w1_voff[p & 1], dst + (p * kThreads + wave * 64) * 16);
  };
  auto stage_wo = [&](int p, int c, unsigned char* dst) {   // chunk c of Wo: rows 64 c .. + 63, the W1 chunk geometry
    dma16(Wob + (size_t)c * kW1Bytes + p * 4096, w1_voff[p & 1], dst + (p * kThreads + wave * 64) * 16);
  };
  auto stage_w2 = [&](int q, int c, unsigned char* dst) {
    dma16(W2b + (size_t)q * 64 * Hd + c * (BH * 2), w2_voff, dst + (q * kThreads + wave * 64) * 16);
  };
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    if (OPROJ) stage_wo(p, 0, ringA);
    else stage_w1(p, 0, ringA);
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) stage_w2(p, 0, ringB);
  if (OPROJ) {
#pragma unroll
    for (int p = 0; p < 8; ++p) stage_wo(p, 1, ringA + kW1Bytes);
  }

  // a tile's input rows as they come from memory (B-operand layout: lane (j = l15, g) holds X[m][32 ks + 8 g .. + 7])
  V8 xn[MT][8];
  auto load_x = [&](int tile, int mt) {
    int m = tile * TR + wave * WR + mt * 16 + l15;
    m = m < M ? m : M - 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) xn[mt][ks] = *reinterpret_cast<const V8*>(X + (size_t)m * C + ks * 32 + grp * 8);
  };
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) load_x(blockIdx.x, mt);
  // b1 and the output LayerNorm's parameters go to LDS once per workgroup (read by ds_read in the loop / epilogue: no
  // vector-memory op in the main loop but the LDS-DMA pieces)
  for (int i = tid; i < Hd / 8; i += kThreads)
    *reinterpret_cast<s16x8*>(sB1 + i * 8) = *reinterpret_cast<const s16x8*>(b1 + i * 8);
  if (ln_g && tid < 64) {
    const int i = tid & 31;
    *reinterpret_cast<s16x8*>(sLn + (tid >> 5) * 256 + i * 8) = *reinterpret_cast<const s16x8*>((tid >> 5 ? ln_b : ln_g) + i * 8);
  }
  if (tid >= 64 && tid < 96) *reinterpret_cast<s16x8*>(sB2 + (tid - 64) * 8) = *reinterpret_cast<const s16x8*>(b2 + (tid - 64) * 8);
  if (OPROJ && tid >= 96 && tid < 128) *reinterpret_cast<s16x8*>(sBo + (tid - 96) * 8) = *reinterpret_cast<const s16x8*>(bo + (tid - 96) * 8);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // tables written; the first barrier below publishes them

  int gc = 0;  // chunks consumed so far: ring stage = gc & 1
  int ga = 0;  // W1-ring stages consumed so far (OPROJ: three of Wo's four chunks per tile pass through it too)
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * TR + wave * WR;
    // ---- the MFMA operand of the first product and the identity: the rows, or their LayerNorm (the post-norm layer's
    // first norm, whose output nothing else reads).  The 256 values of row (mt, l15) sit in the four lanes l15 + 16 g
    // (8 k-steps x 8 values each): statistics are two xor-shuffles away; fp32 two-pass like layernorm_kernel, result
    // rounded to E = the tensor the separate kernel would have written.
    V8 xf[MT][8];
    f32x4 yacc[16][MT];
    if constexpr (OPROJ) {
      // ---- product 0 (see the kernel's header): x1 = LayerNorm(identity + E(Wo . attn + bo)), in the epilogue's lane layout
      const int cb = 16 * (grp & 1) + 8 * (grp >> 1);
      // identity rows, in the epilogue's lane layout: requested now, consumed behind the 256 MFMAs below (requested a tile
      // ahead, with the attention rows, they measured the same and cost the bf16 form 76 bytes of scratch)
      V8 idn[MT][8];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        int m = m0 + mt * 16 + l15;
        m = m < M ? m : M - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (CODETR_FFN_ABL_MASK & 32) idn[mt][j] = xn[mt][j];
          else idn[mt][j] = *reinterpret_cast<const V8*>(ID + (size_t)m * C + 32 * j + cb);
        }
      }
      // the tile's attention rows were requested after the LDS-DMA pieces of Wo[0] / W2[0]: once they have landed, so have
      // those (vmcnt is in order) -- the first chunk takes no counted wait, as chunk 0 of the plain kernel
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(xn[mt][7]) : "memory");
#pragma unroll
      for (int nt = 0; nt < 16; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) yacc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      // Wo's chunks run two products ahead of their use, through three buffers -- A0 / A1 = the W1 ring's stages and the stage
      // of the W2 ring that is free until chunk 0's second product (B1):
      //   Wo[0] -> A0 during the previous tile's last first product (initially: before the loop), Wo[1] -> A1 at the start
      //   of its epilogue (in front of this tile's row loads), Wo[2] -> B1 during chunk 0 here, Wo[3] -> A0 during chunk 1, W1[0] -> A1 during chunk 2.
      // Wo[0] / Wo[1] / W2[0] are older than the tile's rows (waited for above); Wo[2] and Wo[3] take a counted wait (the 8
      // pieces issued during the chunk before are the younger ones).
      unsigned char* const A0 = ringA + (ga & 1) * kW1Bytes;
      unsigned char* const A1 = ringA + ((ga + 1) & 1) * kW1Bytes;
      unsigned char* const B1 = ringB + ((gc + 1) & 1) * kW2Bytes;
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const unsigned char* sW = pc == 0 ? A0 : pc == 1 ? A1 : pc == 2 ? B1 : A0;
        if (pc >= 2 && !(CODETR_FFN_ABL_MASK & 256)) wait_vmcnt<8>();
        __builtin_amdgcn_s_barrier();
        auto read_wo = [&](int ks, V8 (&a)[4]) {
#pragma unroll
          for (int ht = 0; ht < 4; ++ht) {
            const int row = ht * 16 + l15;
            const int chunk = (ks * 4 + grp) ^ (row & 15);
            a[ht] = *reinterpret_cast<const V8*>(sW + row * (C * 2) + chunk * 16);
          }
        };
        V8 aw[2][4];
        read_wo(0, aw[0]);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + 1 < 8) read_wo(ks + 1, aw[(ks + 1) & 1]);
          if (pc == 0 && !(CODETR_FFN_ABL_MASK & 256)) stage_wo(ks, 2, B1);
          else if (pc == 1 && !(CODETR_FFN_ABL_MASK & 256)) stage_wo(ks, 3, A0);
          else if (pc == 2) stage_w1(ks, 0, A1);   // the tile's first W1 chunk: read from A1 by chunk 0 (ga + 3)
#pragma unroll
          for (int i = 0; i < 4 * MT; ++i)
            if (!(CODETR_FFN_ABL_MASK & 128))
              yacc[4 * pc + i / MT][i % MT] = ET::mfma(aw[ks & 1][i / MT], xn[i % MT][ks], yacc[4 * pc + i / MT][i % MT]);
            else asm volatile("" ::"v"(aw[ks & 1][i / MT]));
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT - 2, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      ga += 3;
      // accumulators -> E (+ bo), lanes 16 apart swap halves (as the epilogue), + identity -> E, LayerNorm -> E
      const int oddp = grp & 1;
      if (CODETR_FFN_ABL_MASK & 64) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            xf[mt][j] = xn[mt][j];
            asm volatile("" ::"v"(idn[mt][j]), "v"(yacc[2 * j][mt]), "v"(yacc[2 * j + 1][mt]));
          }
      } else
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        unsigned yp[16][2];
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
          const V4 bb = *reinterpret_cast<const V4*>(sBo + nt * 16 + grp * 4);
          const V4 y = {(E)(yacc[nt][mt][0] + (float)bb[0]), (E)(yacc[nt][mt][1] + (float)bb[1]),
                        (E)(yacc[nt][mt][2] + (float)bb[2]), (E)(yacc[nt][mt][3] + (float)bb[3])};
          __builtin_memcpy(yp[nt], &y, 8);
        }
        float o[8][8];
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned s0 = oddp ? yp[2 * j][0] : yp[2 * j + 1][0], s1 = oddp ? yp[2 * j][1] : yp[2 * j + 1][1];
          const unsigned k0 = oddp ? yp[2 * j + 1][0] : yp[2 * j][0], k1 = oddp ? yp[2 * j + 1][1] : yp[2 * j][1];
          const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16, 64), r1 = (unsigned)__shfl_xor((int)s1, 16, 64);
          const unsigned z[4] = {oddp ? r0 : k0, oddp ? r1 : k1, oddp ? k0 : r0, oddp ? k1 : r1};   // channels 32 j + cb .. + 7
          V8 yv;
          __builtin_memcpy(&yv, z, 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o[j][e] = (float)(E)((float)yv[e] + (float)idn[mt][j][e]);   // identity + output_proj(attn): E + E -> E
            sm += o[j][e];
          }
        }
        if (lnin_g) {
          sm += __shfl_xor(sm, 16, 64);
          sm += __shfl_xor(sm, 32, 64);
          const float mean = sm * (1.0f / C);
          float q = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float d = o[j][e] - mean;
              q = fmaf(d, d, q);
            }
          q += __shfl_xor(q, 16, 64);
          q += __shfl_xor(q, 32, 64);
          const float rstd = rsqrtf(q * (1.0f / C) + lnin_eps);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const V8 gw = *reinterpret_cast<const V8*>(lnin_g + 32 * j + cb);
            const V8 gb = *reinterpret_cast<const V8*>(lnin_b + 32 * j + cb);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[j][e] = fmaf((o[j][e] - mean) * rstd, (float)gw[e], (float)gb[e]);
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) xf[mt][j][e] = (E)o[j][e];
      }
    } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (lnin_g) {
        float sm = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) sm += (float)xn[mt][ks][e];
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        const float mean = sm * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = (float)xn[mt][ks][e] - mean;
            q = fmaf(d, d, q);
          }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.0f / C) + lnin_eps);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const V8 gw = *reinterpret_cast<const V8*>(lnin_g + ks * 32 + grp * 8);
          const V8 gb = *reinterpret_cast<const V8*>(lnin_b + ks * 32 + grp * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            xf[mt][ks][e] = (E)fmaf(((float)xn[mt][ks][e] - mean) * rstd, (float)gw[e], (float)gb[e]);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xf[mt][ks] = xn[mt][ks];
      }
      // Chunk 0 of this tile takes no counted wait (see the schedule below): it relies on the tile's input rows --
      // issued AFTER the LDS-DMA pieces of W1[0] / W2[0] -- having landed.  Without the LayerNorm nothing above
      // consumes them, so make the dependence explicit: the compiler must wait for the last-issued row register here
      // (vmcnt is in order, so the older pieces have landed too), whatever it does with the copies.
      asm volatile("" ::"v"(xn[mt][7]) : "memory");
    }
    }
    // (b2 is added in the epilogue, from LDS: as the accumulators' initial value it is loop-invariant across tiles and
    // the compiler keeps all 64 converted values alive through the whole loop -- spills)
#pragma unroll
    for (int nt = 0; nt < 16; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) yacc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Schedule of one chunk c (ring stage gc & 1; vmcnt counts LDS-DMA pieces, loads and stores in issue order):
    //   T: wait until W1[c] landed (the 8 younger pieces are W2[c]'s), barrier.  Not in a tile's chunk 0: its pieces
    //     are older than the tile's input rows, which the code above has waited for -- and a counted wait there would
    //     also wait for the previous tile's output stores
    //   product 1, one LDS-DMA piece of W1[c+1] per k-step
    //   M: wait until W2[c] landed (the 8 younger pieces are W1[c+1]'s), barrier
    //   product 2, one piece of W2[c+1] per pair of output tiles
    // W1[c+1] goes to the stage product 1 of chunk c-1 read (every wave passed M of c-1), W2[c+1] to the one product 2
    // of chunk c-1 read (every wave passed T of c).  Chunk nchunks wraps to chunk 0 of the next tile (past the last
    // tile: a fetch nobody reads, drained at the end).
    for (int c = 0; c < nchunks; ++c, ++gc, ++ga) {
      const int cn = c + 1 < nchunks ? c + 1 : 0;
      const unsigned char* sW1 = ringA + (ga & 1) * kW1Bytes;
      const unsigned char* sW2 = ringB + (gc & 1) * kW2Bytes;
      unsigned char* nW1 = ringA + ((ga + 1) & 1) * kW1Bytes;
      unsigned char* nW2 = ringB + ((gc + 1) & 1) * kW2Bytes;
      const unsigned char* nsrc1 = (OPROJ && c + 1 == nchunks) ? Wob : W1b + (size_t)cn * kW1Bytes;
      if (!(CODETR_FFN_ABL_MASK & 8)) {
        if (c > 0) wait_vmcnt<8>();
        else if (OPROJ) wait_vmcnt<0>();   // W1[0] was staged during product 0's third chunk
        __builtin_amdgcn_s_barrier();  // T
      }

      // ---- H^T = W1c . X^T : D[i = h][j = m] ----
      f32x4 hacc[4][MT];
#pragma unroll
      for (int ht = 0; ht < 4; ++ht) {
        // bias of this lane's 4 consecutive hidden units of tile ht
        const V4 bv = *reinterpret_cast<const V4*>(sB1 + c * BH + ht * 16 + grp * 4);
        const f32x4 b4 = {(float)bv[0], (float)bv[1], (float)bv[2], (float)bv[3]};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hacc[ht][mt] = b4;
      }
      // one wave per SIMD: nobody else hides LDS latency, so the A fragments of k-step ks+1 are read while the
      // MFMAs of k-step ks issue (explicit register double buffering)
      auto read_w1 = [&](int ks, V8 (&a)[4]) {
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
          const int row = ht * 16 + l15;
          const int chunk = (ks * 4 + grp) ^ (row & 15);
          a[ht] = *reinterpret_cast<const V8*>(sW1 + row * (C * 2) + chunk * 16);
        }
      };
      V8 aw[2][4];
      read_w1(0, aw[0]);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8 && !(CODETR_FFN_ABL_MASK & 4)) read_w1(ks + 1, aw[(ks + 1) & 1]);
        // (OPROJ: the next tile starts with product 0 -- the source is a scalar select, no branch inside the schedule)
        if (!(CODETR_FFN_ABL_MASK & 1)) dma16(nsrc1 + ks * 4096, w1_voff[ks & 1], nW1 + (ks * kThreads + wave * 64) * 16);
#pragma unroll
        for (int i = 0; i < 4 * MT; ++i)
          if (!(CODETR_FFN_ABL_MASK & 2))
            hacc[i / MT][i % MT] = ET::mfma(aw[(CODETR_FFN_ABL_MASK & 4) ? 0 : (ks & 1)][i / MT], xf[i % MT][ks], hacc[i / MT][i % MT]);
          else if (!(CODETR_FFN_ABL_MASK & 4)) asm volatile("" ::"v"(aw[ks & 1][i / MT]));
        // 2 MFMAs, then the 4 reads of the next step and the DMA piece, then the other 6
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT - 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- ReLU + pack: B operand of the second product, k-slot 8g+j = rows 4g..4g+3 of tiles 2s and 2s+1 ----
      V8 pf[2][MT];
      if (CODETR_FFN_ABL_MASK & 16) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pf[s][mt] = xf[mt][s];
#pragma unroll
        for (int ht = 0; ht < 4; ++ht)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(hacc[ht][mt]));
      } else
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) pf[s][mt][h * 4 + r] = (E)hacc[2 * s + h][mt][r];
      // ReLU on the packed halves (one op per two values).  max(NaN, 0) = 0 drops a NaN of the hidden unit, but a NaN
      // there can only come from a NaN / inf in this row of X, which the residual add puts back.
      if (!(CODETR_FFN_ABL_MASK & 16))
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) pf[s][mt] = ET::relu(pf[s][mt]);
      if (!(CODETR_FFN_ABL_MASK & 8)) {
        if (c > 0) wait_vmcnt<8>();
        __builtin_amdgcn_s_barrier();  // M
      }
      // ---- Y^T += W2c . relu(H)^T : D[i = n][j = m], k = hidden unit (permuted identically on both operands) ----
      // W2 fragments (pre-packed: the 8 k-slots of lane group g are 16 contiguous bytes), read two n-tiles ahead
      auto read_w2 = [&](int ntp, V8 (&a)[4]) {  // n-tiles 2*ntp, 2*ntp+1; index [t*2 + s]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int n = (2 * ntp + t) * 16 + l15;
          const unsigned char* rowp = sW2 + n * (BH * 2);
          const int sw = (n >> 1) & 7;
#pragma unroll
          for (int s = 0; s < 2; ++s) a[t * 2 + s] = *reinterpret_cast<const V8*>(rowp + ((4 * s + grp) ^ sw) * 16);
        }
      };
      V8 a2[2][4];
      read_w2(0, a2[0]);
#pragma unroll
      for (int ntp = 0; ntp < 8; ++ntp) {
        if (ntp + 1 < 8 && !(CODETR_FFN_ABL_MASK & 4)) read_w2(ntp + 1, a2[(ntp + 1) & 1]);
        if (!(CODETR_FFN_ABL_MASK & 1)) stage_w2(ntp, cn, nW2);
#pragma unroll
        for (int i = 0; i < 4 * MT; ++i) {   // MFMA index i -> (t, s, mt)
          const int ts = i / MT, mt = i % MT, nt = 2 * ntp + ts / 2;
          if (!(CODETR_FFN_ABL_MASK & 2))
            yacc[nt][mt] = ET::mfma(a2[(CODETR_FFN_ABL_MASK & 4) ? 0 : (ntp & 1)][ts], pf[ts & 1][mt], yacc[nt][mt]);
          else if (!(CODETR_FFN_ABL_MASK & 4)) asm volatile("" ::"v"(a2[ntp & 1][ts]));
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT - 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue out of the accumulators (no LDS staging): lane (column l15 = row m of the tile, group g) holds
    // Y^T[n = 16 nt + 4 g + r][m].  y -> E there; then lanes g and g ^ 1 (16 lanes apart) swap halves so that every lane
    // owns 8 CONSECUTIVE channels of 8 of the 16 tiles (g even: the even tiles, g odd: the odd ones; g >> 1 picks
    // channels 0-7 or 8-15): channels 32 j + cbase .. + 7, j < 8 -- every load and store is 16 bytes per lane and a
    // wave-instruction touches 64 contiguous bytes per row.  The identity X[m][32 j + 8 v ..] (v = cbase / 8) is the
    // register xf[mt][j] of lane group v: one lane permutation (groups 1 and 2 swap) instead of a second read of X.
    // Then + identity -> E, LayerNorm over the row (64 in-lane values + the four lanes of a row), + pos.
    // The next tile's input rows are requested as the epilogue goes (each half once the registers of the half just
    // written are free) and arrive while it runs.
    const int next_tile = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;
    if (OPROJ) {
      // Wo[1] -> the W1 stage the last chunk's first product read (every wave is past that chunk's barrier M; nobody reads
      // the W1 ring in a second product).  Issued BEFORE the next tile's rows, so "rows landed" still implies "Wo[1] landed".
#pragma unroll
      for (int p = 0; p < 8; ++p) stage_wo(p, 1, ringA + ((ga + 1) & 1) * kW1Bytes);
    }
    const int odd = grp & 1;
    const int cbase = 16 * odd + 8 * (grp >> 1);
    const int src_lane4 = (l15 + 16 * (2 * odd + (grp >> 1))) * 4;   // byte address of the lane that holds the identity
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + l15;
      if (MT == 2 && mt == 0) load_x(next_tile, 0);
      unsigned yp[16][2];
#pragma unroll
      for (int nt = 0; nt < 16; ++nt) {
        const V4 bb = *reinterpret_cast<const V4*>(sB2 + nt * 16 + grp * 4);
        const V4 y = {(E)(yacc[nt][mt][0] + (float)bb[0]), (E)(yacc[nt][mt][1] + (float)bb[1]),
                      (E)(yacc[nt][mt][2] + (float)bb[2]), (E)(yacc[nt][mt][3] + (float)bb[3])};
        __builtin_memcpy(yp[nt], &y, 8);
      }
      float o[8][8];
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // the partner takes this lane's half of the tile the partner owns, and gives its half of this lane's tile
        const unsigned s0 = odd ? yp[2 * j][0] : yp[2 * j + 1][0], s1 = odd ? yp[2 * j][1] : yp[2 * j + 1][1];
        const unsigned k0 = odd ? yp[2 * j + 1][0] : yp[2 * j][0], k1 = odd ? yp[2 * j + 1][1] : yp[2 * j][1];
        const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16, 64), r1 = (unsigned)__shfl_xor((int)s1, 16, 64);
        const unsigned z[4] = {odd ? r0 : k0, odd ? r1 : k1, odd ? k0 : r0, odd ? k1 : r1};   // channels cbase .. + 7
        V8 yv, xid;
        __builtin_memcpy(&yv, z, 16);
        int xw[4];
        __builtin_memcpy(xw, &xf[mt][j], 16);
        if (!OPROJ)   // (OPROJ: the operand already has the epilogue's lane layout)
#pragma unroll
        for (int d = 0; d < 4; ++d) xw[d] = __builtin_amdgcn_ds_bpermute(src_lane4, xw[d]);
        __builtin_memcpy(&xid, xw, 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[j][e] = (float)(E)((float)yv[e] + (float)xid[e]);   // identity + ffn(x): E + E -> E
          sm += o[j][e];
        }
      }
      V8 pr[8];
      if (Y2) {   // (requested here: the statistics and the normalisation below cover part of the latency)
        const unsigned short* prow = pos + (size_t)(m < M ? m : M - 1) * C + cbase;
#pragma unroll
        for (int j = 0; j < 8; ++j) pr[j] = *reinterpret_cast<const V8*>(prow + 32 * j);
      }
      if (mt == 0) load_x(next_tile, MT - 1);   // xf[0] / yacc[.][0] are dead from here on (MT == 1: the tile's only rows)
      if (ln_g) {
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        const float mean = sm * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = o[j][e] - mean;
            q = fmaf(d, d, q);
          }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.0f / C) + ln_eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const V8 gw = *reinterpret_cast<const V8*>(sLn + 32 * j + cbase);
          const V8 gb = *reinterpret_cast<const V8*>(sLn + 256 + 32 * j + cbase);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[j][e] = (float)(E)fmaf((o[j][e] - mean) * rstd, (float)gw[e], (float)gb[e]);
        }
      }
      if (m < M) {
        unsigned short* yrow = Y + (size_t)m * C + cbase;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          V8 ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = (E)o[j][e];
          *reinterpret_cast<V8*>(yrow + 32 * j) = ov;
        }
        if (Y2) {  // the next layer's attention input: this row + its positional encoding (E + E -> E)
          unsigned short* y2row = Y2 + (size_t)m * C + cbase;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            V8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (E)(o[j][e] + (float)pr[j][e]);
            *reinterpret_cast<V8*>(y2row + 32 * j) = ov;
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// W2 [256, hidden] -> same shape with the columns of every 64-block reordered to MFMA k-slot order:
// new column 32s + 8g + j  <-  old column 32s + 4g + j (j < 4)  |  32s + 16 + 4g + (j - 4) (j >= 4)
__global__ void pack_w2_kernel(const unsigned short* __restrict__ w2, unsigned short* __restrict__ out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i & 63);
  const int s = col >> 5, g = (col >> 3) & 3, j = col & 7;
  const int old = 32 * s + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
  out[i] = w2[(i & ~(int64_t)63) + old];
}

}  // namespace

namespace {

template <class ET>
int ffn_entry(void* stream, const void* x_dev, const void* w1_dev, const void* b1_dev,
                            const void* w2_packed_dev, const void* b2_dev, void* y_dev, int64_t M, int64_t C_in,
                            int64_t hidden, const void* ln_in_gamma_dev, const void* ln_in_beta_dev, float ln_in_eps,
                            const void* ln_gamma_dev, const void* ln_beta_dev, float ln_eps, const void* pos_dev,
                            void* y_plus_pos_dev, const void* wo_dev = nullptr, const void* bo_dev = nullptr,
                            const void* identity_dev = nullptr) {
  const void* w2_dev = w2_packed_dev;
  if (!x_dev || !w1_dev || !b1_dev || !w2_dev || !b2_dev || !y_dev || M <= 0 || hidden <= 0) return CODETR_E_BADARG;
  const bool oproj = wo_dev != nullptr;
  if (oproj && (!bo_dev || !identity_dev)) return CODETR_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(wo_dev) | reinterpret_cast<uintptr_t>(bo_dev) | reinterpret_cast<uintptr_t>(identity_dev)) & 15)
    return CODETR_E_BADARG;
  if ((ln_gamma_dev == nullptr) != (ln_beta_dev == nullptr) || (pos_dev == nullptr) != (y_plus_pos_dev == nullptr) ||
      (ln_in_gamma_dev == nullptr) != (ln_in_beta_dev == nullptr))
    return CODETR_E_BADARG;
  if (C_in != C || hidden % BH != 0 || hidden > kMaxHidden) return CODETR_E_UNSUPPORTED;
  if (M > 0x7fffffffLL - 256 || hidden > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  // every row / weight / parameter access is a 16-byte one
  if ((reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(w1_dev) | reinterpret_cast<uintptr_t>(b1_dev) |
       reinterpret_cast<uintptr_t>(w2_dev) | reinterpret_cast<uintptr_t>(b2_dev) | reinterpret_cast<uintptr_t>(y_dev) |
       reinterpret_cast<uintptr_t>(ln_in_gamma_dev) | reinterpret_cast<uintptr_t>(ln_in_beta_dev) |
       reinterpret_cast<uintptr_t>(ln_gamma_dev) | reinterpret_cast<uintptr_t>(ln_beta_dev) |
       reinterpret_cast<uintptr_t>(pos_dev) | reinterpret_cast<uintptr_t>(y_plus_pos_dev)) & 15)
    return CODETR_E_BADARG;
  // 128 rows per tile (2 x 16 rows per wave; 3 x 16 does not fit the register file with the interleaved DMA issue).
  // Persistent grid: one workgroup per CU.  A left-over partial round that fills at most half of the CUs (1 599 tiles on
  // 256 CUs at one 1920x1280 image: 63 tiles in a 7th round) is served by a second launch with 64-row tiles instead:
  // twice the workgroups, half the MFMA time per tile -- half a round instead of a whole one.
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const int64_t ntiles_all = (M + 127) / 128;
  const int64_t left = ntiles_all % cus;
  const bool split = left > 0 && 2 * left <= cus;
  const int64_t M1 = split ? (ntiles_all - left) * 128 : M;   // rows of the 128-row launch
  auto launch = [&](auto mt_tag, int64_t row0, int64_t rows) {
    constexpr int MTT = decltype(mt_tag)::value;
    const int ntiles = (int)((rows + 64 * MTT - 1) / (64 * MTT));
    const unsigned blocks = (unsigned)(ntiles < cus ? ntiles : cus);
    const size_t off = (size_t)row0 * C;
    auto at = [&](const void* p) { return p ? static_cast<const unsigned short*>(p) + off : nullptr; };
#define CODETR_FFN_ARGS \
    at(x_dev), static_cast<const unsigned short*>(w1_dev), static_cast<const unsigned short*>(b1_dev), \
        static_cast<const unsigned short*>(w2_dev), static_cast<const unsigned short*>(b2_dev), \
        const_cast<unsigned short*>(at(y_dev)), (int)rows, (int)hidden, static_cast<const unsigned short*>(ln_gamma_dev), \
        static_cast<const unsigned short*>(ln_beta_dev), ln_eps, at(pos_dev), const_cast<unsigned short*>(at(y_plus_pos_dev)), \
        static_cast<const unsigned short*>(ln_in_gamma_dev), static_cast<const unsigned short*>(ln_in_beta_dev), ln_in_eps, ntiles
    if (oproj)
      hipLaunchKernelGGL((ffn_fused_kernel<ET, MTT, true>), dim3(blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                         CODETR_FFN_ARGS, static_cast<const unsigned short*>(wo_dev),
                         static_cast<const unsigned short*>(bo_dev), at(identity_dev));
    else
      hipLaunchKernelGGL((ffn_fused_kernel<ET, MTT, false>), dim3(blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                         CODETR_FFN_ARGS);
#undef CODETR_FFN_ARGS
  };
  if (M1 > 0) launch(std::integral_constant<int, 2>{}, 0, M1);
  if (split) launch(std::integral_constant<int, 1>{}, M1, M - M1);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // namespace

extern "C" {

int codetr_ffn_pack_w2_f16(void* stream, const void* w2_dev, void* w2_packed_dev, int64_t C_out, int64_t hidden) {
  if (!w2_dev || !w2_packed_dev || C_out <= 0 || hidden <= 0) return CODETR_E_BADARG;
  if (hidden % BH != 0) return CODETR_E_UNSUPPORTED;
  const int64_t total = C_out * hidden;
  hipLaunchKernelGGL(pack_w2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(w2_dev), static_cast<unsigned short*>(w2_packed_dev), total);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

int codetr_ffn_relu_ln2_f16(void* stream, const void* x_dev, const void* w1_dev, const void* b1_dev,
                            const void* w2_packed_dev, const void* b2_dev, void* y_dev, int64_t M, int64_t C_in,
                            int64_t hidden, const void* ln_in_gamma_dev, const void* ln_in_beta_dev, float ln_in_eps,
                            const void* ln_gamma_dev, const void* ln_beta_dev, float ln_eps, const void* pos_dev,
                            void* y_plus_pos_dev) {
  return ffn_entry<F16E>(stream, x_dev, w1_dev, b1_dev, w2_packed_dev, b2_dev, y_dev, M, C_in, hidden, ln_in_gamma_dev,
                         ln_in_beta_dev, ln_in_eps, ln_gamma_dev, ln_beta_dev, ln_eps, pos_dev, y_plus_pos_dev);
}

int codetr_ffn_relu_ln2_bf16(void* stream, const void* x_dev, const void* w1_dev, const void* b1_dev,
                            const void* w2_packed_dev, const void* b2_dev, void* y_dev, int64_t M, int64_t C_in,
                            int64_t hidden, const void* ln_in_gamma_dev, const void* ln_in_beta_dev, float ln_in_eps,
                            const void* ln_gamma_dev, const void* ln_beta_dev, float ln_eps, const void* pos_dev,
                            void* y_plus_pos_dev) {
  return ffn_entry<BF16E>(stream, x_dev, w1_dev, b1_dev, w2_packed_dev, b2_dev, y_dev, M, C_in, hidden, ln_in_gamma_dev,
                         ln_in_beta_dev, ln_in_eps, ln_gamma_dev, ln_beta_dev, ln_eps, pos_dev, y_plus_pos_dev);
}

int codetr_ffn_oproj_relu_ln2_f16(void* stream, const void* attn_dev, const void* wo_dev, const void* bo_dev,
                                  const void* identity_dev, const void* w1_perm_dev, const void* b1_dev,
                                  const void* w2_packed_dev, const void* b2_dev, void* y_dev, int64_t M, int64_t C_in,
                                  int64_t hidden, const void* ln_in_gamma_dev, const void* ln_in_beta_dev, float ln_in_eps,
                                  const void* ln_gamma_dev, const void* ln_beta_dev, float ln_eps, const void* pos_dev,
                                  void* y_plus_pos_dev) {
  if (!wo_dev) return CODETR_E_BADARG;
  return ffn_entry<F16E>(stream, attn_dev, w1_perm_dev, b1_dev, w2_packed_dev, b2_dev, y_dev, M, C_in, hidden, ln_in_gamma_dev,
                         ln_in_beta_dev, ln_in_eps, ln_gamma_dev, ln_beta_dev, ln_eps, pos_dev, y_plus_pos_dev, wo_dev, bo_dev,
                         identity_dev);
}

int codetr_ffn_oproj_relu_ln2_bf16(void* stream, const void* attn_dev, const void* wo_dev, const void* bo_dev,
                                   const void* identity_dev, const void* w1_perm_dev, const void* b1_dev,
                                   const void* w2_packed_dev, const void* b2_dev, void* y_dev, int64_t M, int64_t C_in,
                                   int64_t hidden, const void* ln_in_gamma_dev, const void* ln_in_beta_dev, float ln_in_eps,
                                   const void* ln_gamma_dev, const void* ln_beta_dev, float ln_eps, const void* pos_dev,
                                   void* y_plus_pos_dev) {
  if (!wo_dev) return CODETR_E_BADARG;
  return ffn_entry<BF16E>(stream, attn_dev, w1_perm_dev, b1_dev, w2_packed_dev, b2_dev, y_dev, M, C_in, hidden,
                          ln_in_gamma_dev, ln_in_beta_dev, ln_in_eps, ln_gamma_dev, ln_beta_dev, ln_eps, pos_dev,
                          y_plus_pos_dev, wo_dev, bo_dev, identity_dev);
}

/* column j of the permuted W1 takes column idx[j] of nn.Linear's weight: the epilogue's lane layout (a lane owns channels
 * 32 s + 16 (g & 1) + 8 (g >> 1) .. + 7 of a row) as MFMA k-slot order (32 s + 8 g ..) */
int codetr_ffn_oproj_w1_index(int64_t C_in, int32_t* idx_host) {
  if (!idx_host || C_in != C) return CODETR_E_BADARG;
  for (int s = 0; s < C / 32; ++s)
    for (int g = 0; g < 4; ++g)
      for (int e = 0; e < 8; ++e) idx_host[32 * s + 8 * g + e] = 32 * s + 16 * (g & 1) + 8 * (g >> 1) + e;
  return 0;
}

int codetr_ffn_relu_ln_f16(void* stream, const void* x_dev, const void* w1_dev, const void* b1_dev,
                           const void* w2_packed_dev, const void* b2_dev, void* y_dev, int64_t M, int64_t C_in,
                           int64_t hidden, const void* ln_gamma_dev, const void* ln_beta_dev, float ln_eps,
                           const void* pos_dev, void* y_plus_pos_dev) {
  return codetr_ffn_relu_ln2_f16(stream, x_dev, w1_dev, b1_dev, w2_packed_dev, b2_dev, y_dev, M, C_in, hidden, nullptr,
                                 nullptr, 0.f, ln_gamma_dev, ln_beta_dev, ln_eps, pos_dev, y_plus_pos_dev);
}

int codetr_ffn_relu_f16(void* stream, const void* x_dev, const void* w1_dev, const void* b1_dev,
                        const void* w2_packed_dev, const void* b2_dev, void* y_dev, int64_t M, int64_t C_in,
                        int64_t hidden) {
  return codetr_ffn_relu_ln_f16(stream, x_dev, w1_dev, b1_dev, w2_packed_dev, b2_dev, y_dev, M, C_in, hidden, nullptr,
                                nullptr, 0.f, nullptr, nullptr);
}

}  // extern "C"
