// Fused transformer FFN for MI355X (gfx950):  Y = X + relu(X . W1^T + b1) . W2^T + b2
// (reference codetr/transformer_mmcv.py:484-500: Linear -> ReLU -> Linear + identity; encoder / decoder FFN of
// Co-DINO: C = 256, hidden = 2048 -- 37 % of the model's flops at 1920x1280).
//
// As two GEMMs the hidden activation [M, 2048] (838 MB in fp16 at M = 204 600) is written to HBM by the first and
// read back by the second, and each GEMM re-reads its activation tile from L2 once per 128-column output tile.  Here
// the hidden activation never leaves the CU:
//   * a 256-thread workgroup owns 128 rows; each wave keeps its 32 rows of X as MFMA B-fragments in registers
//     (2 m-tiles x 8 k-steps) for the whole kernel and its 32 x 256 slice of Y in accumulators;
//   * the hidden dimension is walked in chunks of 64: H^T[h][m] = W1c . X^T  (K = 256), bias folded into the
//     accumulator init, ReLU, fp16 pack -- and the packed accumulator IS the B operand of the second product
//     (cdna_hip_programming.md section 3 "accumulator tile as the next MFMA's operand": the k-slot permutation
//     8g+j <-> rows {4g..4g+3} of two 16-row tiles is baked into W2 once, by codetr_ffn_pack_w2_f16, so that the
//     matching A fragment is one ds_read_b128);
//     Y^T[n][m] += W2c . relu(H)^T  (K = 64);
//   * W1 / W2 chunks (32 KiB each) stream through a 2-stage LDS ring by LDS-DMA, XOR-swizzled on the source address
//     so that the ds_read_b128 fragment reads are conflict-free; one barrier per chunk (128 MFMAs per wave);
//     with one wave per SIMD nothing else hides latency or issue cost, so the schedule is spelled out: fragments are
//     read one step ahead of their MFMAs (behind the first two MFMAs of a group, so the wait in front of the group
//     is for reads that landed long ago), and the 16 LDS-DMA pieces of the next chunk go out one per MFMA group
//     instead of as a burst at the top of the chunk (~100 issue cycles per piece with the SIMD otherwise idle);
//     ablation at M = 204 600: no DMA -13 %, no barrier -5 %; b1 sits in LDS so that no other vector-memory op
//     (and no s_waitcnt vmcnt) lands between the DMA pieces;
//   * the epilogue adds the residual X and streams whole rows out through LDS, like the linear kernel.
// Algorithmic HBM traffic: X once in, Y once out (2 x M x 256 x 2 B) + 2 MB of weights re-read from L2 per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codetr_hip.h"

namespace {

constexpr int C = 256;         // model width (K of the first product, N of the second)
constexpr int BH = 64;         // hidden units per chunk
constexpr int kThreads = 256;
constexpr int kW1Bytes = BH * C * 2;   // 32 KiB: [64 h][256 k]
constexpr int kW2Bytes = C * BH * 2;   // 32 KiB: [256 n][64 h]
constexpr int kStageBytes = kW1Bytes + kW2Bytes;
constexpr int kOutPitch = C * 2 + 16;  // staged output row: 512 B + 16
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

__device__ __forceinline__ unsigned xcd_tile(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// One LDS-DMA piece (256 threads x 16 B = 4 KiB) of chunk `c` (hidden units c*64 .. c*64+63) of W1 [Hd, 256] and
// W2 [256, Hd]: pieces 0..7 fill the W1 image, 8..15 the W2 image of one LDS stage.
template <int PIECE>
__device__ __forceinline__ void stage_piece(const unsigned short* __restrict__ W1, const unsigned short* __restrict__ W2,
                                            int Hd, int c, unsigned char* stage, int tid) {
  const int wave = tid >> 6;
  if constexpr (PIECE < 8) {
    // W1 chunk: 64 rows x 32 chunks of 16 B; LDS position p of row r holds source chunk p ^ (r & 15) (low 4 bits)
    const int u = PIECE * kThreads + tid;
    const int r = u >> 5, pos = u & 31;
    const int chunk = pos ^ (r & 15);
    const unsigned short* g = W1 + (size_t)(c * BH + r) * C + chunk * 8;
    unsigned char* l = stage + (PIECE * kThreads + wave * 64) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  } else {
    // W2 chunk: 256 rows (n) x 8 chunks of 16 B (64 hidden units); position p of row n holds chunk p ^ ((n >> 1) & 7)
    constexpr int q = PIECE - 8;
    const int u = q * kThreads + tid;
    const int n = u >> 3, pos = u & 7;
    const int chunk = pos ^ ((n >> 1) & 7);
    const unsigned short* g = W2 + (size_t)n * Hd + c * BH + chunk * 8;
    unsigned char* l = stage + kW1Bytes + (q * kThreads + wave * 64) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  }
}

template <int... P>
__device__ __forceinline__ void stage_pieces(const unsigned short* __restrict__ W1, const unsigned short* __restrict__ W2,
                                             int Hd, int c, unsigned char* stage, int tid) {
  (stage_piece<P>(W1, W2, Hd, c, stage, tid), ...);
}

__device__ __forceinline__ void stage_chunk(const unsigned short* __restrict__ W1, const unsigned short* __restrict__ W2,
                                            int Hd, int c, unsigned char* stage, int tid) {
  stage_pieces<0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15>(W1, W2, Hd, c, stage, tid);
}

// piece number as a loop variable of an unrolled loop
template <int N>
struct PieceSwitch {
  static __device__ __forceinline__ void run(int piece, const unsigned short* __restrict__ W1,
                                             const unsigned short* __restrict__ W2, int Hd, int c, unsigned char* stage,
                                             int tid) {
    if (piece == N) stage_piece<N>(W1, W2, Hd, c, stage, tid);
    else PieceSwitch<N - 1>::run(piece, W1, W2, Hd, c, stage, tid);
  }
};
template <>
struct PieceSwitch<-1> {
  static __device__ __forceinline__ void run(int, const unsigned short*, const unsigned short*, int, int, unsigned char*,
                                             int) {}
};

// MT = 16-row tiles per wave (rows per workgroup = 64 * MT)
template <class ET, int MT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_fused_kernel(
    const unsigned short* __restrict__ X, const unsigned short* __restrict__ W1, const unsigned short* __restrict__ b1,
    const unsigned short* __restrict__ W2, const unsigned short* __restrict__ b2, unsigned short* __restrict__ Y, int M,
    int Hd, const unsigned short* __restrict__ ln_g, const unsigned short* __restrict__ ln_b, float ln_eps,
    const unsigned short* __restrict__ pos, unsigned short* __restrict__ Y2, int row0,
    const unsigned short* __restrict__ lnin_g, const unsigned short* __restrict__ lnin_b, float lnin_eps) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kStageBytes + kMaxHidden * 2 + 256 * 8];  // 146 KiB, one object
  using E = typename ET::e;
  using V8 = typename ET::v8;
  using V4 = typename ET::v4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, grp = lane >> 4;
  constexpr int BM = 64 * MT, WR = 16 * MT;  // rows per workgroup / per wave
  const int m0 = row0 + (int)xcd_tile(blockIdx.x, gridDim.x) * BM + wave * WR;  // this wave's first row
  const int nchunks = Hd / BH;

  stage_chunk(W1, W2, Hd, 0, lds, tid);

  // X fragments of this wave's 32 rows (B operand: lane (j = l15, g) holds X[m][32*ks + 8g .. +7]), kept for good
  V8 xf[MT][8];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = m0 + mt * 16 + l15;
    m = m < M ? m : M - 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      xf[mt][ks] = *reinterpret_cast<const V8*>(X + (size_t)m * C + ks * 32 + grp * 8);
  }
  // Optional LayerNorm of the INPUT rows (the post-norm layer's first norm, whose output nothing else reads): the
  // 256 values of row (mt, l15) sit in the four lanes l15 + 16 g of this wave (8 k-steps x 8 values each), so the
  // statistics are two xor-shuffles away; fp32 two-pass like layernorm_kernel, result rounded to f16 = the operand the
  // separate kernel would have written.  (mean, rstd) go to LDS so that the epilogue rebuilds exactly the same rows
  // for the residual.
  float* sStat = reinterpret_cast<float*>(lds + 2 * kStageBytes + kMaxHidden * 2) + wave * (WR * 2);
  if (lnin_g) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float sm = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += (float)xf[mt][ks][e];
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      const float mean = sm * (1.0f / C);
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)xf[mt][ks][e] - mean;
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const float rstd = rsqrtf(q * (1.0f / C) + lnin_eps);
      if (grp == 0) {
        sStat[(mt * 16 + l15) * 2] = mean;
        sStat[(mt * 16 + l15) * 2 + 1] = rstd;
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const V8 gw = *reinterpret_cast<const V8*>(lnin_g + ks * 32 + grp * 8);
        const V8 gb = *reinterpret_cast<const V8*>(lnin_b + ks * 32 + grp * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          xf[mt][ks][e] = (E)fmaf(((float)xf[mt][ks][e] - mean) * rstd, (float)gw[e], (float)gb[e]);
      }
    }
  }
  // Y accumulators start at b2 (lane's 4 consecutive n of tile nt: n = 16*nt + 4*grp + r)
  f32x4 yacc[16][MT];
#pragma unroll
  for (int nt = 0; nt < 16; ++nt) {
    const V4 bb = *reinterpret_cast<const V4*>(b2 + nt * 16 + grp * 4);
    const f32x4 b4 = {(float)bb[0], (float)bb[1], (float)bb[2], (float)bb[3]};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) yacc[nt][mt] = b4;
  }

  // b1 goes to LDS once (read per chunk by ds_read_b64: no vector-memory op in the main loop but the LDS-DMA pieces,
  // so no s_waitcnt vmcnt lands between them)
  unsigned short* sB1 = reinterpret_cast<unsigned short*>(lds + 2 * kStageBytes);
  for (int i = tid; i < Hd / 8; i += kThreads)
    *reinterpret_cast<s16x8*>(sB1 + i * 8) = *reinterpret_cast<const s16x8*>(b1 + i * 8);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // ... written before the first barrier below

  for (int c = 0; c < nchunks; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk c landed (and the X loads issued earlier)
    __builtin_amdgcn_s_barrier();                      // ... for everyone; everyone is done with chunk c-1's stage
    // the next chunk's 16 LDS-DMA pieces are issued one per MFMA group below (a burst here costs the wave ~100 issue
    // cycles per piece with nothing else to run on its SIMD); past the last chunk they re-fetch it (no branch in the
    // pinned schedule), which the s_waitcnt before the epilogue barrier drains
    const int cn = c + 1 < nchunks ? c + 1 : c;
    unsigned char* next_stage = lds + ((c + 1) & 1) * kStageBytes;
    const unsigned char* sW1 = lds + (c & 1) * kStageBytes;
    const unsigned char* sW2 = sW1 + kW1Bytes;

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
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      // group ks: 2 MFMAs, then the 4 reads of step ks+1 (the reads of step ks, issued a group earlier, have landed
      // by the time hipcc's s_waitcnt in front of the first MFMA runs), one LDS-DMA piece of the next chunk, then the
      // other 4*MT-2 MFMAs.  The DMA instruction ends a scheduling region, so the order is spelled out in source and
      // the pins only keep the reads behind the first two MFMAs.
#pragma unroll
      for (int i = 0; i < 2; ++i)
        hacc[i / MT][i % MT] =
            ET::mfma(aw[ks & 1][i / MT], xf[i % MT][ks], hacc[i / MT][i % MT]);
      if (ks + 1 < 8) {
        read_w1(ks + 1, aw[(ks + 1) & 1]);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      }
      PieceSwitch<7>::run(ks, W1, W2, Hd, cn, next_stage, tid);
#pragma unroll
      for (int i = 2; i < 4 * MT; ++i)
        hacc[i / MT][i % MT] =
            ET::mfma(aw[ks & 1][i / MT], xf[i % MT][ks], hacc[i / MT][i % MT]);
    }
    // ---- ReLU + pack: B operand of the second product, k-slot 8g+j = rows 4g..4g+3 of tiles 2s and 2s+1 ----
    V8 pf[2][MT];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) pf[s][mt][h * 4 + r] = (E)hacc[2 * s + h][mt][r];
    // ReLU on the packed halves (v_pk_max_f16: one op per two values).  max(NaN, 0) = 0 drops a NaN of the hidden
    // unit, but a NaN there can only come from a NaN / inf in this row of X, which the residual add puts back.
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        pf[s][mt] = ET::relu(pf[s][mt]);
      }
    // ---- Y^T += W2c . relu(H)^T : D[i = n][j = m], k = hidden unit (permuted identically on both operands) ----
    // W2 fragments (pre-packed: the 8 k-slots of lane group g are 16 contiguous bytes) are read two n-tiles
    // ahead of their MFMAs, same double buffering
    auto read_w2 = [&](int ntp, V8 (&a)[4]) {  // n-tiles 2*ntp, 2*ntp+1; index [t*2 + s]
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (2 * ntp + t) * 16 + l15;
        const unsigned char* rowp = sW2 + n * (BH * 2);
        const int sw = (n >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 2; ++s)
          a[t * 2 + s] = *reinterpret_cast<const V8*>(rowp + ((4 * s + grp) ^ sw) * 16);
      }
    };
    V8 a2[2][4];
    read_w2(0, a2[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int ntp = 0; ntp < 8; ++ntp) {
      // same group shape: MFMA index i -> (t, s, mt)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ts = i / MT, mt = i % MT, nt = 2 * ntp + ts / 2;
        yacc[nt][mt] = ET::mfma(a2[ntp & 1][ts], pf[ts & 1][mt], yacc[nt][mt]);
      }
      if (ntp + 1 < 8) {
        read_w2(ntp + 1, a2[(ntp + 1) & 1]);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      }
      PieceSwitch<15>::run(8 + ntp, W1, W2, Hd, cn, next_stage, tid);
#pragma unroll
      for (int i = 2; i < 4 * MT; ++i) {
        const int ts = i / MT, mt = i % MT, nt = 2 * ntp + ts / 2;
        yacc[nt][mt] = ET::mfma(a2[ntp & 1][ts], pf[ts & 1][mt], yacc[nt][mt]);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant DMA of the last iteration has landed
  __builtin_amdgcn_s_barrier();                      // every wave is done with the last stage: LDS is free

  // ---- epilogue: Y tile of this wave (32 rows x 256) through LDS, + residual X, whole 512-byte rows out ----
  unsigned char* stage = lds + wave * (WR * kOutPitch);
#pragma unroll
  for (int nt = 0; nt < 16; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const V4 o = {(E)yacc[nt][mt][0], (E)yacc[nt][mt][1], (E)yacc[nt][mt][2],
                       (E)yacc[nt][mt][3]};
      *reinterpret_cast<V4*>(stage + (mt * 16 + l15) * kOutPitch + (nt * 16 + grp * 4) * 2) = o;
    }
  __builtin_amdgcn_wave_barrier();
  // WR rows x 32 chunks of 16 B: lane -> (row = it*2 + lane/32, chunk = lane%32).  With one wave per SIMD a
  // load -> add -> store chain per row pair would expose one memory latency per iteration (measured +85 us per launch
  // once the epilogue also read pos): all residual / pos rows are requested first -- the accumulators are dead by
  // now, the registers are free -- and consumed afterwards.
  const int chunk = lane & 31;
  V8 xr[WR / 2], pr[WR / 2];
#pragma unroll
  for (int it = 0; it < WR / 2; ++it) {
    int m = m0 + it * 2 + (lane >> 5);
    m = m < M ? m : M - 1;
    xr[it] = *reinterpret_cast<const V8*>(X + (size_t)m * C + chunk * 8);
  }
  if (Y2) {
#pragma unroll
    for (int it = 0; it < WR / 2; ++it) {
      int m = m0 + it * 2 + (lane >> 5);
      m = m < M ? m : M - 1;
      pr[it] = *reinterpret_cast<const V8*>(pos + (size_t)m * C + chunk * 8);
    }
  }
  V8 gw, gb, gin_w, gin_b;
  if (ln_g) {
    gw = *reinterpret_cast<const V8*>(ln_g + chunk * 8);
    gb = *reinterpret_cast<const V8*>(ln_b + chunk * 8);
  }
  if (lnin_g) {
    gin_w = *reinterpret_cast<const V8*>(lnin_g + chunk * 8);
    gin_b = *reinterpret_cast<const V8*>(lnin_b + chunk * 8);
  }
#pragma unroll
  for (int it = 0; it < WR / 2; ++it) {
    const int row = it * 2 + (lane >> 5);
    const int m = m0 + row;
    const V8 y = *reinterpret_cast<const V8*>(stage + row * kOutPitch + chunk * 16);
    V8 xrow = xr[it];
    if (lnin_g) {  // identity = LayerNorm(input row), the same arithmetic on the same statistics as the prologue
      const float mean = sStat[row * 2], rstd = sStat[row * 2 + 1];
#pragma unroll
      for (int e = 0; e < 8; ++e) xrow[e] = (E)fmaf(((float)xrow[e] - mean) * rstd, (float)gin_w[e], (float)gin_b[e]);
    }
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (E)((float)y[e] + (float)xrow[e]);  // identity + ffn(x): fp16 + fp16 -> fp16
    if (ln_g) {
      // LayerNorm over the row (its 256 values sit in the 32 lanes of this half-wave): the arithmetic of
      // layernorm_kernel<LnHalf, 32, 1>, statement for statement, so the result is bit-identical to running that
      // kernel on the stored sum
      float sm = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sm += (float)o[e];
#pragma unroll
      for (int d = 16; d > 0; d >>= 1) sm += __shfl_xor(sm, d, 64);
      const float mean = sm * (1.0f / C);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dv = (float)o[e] - mean;
        q = fmaf(dv, dv, q);
      }
#pragma unroll
      for (int d = 16; d > 0; d >>= 1) q += __shfl_xor(q, d, 64);
      const float rstd = rsqrtf(q * (1.0f / C) + ln_eps);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (E)fmaf(((float)o[e] - mean) * rstd, (float)gw[e], (float)gb[e]);
    }
    if (m < M) {
      *reinterpret_cast<V8*>(Y + (size_t)m * C + chunk * 8) = o;
      if (Y2) {  // the next layer's attention input: this row + its positional encoding (fp16 + fp16 -> fp16)
        V8 o2;
#pragma unroll
        for (int e = 0; e < 8; ++e) o2[e] = (E)((float)o[e] + (float)pr[it][e]);
        *reinterpret_cast<V8*>(Y2 + (size_t)m * C + chunk * 8) = o2;
      }
    }
  }
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
                            void* y_plus_pos_dev) {
  const void* w2_dev = w2_packed_dev;
  if (!x_dev || !w1_dev || !b1_dev || !w2_dev || !b2_dev || !y_dev || M <= 0 || hidden <= 0) return CODETR_E_BADARG;
  if ((ln_gamma_dev == nullptr) != (ln_beta_dev == nullptr) || (pos_dev == nullptr) != (y_plus_pos_dev == nullptr) ||
      (ln_in_gamma_dev == nullptr) != (ln_in_beta_dev == nullptr))
    return CODETR_E_BADARG;
  if (C_in != C || hidden % BH != 0 || hidden > kMaxHidden) return CODETR_E_UNSUPPORTED;
  if (M > 0x7fffffffLL - 256 || hidden > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  // MT = 2 (128 rows per workgroup).  MT = 3 fits the register file only without the interleaved DMA issue (236 VGPR
  // + 240 AGPR, 570 us at M = 204 600 against 545 us for this variant); with it hipcc spills (1147 us).
  // (Splitting off the last, mostly empty round of 256 workgroups as 64-row workgroups measured -1 %: workgroups are
  // not dispatched in lockstep rounds, so the tail is already spread.)
  constexpr int kMT = 2;
  const unsigned blocks = (unsigned)((M + 64 * kMT - 1) / (64 * kMT));
  hipLaunchKernelGGL((ffn_fused_kernel<ET, kMT>), dim3(blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(x_dev), static_cast<const unsigned short*>(w1_dev),
                     static_cast<const unsigned short*>(b1_dev), static_cast<const unsigned short*>(w2_dev),
                     static_cast<const unsigned short*>(b2_dev), static_cast<unsigned short*>(y_dev), (int)M,
                     (int)hidden, static_cast<const unsigned short*>(ln_gamma_dev),
                     static_cast<const unsigned short*>(ln_beta_dev), ln_eps,
                     static_cast<const unsigned short*>(pos_dev), static_cast<unsigned short*>(y_plus_pos_dev), 0,
                     static_cast<const unsigned short*>(ln_in_gamma_dev),
                     static_cast<const unsigned short*>(ln_in_beta_dev), ln_in_eps);
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
