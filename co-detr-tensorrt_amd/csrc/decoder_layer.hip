// One DINO decoder layer's row-local work in ONE launch for MI355X (gfx950).
//
// Reference codetr/transformer.py:193-230 (DinoTransformerDecoder.forward: per layer the reference-point scaling, the
// sine embedding, ref_point_head, the layer, the box refinement) and :233-277 / transformer_mmcv.py:583-749
// (DetrTransformerDecoderLayer = self_attn, norm, cross_attn, norm, ffn, norm).  As separate launches that is ~19
// kernels per layer on 900 rows -- 7-11 us each for a microsecond of work -- and half of the launches of a whole
// forward.  Everything in a layer except the self-attention's softmax(QK^T)V is ROW-LOCAL (a query row needs only its
// own activations, the weights and the read-only value map), so between two self-attentions one kernel serves:
//
//   TAIL of layer l (skipped in the first launch):
//     x1 = LN1(x + attn . Wo^T + bo)                                        transformer_mmcv.py:394-428 (out_proj, identity)
//     proj = (x1 + qpos) . [W_offsets | W_logits]^T + b                      multi_scale_deformable_attention.py:165-178
//     s = MSDA(value_l, softmax(logits), ref_xy + off / P * ref_wh * 0.5)    :180-196 + csrc/ms_deform_attn.cu:31-77, 211-261
//     x2 = LN2(x1 + s . Wout^T + bout)
//     x3 = LN3(x2 + relu(x2 . W1^T + b1) . W2^T + b2)                        transformer_mmcv.py:484-500
//     ref' = ref + reg_branch_l(x3)   (unactivated boxes)                    transformer.py:219-227
//   HEAD of layer l + 1 (the last launch applies the decoder's output norm instead):
//     ref_in = sigmoid(ref') * valid_ratios;  qpos' = ref_point_head(sine_embed(ref_in[:, 0]))     transformer.py:208-217, 157-190
//     [q | k] = (x3 + qpos') . Wqk^T + bqk,  v = x3 . Wv^T + bv              nn.MultiheadAttention in-projection
//
// A workgroup owns 16 query rows (one MFMA row tile) and streams every weight matrix from L2 exactly once:
// Y^T[n][m] = W . X^T with W fragments read straight from global memory (16 rows x 64 B per wave instruction) and the
// 16 activation rows as the B operand out of LDS; a lane ends up with 4 consecutive output columns of one row.  The
// weights do not depend on the data, so the fragments of the NEXT product are requested before the epilogue / barrier /
// LayerNorm of the current one.  8 waves: each takes every 8th 16-column tile.  ~3.6 MB of weights per layer and
// workgroup; the launch is bound by that L2 stream (and, at one image, by nothing else: 57 workgroups).
// Rounding points follow the unfused path (fp16 where it materialises a tensor): tests compare the two.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>
#include <utility>

#include "codetr_hip.h"

// diagnostic builds only (-DCODETR_DEC_ABL=mask; WRONG results by construction, never shipped): 1 = no MSDA gather
#ifndef CODETR_DEC_ABL
#define CODETR_DEC_ABL 0
#endif

#ifdef CODETR_DEC_STAMPS   // diagnostic builds only: in-kernel clock at the phase boundaries of workgroup 0, wave 0
__device__ unsigned long long g_dec_stamps[32];
#define DEC_STAMP(i)                                                          \
  do {                                                                        \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_dec_stamps[(TAIL && HEAD ? 0 : 16) + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define DEC_STAMP(i) \
  do {               \
  } while (0)
#endif

namespace {

constexpr int kC = 256;          // embed dims
constexpr int kM = 8;            // heads
constexpr int kD = 32;           // head dim
constexpr int kF = 2048;         // FFN hidden width (8 chunks of 256 in the static weight schedule)
constexpr int kRows = 16;        // query rows per workgroup
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kSC = kC + 8;      // LDS row stride (halfs) of a [16][256] fp16 buffer: 528 B, conflict-free fragment reads
constexpr int kS2 = 2 * kC + 8;  // ... of a [16][512] buffer
constexpr int kSF = kC + 4;      // fp32 row stride of the [16][256] pre-LayerNorm buffer
constexpr int kMaxLP = 32;
constexpr int kMaxL = 8;
#ifndef CODETR_DEC_DEPTH
#define CODETR_DEC_DEPTH 32
#endif
constexpr int kDepth = CODETR_DEC_DEPTH;       // weight fragments (1 KB each) a wave keeps in flight: 256 KB per workgroup

// Element type of activations and weights.  The kernel is written against `f16` / `f16x8` / `f16x4` and one MFMA wrapper;
// decoder_layer_bf16.hip compiles this same source with CODETR_DEC_BF16 defined: bf16 storage, v_mfma_f32_16x16x32_bf16,
// the same fp32 arithmetic everywhere in between (LayerNorms, softmax, sampling geometry, the fp32 blend of the gather),
// entry point codetr_decoder_layer_bf16.  The pure-host queries exist once, in the fp16 translation unit.
#ifdef CODETR_DEC_BF16
typedef __bf16 f16;
typedef __bf16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 f16x4 __attribute__((ext_vector_type(4)));
#define CODETR_DEC_ENTRY codetr_decoder_layer_bf16
#else
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define CODETR_DEC_ENTRY codetr_decoder_layer_f16
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- packed weight blobs (include/codetr_hip.h documents the order): matrices first, then the small vectors, which
// the kernel copies to LDS once (a bias read from global memory would sit behind the whole weight stream: vector
// memory returns in order) ----
struct TailW {
  int wo, wol, wout, w1, w2, wr1, wr2, wr3, vec;                                  // matrices, start of the vectors
  int bo, g1, e1, bol, bout, g2, e2, b1, b2, g3, e3, br1, br2, br3, nvec, total;  // vectors: offsets from `vec`
};
__host__ __device__ inline TailW tail_layout(int n_ol) {
  TailW t;
  int o = 0;
  auto take = [&](int n) { const int at = o; o += n; return at; };
  t.wo = take(kC * kC); t.wol = take(512 * kC); t.wout = take(kC * kC); t.w1 = take(kF * kC); t.w2 = take(kC * kF);
  t.wr1 = take(kC * kC); t.wr2 = take(kC * kC); t.wr3 = take(16 * kC);
  t.vec = o;
  o = 0;
  t.bo = take(kC); t.g1 = take(kC); t.e1 = take(kC); t.bol = take(n_ol); t.bout = take(kC); t.g2 = take(kC); t.e2 = take(kC);
  t.b1 = take(kF); t.b2 = take(kC); t.g3 = take(kC); t.e3 = take(kC); t.br1 = take(kC); t.br2 = take(kC); t.br3 = take(8);
  t.nvec = o;
  t.total = t.vec + t.nvec;
  return t;
}
constexpr int kHeadWqk = 0, kHeadWv = 2 * kC * kC, kHeadVec = kHeadWv + kC * kC, kHeadBqk = 0, kHeadBv = 2 * kC,
              kHeadNVec = 3 * kC, kHeadTotal = kHeadVec + kHeadNVec;
constexpr int kPosW1 = 0, kPosW2 = kC * 2 * kC, kPosVec = kPosW2 + kC * kC, kPosB1 = 0, kPosB2 = kC, kPosNVec = 2 * kC,
              kPosTotal = kPosVec + kPosNVec;
constexpr int kMaxTailVec = 11 * kC + 512 + kF + 8;   // n_ol <= 512

struct DecArgs {
  const f16* x; const f16* attn; const f16* qpos; const f16* ref; const float* vr32; const f16* value;
  const int64_t* shapes; const int64_t* starts;
  const f16* tail_w; const f16* pos_w; const f16* head_w; const f16* final_norm;
  f16* x_out; f16* ref_out; f16* qpos_out; f16* qk_out; f16* v_out;
  int rows, Nq, S, L, P, n_ol;
  float eps, log2_temperature;
  TailW tw;
};

struct Entry {
  u32x4 off;
  f32x4 w;
};

// ---- the static weight schedule ----
// Every wave consumes the same sequence of weight fragments (a fragment = the wave's 16 rows x 32 k of one matrix,
// 1 KB): per product, tiles wave, wave + 8, ... x 8 k-steps.  The sequence is known at compile time, so fragment
// i + kDepth is requested when fragment i is consumed -- across products, barriers, LayerNorms: the stream never drains.
enum : int { M_WO, M_WOL, M_WOUT, M_W1, M_W2, M_WR1, M_WR2, M_WR3, M_WP1, M_WP2, M_WQK, M_WV, M_COUNT };
struct Ph {
  int mat, nt, sub;   // matrix, 16-row tiles per wave, sub-block (FFN chunk / K half of Wp1)
};
constexpr int kTailPhases = 3 + 2 * (kF / 256) + 3, kHeadPhases = 5;
template <bool TAIL, bool HEAD>
struct Sched {
  static constexpr int kHeadBase = TAIL ? kTailPhases : 0;
  static constexpr int kN = kHeadBase + (HEAD ? kHeadPhases : 0);
  static constexpr Ph at(int i) {
    if (TAIL) {
      if (i == 0) return Ph{M_WO, 2, 0};
      if (i == 1) return Ph{M_WOL, 4, 0};
      if (i == 2) return Ph{M_WOUT, 2, 0};
      if (i < 3 + 2 * (kF / 256)) return ((i - 3) & 1) ? Ph{M_W2, 2, (i - 3) >> 1} : Ph{M_W1, 2, (i - 3) >> 1};
      if (i == kTailPhases - 3) return Ph{M_WR1, 2, 0};
      if (i == kTailPhases - 2) return Ph{M_WR2, 2, 0};
      if (i == kTailPhases - 1) return Ph{M_WR3, 1, 0};
      i -= kTailPhases;
    }
    return i == 0 ? Ph{M_WP1, 2, 0} : i == 1 ? Ph{M_WP1, 2, 1} : i == 2 ? Ph{M_WP2, 2, 0} : i == 3 ? Ph{M_WQK, 4, 0} : Ph{M_WV, 2, 0};
  }
  static constexpr int start(int p) {
    int s = 0;
    for (int i = 0; i < p; ++i) s += at(i).nt * 8;
    return s;
  }
  static constexpr int total = start(kN);
  static constexpr int limit(int p) { return (TAIL && p <= 1) ? start(3) : total; }   // phases 0, 1 = Wo, Wol; 2 = Wout
  struct Loc {
    int p, t, ks;
  };
  static constexpr Loc locate(int I) {
    int p = 0;
    while (I >= at(p).nt * 8) {
      I -= at(p).nt * 8;
      ++p;
    }
    return Loc{p, I / 8, I % 8};
  }
};
// phase numbers of the tail (the head's are Sched::kHeadBase + 0..4)
constexpr int P_WO = 0, P_WOL = 1, P_WOUT = 2, P_FFN = 3, P_WR1 = kTailPhases - 3, P_WR2 = kTailPhases - 2, P_WR3 = kTailPhases - 1;

struct Stream {
  f16x8 fifo[kDepth];
  const f16* mat[M_COUNT];
  int wave, lane8, rot;
};

// Matrices are stored FRAGMENT-MAJOR (packed once on the host, include/codetr_hip.h): fragment (tile, ks) of a [N][K]
// matrix is the 1 KB block at ((tile * K/32 + ks) * 64 + lane) * 8 halfs holding W[16 tile + l15][32 ks + 8 grp .. + 7] for
// lane = 16 grp + l15 -- a wave instruction reads 8 whole 128-byte lines (row-major rows would be 16 half lines) and
// the lane's address is a scalar + 16 * lane.
template <class S, int I>
__device__ __forceinline__ void issue(Stream& c) {
  if constexpr (I < S::total) {
    constexpr typename S::Loc loc = S::locate(I);
    constexpr Ph ph = S::at(loc.p);
    constexpr int KST = ph.mat == M_W2 ? kF / 32 : (ph.mat == M_WP1 ? 2 * kC / 32 : kC / 32);   // k-steps of the whole matrix
    if constexpr (ph.mat == M_W1 || ph.mat == M_W2) {
      // FFN chunk `sub` of the schedule is hidden chunk (sub + rot) & 7 of the matrices: the workgroups of an XCD walk
      // the chunks in rotated orders, so that they do not all ask one L2 channel for the same lines at the same time
      const int ch = (ph.sub + c.rot) & (kF / 256 - 1);
      const int tile = (ph.mat == M_W1 ? ch * 16 : 0) + loc.t * kWaves + c.wave;
      const int ks = (ph.mat == M_W2 ? ch * 8 : 0) + loc.ks;
      c.fifo[I % kDepth] = *reinterpret_cast<const f16x8*>(c.mat[ph.mat] + ((size_t)(tile * KST + ks) * 64) * 8 + c.lane8);
    } else {
      // (rotating the k-steps of the other products per workgroup as well was measured: slower, 858 -> 933 us per decoder)
      constexpr int ks = (ph.mat == M_WP1 ? ph.sub * 8 : 0) + loc.ks;
      constexpr int tile_c = loc.t * kWaves;               // + wave
      const int tile = ph.mat == M_WR3 ? 0 : tile_c + c.wave;   // (Wr3 is ONE tile: every wave walks it, wave 0's result counts)
      const f16* p = c.mat[ph.mat] + ((size_t)(tile * KST + ks) * 64) * 8 + c.lane8;
      c.fifo[I % kDepth] = *reinterpret_cast<const f16x8*>(p);
    }
  }
}
// fragments issued once step I of the schedule has run: kDepth ahead, but never past the phase's limit -- in front of
// the MSDA gather the stream is allowed to run down to the next product's fragments, so that the gather has the
// registers for ten points' corner rows in flight
template <class S>
constexpr int issued_upto(int I) {
  if (I < 0) return kDepth < S::limit(0) ? kDepth : S::limit(0);
  const int want = I + 1 + kDepth, lim = S::limit(S::locate(I < S::total ? I : S::total - 1).p);
  const int prev = issued_upto<S>(I - 1);
  const int now = want < lim ? want : lim;
  return now > prev ? now : prev;
}
template <class S, int LO, int... Ks>
__device__ __forceinline__ void issue_range(Stream& c, std::integer_sequence<int, Ks...>) {
  (issue<S, LO + Ks>(c), ...);
}
template <class S, int... Is>
__device__ __forceinline__ void prime(Stream& c, std::integer_sequence<int, Is...>) {
  (issue<S, Is>(c), ...);
}

#ifdef CODETR_DEC_BF16
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#else
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
#endif

// one product of the schedule: acc[t] += W(tile wave + 8 t) . X^T over 8 k-steps, every consumed fragment replaced by
// the one kDepth further down the stream
template <class S, int P, int NT, int J>
__device__ __forceinline__ void gemm_step(Stream& c, f32x4 (&acc)[NT], const f16x8 (&xf)[8]) {
  constexpr int I = S::start(P) + J;
  acc[J / 8] = mfma16(c.fifo[I % kDepth], xf[J % 8], acc[J / 8]);
  constexpr int lo = issued_upto<S>(I - 1), hi = issued_upto<S>(I);
  issue_range<S, lo>(c, std::make_integer_sequence<int, hi - lo>{});
}
template <class S, int P, int NT, int... Js>
__device__ __forceinline__ void gemm_steps(Stream& c, f32x4 (&acc)[NT], const f16x8 (&xf)[8], std::integer_sequence<int, Js...>) {
  (gemm_step<S, P, NT, Js>(c, acc, xf), ...);
}
template <class S, int P, int NT>
__device__ __forceinline__ void gemm(Stream& c, f32x4 (&acc)[NT], const f16x8 (&xf)[8]) {
  static_assert(S::at(P).nt == NT, "accumulator tiles do not match the schedule");
  gemm_steps<S, P, NT>(c, acc, xf, std::make_integer_sequence<int, NT * 8>{});
}

// the 16 activation rows of an LDS buffer as MFMA B fragments: lane (row l15, group grp) holds X[row][32 ks + 8 grp ..]
__device__ __forceinline__ void xload(f16x8 (&xf)[8], const f16* xs, const int stride, const int l15, const int grp) {
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) xf[ks] = *reinterpret_cast<const f16x8*>(xs + l15 * stride + ks * 32 + grp * 8);
}

__device__ __forceinline__ f16x4 ld4(const f16* p) { return *reinterpret_cast<const f16x4*>(p); }
__device__ __forceinline__ void st4(f16* p, f16x4 v) { *reinterpret_cast<f16x4*>(p) = v; }

// workgroup barrier that orders LDS traffic only: the weight fragments in flight stay in flight
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LayerNorm of rows 2 * wave, 2 * wave + 1 of an fp16 (SRC32 = false) or fp32 [16][256] LDS buffer; gamma / beta in
// LDS.  Two-pass statistics in fp32 as csrc/layernorm.hip.  `emit(row, col, y[4])` receives the result.
template <bool SRC32, class Emit>
__device__ __forceinline__ void ln_rows(const void* src, const f16* gamma, const f16* beta, const float eps, const int wave,
                                        const int lane, Emit emit) {
  const f16x4 g4 = ld4(gamma + 4 * lane), b4 = ld4(beta + 4 * lane);
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int row = 2 * wave + rr;
    float v[4];
    if (SRC32) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(static_cast<const float*>(src) + row * kSF + 4 * lane);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = t[e];
    } else {
      const f16x4 t = ld4(static_cast<const f16*>(src) + row * kSC + 4 * lane);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (float)t[e];
    }
    const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / kC);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[e] - mean;
      q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / kC) + eps);
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = fmaf((v[e] - mean) * rstd, (float)g4[e], (float)b4[e]);
    emit(row, 4 * lane, y);
  }
}

template <bool TAIL, bool HEAD>
__global__ __launch_bounds__(kThreads) void decoder_layer_kernel(const DecArgs a) {
  using S = Sched<TAIL, HEAD>;
  constexpr int HB_ = S::kHeadBase;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // ---- LDS map ----
  f16* XA = reinterpret_cast<f16*>(smem);                 // [16][kSC]  current activations (x, x1, x2, x3)
  f16* XB = XA + kRows * kSC;                             // [16][kSC]
  f16* XC = XB + kRows * kSC;                             // [16][kSC]
  f16* XD = XC + kRows * kSC;                             // [16][kSC]
  f16* QP = XD + kRows * kSC;                             // [16][kSC]  query_pos of the layer
  f16* PJ = QP + kRows * kSC;                             // [16][kS2]  (offsets | logits) of the rows; later the sine embedding
  f16* VT = PJ + kRows * kS2;                             // tail vectors (biases, LayerNorm parameters)
  f16* VH = VT + kMaxTailVec;                             // head vectors
  f16* VP = VH + kHeadNVec;                               // ref_point_head vectors
  f16* VF = VP + kPosNVec;                                // output norm
  float* RF = reinterpret_cast<float*>(VF + 2 * kC);      // [16][4] sigmoid(ref) fp32, then [16][4] unactivated boxes
  int* s_meta = reinterpret_cast<int*>(RF + 2 * kRows * 4);   // [kMaxL][4] level table (H, W, start, -)
  float* s_vr = reinterpret_cast<float*>(s_meta + kMaxL * 4); // [16 rows][kMaxL][2] valid ratios of each row's image
  unsigned char* EB = reinterpret_cast<unsigned char*>(s_vr + kRows * kMaxL * 2);   // big region: entries | hidden | fp32 rows
  Entry* entries = reinterpret_cast<Entry*>(EB);          // [LP][128 pairs]
  f16* HB = reinterpret_cast<f16*>(EB);                   // [2][16][kSC] hidden chunks of the FFN
  float* YF = reinterpret_cast<float*>(EB + 2 * kRows * kSC * sizeof(f16));   // [16][kSF] fp32 pre-LayerNorm rows

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, grp = lane >> 4;
  const int row0 = blockIdx.x * kRows;
  const int L = a.L, P = a.P, LP = L * P;
  // global row of the lane's MFMA output row / of a 32-thread row group (clamped: the last workgroup recomputes the
  // last valid row, its stores are masked)
  const int mrow = row0 + l15 < a.rows ? row0 + l15 : a.rows - 1;
  const bool mrow_ok = row0 + l15 < a.rows;
  const int cr = tid >> 5, cc = (tid & 31) * 8;   // row-copy role: 32 threads x 16 B per row
  const int crow = row0 + cr < a.rows ? row0 + cr : a.rows - 1;
  const TailW& tw = a.tw;

  // ---- everything small goes to LDS first: rows, vectors, level table, valid ratios, reference boxes ----
  *reinterpret_cast<f16x8*>(XA + cr * kSC + cc) = *reinterpret_cast<const f16x8*>(a.x + (size_t)crow * kC + cc);
  if (TAIL) {
    *reinterpret_cast<f16x8*>(XB + cr * kSC + cc) = *reinterpret_cast<const f16x8*>(a.attn + (size_t)crow * kC + cc);
    *reinterpret_cast<f16x8*>(QP + cr * kSC + cc) = *reinterpret_cast<const f16x8*>(a.qpos + (size_t)crow * kC + cc);
    for (int i = tid * 8; i < tw.nvec; i += kThreads * 8)
      *reinterpret_cast<f16x8*>(VT + i) = *reinterpret_cast<const f16x8*>(a.tail_w + tw.vec + i);
    if (!HEAD && tid < 2 * kC / 8) *reinterpret_cast<f16x8*>(VF + tid * 8) = *reinterpret_cast<const f16x8*>(a.final_norm + tid * 8);
    if (tid < L) {
      s_meta[4 * tid] = (int)a.shapes[2 * tid];
      s_meta[4 * tid + 1] = (int)a.shapes[2 * tid + 1];
      s_meta[4 * tid + 2] = (int)a.starts[tid];
      s_meta[4 * tid + 3] = 0;
    }
  }
  if (HEAD) {
    if (tid < kHeadNVec / 8) *reinterpret_cast<f16x8*>(VH + tid * 8) = *reinterpret_cast<const f16x8*>(a.head_w + kHeadVec + tid * 8);
    if (tid < kPosNVec / 8) *reinterpret_cast<f16x8*>(VP + tid * 8) = *reinterpret_cast<const f16x8*>(a.pos_w + kPosVec + tid * 8);
  }
  if (tid < kRows * L * 2) {
    const int r = tid / (L * 2), j = tid % (L * 2);
    const int gr = row0 + r < a.rows ? row0 + r : a.rows - 1;
    s_vr[r * kMaxL * 2 + j] = a.vr32[(size_t)(gr / a.Nq) * L * 2 + j];
  }
  if (tid < kRows * 4) {
    const int r = tid >> 2, gr = row0 + r < a.rows ? row0 + r : a.rows - 1;
    const float v = (float)a.ref[(size_t)gr * 4 + (tid & 3)];
    RF[tid] = 1.0f / (1.0f + expf(-v));      // (query_sine_embed.hip: s32)
    RF[kRows * 4 + tid] = v;                 // unactivated
  }

  // ---- start the weight stream ----
  Stream st;
  st.wave = wave;
  st.lane8 = lane * 8;
#ifdef CODETR_DEC_NOROT
  st.rot = 0;
#else
  st.rot = (blockIdx.x >> 3) & 7;   // (blocks x, x + 8, ... share an XCD)
#endif
  if (TAIL) {
    st.mat[M_WO] = a.tail_w + tw.wo;
    st.mat[M_WOL] = a.tail_w + tw.wol;
    st.mat[M_WOUT] = a.tail_w + tw.wout;
    st.mat[M_W1] = a.tail_w + tw.w1;
    st.mat[M_W2] = a.tail_w + tw.w2;
    st.mat[M_WR1] = a.tail_w + tw.wr1;
    st.mat[M_WR2] = a.tail_w + tw.wr2;
    st.mat[M_WR3] = a.tail_w + tw.wr3;
  }
  if (HEAD) {
    st.mat[M_WP1] = a.pos_w + kPosW1;
    st.mat[M_WP2] = a.pos_w + kPosW2;
    st.mat[M_WQK] = a.head_w + kHeadWqk;
    st.mat[M_WV] = a.head_w + kHeadWv;
  }
  DEC_STAMP(0);
  prime<S>(st, std::make_integer_sequence<int, issued_upto<S>(-1)>{});
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(issued_upto<S>(-1)) : "memory");   // the rows / vectors above have landed; the stream flies on
  lds_barrier();
  DEC_STAMP(1);

  if constexpr (TAIL) {
    // ================= out-projection of the self-attention + identity, LN1 =================
    {
      f16x8 xf[8];
      xload(xf, XB, kSC, l15, grp);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      gemm<S, P_WO, 2>(st, acc, xf);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (wave + t * kWaves) * 16 + 4 * grp;
        const f16x4 b4 = ld4(VT + tw.bo + n), r4 = ld4(XA + l15 * kSC + n);
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)((float)(f16)(acc[t][e] + (float)b4[e]) + (float)r4[e]);
        st4(XC + l15 * kSC + n, o);
      }
    }
    lds_barrier();
    DEC_STAMP(2);
    ln_rows<false>(XC, VT + tw.g1, VT + tw.e1, a.eps, wave, lane, [&](int row, int col, const float (&y)[4]) {
      const f16x4 p4 = ld4(QP + row * kSC + col);
      f16x4 x1, q2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x1[e] = (f16)y[e];
        q2[e] = (f16)((float)x1[e] + (float)p4[e]);
      }
      st4(XA + row * kSC + col, x1);
      st4(XB + row * kSC + col, q2);
    });
    lds_barrier();
    DEC_STAMP(3);
    // ================= (offsets | logits) projection =================
    {
      const int nt_ol = a.n_ol >> 4;
      f16x8 xf[8];
      xload(xf, XB, kSC, l15, grp);
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      gemm<S, P_WOL, 4>(st, acc, xf);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (wave + t * kWaves < nt_ol) {
          const int n = (wave + t * kWaves) * 16 + 4 * grp;
          const f16x4 b4 = ld4(VT + tw.bol + n);
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[t][e] + (float)b4[e]);
          st4(PJ + l15 * kS2 + n, o);
        }
    }
    lds_barrier();
    DEC_STAMP(4);
    // ================= MSDA: the 4 lanes of a (row, head) pair prepare its L*P points, then gather =================
    {
      const int pl = tid >> 2, sub = tid & 3, r = pl >> 3, m = pl & 7;
      const int gr = row0 + r < a.rows ? row0 + r : a.rows - 1;
      const int b = gr / a.Nq;
      const f16* pj = PJ + r * kS2;
      const float* s32 = RF + r * 4;
      const float* vr = s_vr + r * kMaxL * 2;
      constexpr int KMAX = kMaxLP / 4;
      float px[KMAX], py[KMAX], pw[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const int pt = sub + 4 * k;
        px[k] = py[k] = 0.f;
        pw[k] = -INFINITY;
        if (pt < LP) {
          const int col = m * LP + pt;
          px[k] = (float)pj[2 * col];
          py[k] = (float)pj[2 * col + 1];
          pw[k] = (float)pj[kM * LP * 2 + col];
        }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) mx = fmaxf(mx, pw[k]);
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        pw[k] = __expf(pw[k] - mx);
        sum += pw[k];
      }
      sum += __shfl_xor(sum, 2, 64);
      sum += __shfl_xor(sum, 1, 64);
      const float inv = 1.0f / sum;
      const unsigned row_bytes = kM * kD * sizeof(f16);
      const unsigned pair_base = (unsigned)b * (unsigned)a.S * row_bytes + (unsigned)m * (kD * sizeof(f16));
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const int pt = sub + 4 * k;
        if (pt >= LP) continue;
        const int l = pt / P;
        const int H = s_meta[4 * l], W = s_meta[4 * l + 1];
        const unsigned start = (unsigned)s_meta[4 * l + 2];
        const float Hf = (float)H, Wf = (float)W;
        // reference box on level l: sigmoid(ref) * valid ratio, fp32 (query_sine_embed.hip: ref_in32)
        const float rx = s32[0] * vr[2 * l], ry = s32[1] * vr[2 * l + 1];
        const float rw = s32[2] * vr[2 * l], rh = s32[3] * vr[2 * l + 1];
        float x = px[k] * (rw * (0.5f / (float)P)), y = py[k] * (rh * (0.5f / (float)P));
        x += rx;
        y += ry;
        const float aw = pw[k] * inv;
        const float h_im = fmaf(y, Hf, -0.5f);
        const float w_im = fmaf(x, Wf, -0.5f);
        const bool gate = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h0 = (int)hf, w0 = (int)wf;
        const float lh = h_im - hf, lw = w_im - wf;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const bool h0ok = h0 >= 0, w0ok = w0 >= 0, h1ok = h0 + 1 <= H - 1, w1ok = w0 + 1 <= W - 1;
        const float g_aw = gate ? aw : 0.f;
        Entry en;
        en.w[0] = (h0ok && w0ok) ? hh * hw * g_aw : 0.f;
        en.w[1] = (h0ok && w1ok) ? hh * lw * g_aw : 0.f;
        en.w[2] = (h1ok && w0ok) ? lh * hw * g_aw : 0.f;
        en.w[3] = (h1ok && w1ok) ? lh * lw * g_aw : 0.f;
        const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h0 + 1, 0), H - 1);
        const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w0 + 1, 0), W - 1);
        const unsigned base = pair_base + start * row_bytes;
        en.off[0] = base + (unsigned)(h0c * W + w0c) * row_bytes;
        en.off[1] = base + (unsigned)(h0c * W + w1c) * row_bytes;
        en.off[2] = base + (unsigned)(h1c * W + w0c) * row_bytes;
        en.off[3] = base + (unsigned)(h1c * W + w1c) * row_bytes;
        entries[pt * 128 + pl] = en;
      }
      lds_barrier();
      DEC_STAMP(5);
      const unsigned lane_byte = (unsigned)sub * 16;
      const unsigned char* vbase = reinterpret_cast<const unsigned char*>(a.value);
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      constexpr int GROUP = 10;   // corner rows of ten points in flight per lane (the weight stream is run down here)
      const Entry* my = entries + pl;
      int i = (CODETR_DEC_ABL & 1) ? LP : 0;
      for (; i < LP; i += GROUP) {
        f16x8 raw[GROUP][4];
#pragma unroll
        for (int g = 0; g < GROUP; ++g)
          if (i + g < LP) {
            const u32x4 off = my[(i + g) * 128].off;
#pragma unroll
            for (int k = 0; k < 4; ++k) raw[g][k] = *reinterpret_cast<const f16x8*>(vbase + (size_t)(off[k] + lane_byte));
          }
#pragma unroll
        for (int g = 0; g < GROUP; ++g)
          if (i + g < LP) {
            const f32x4 w = my[(i + g) * 128].w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float wk = w[k];
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(wk, (float)raw[g][k][j], acc[j]);
            }
          }
      }
      f16x8 packed;
#pragma unroll
      for (int j = 0; j < 8; ++j) packed[j] = (f16)acc[j];
      *reinterpret_cast<f16x8*>(XB + r * kSC + m * kD + sub * 8) = packed;
    }
    lds_barrier();
    DEC_STAMP(6);
    // ================= output projection + identity, LN2 =================
    {
      f16x8 xf[8];
      xload(xf, XB, kSC, l15, grp);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      gemm<S, P_WOUT, 2>(st, acc, xf);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (wave + t * kWaves) * 16 + 4 * grp;
        const f16x4 b4 = ld4(VT + tw.bout + n), r4 = ld4(XA + l15 * kSC + n);
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)((float)(f16)(acc[t][e] + (float)b4[e]) + (float)r4[e]);
        st4(XC + l15 * kSC + n, o);
      }
    }
    lds_barrier();
    ln_rows<false>(XC, VT + tw.g2, VT + tw.e2, a.eps, wave, lane, [&](int row, int col, const float (&y)[4]) {
      f16x4 x2;
#pragma unroll
      for (int e = 0; e < 4; ++e) x2[e] = (f16)y[e];
      st4(XA + row * kSC + col, x2);
    });
    lds_barrier();
    DEC_STAMP(7);
    // ================= FFN: hidden chunks of 256, Y accumulated in registers, LN3 =================
    {
      f16x8 xf[8];
      xload(xf, XA, kSC, l15, grp);
      f32x4 yacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      auto chunk = [&](auto cc_) {
        constexpr int c = decltype(cc_)::value;
        f16* hb = HB + (c & 1) * kRows * kSC;
        f32x4 hacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        gemm<S, P_FFN + 2 * c, 2>(st, hacc, xf);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int n = (wave + t * kWaves) * 16 + 4 * grp;
          const f16x4 b4 = ld4(VT + tw.b1 + ((c + st.rot) & (kF / 256 - 1)) * 256 + n);
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (f16)fmaxf(hacc[t][e] + (float)b4[e], 0.f);
          st4(hb + l15 * kSC + n, o);
        }
        lds_barrier();
        f16x8 hf[8];
        xload(hf, hb, kSC, l15, grp);
        gemm<S, P_FFN + 2 * c + 1, 2>(st, yacc, hf);
      };
      chunk(std::integral_constant<int, 0>{});
      chunk(std::integral_constant<int, 1>{});
      chunk(std::integral_constant<int, 2>{});
      chunk(std::integral_constant<int, 3>{});
      chunk(std::integral_constant<int, 4>{});
      chunk(std::integral_constant<int, 5>{});
      chunk(std::integral_constant<int, 6>{});
      chunk(std::integral_constant<int, 7>{});
      static_assert(kF / 256 == 8, "the chunk list above");
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (wave + t * kWaves) * 16 + 4 * grp;
        const f16x4 b4 = ld4(VT + tw.b2 + n), r4 = ld4(XA + l15 * kSC + n);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = yacc[t][e] + (float)b4[e] + (float)r4[e];
        *reinterpret_cast<f32x4*>(YF + l15 * kSF + n) = o;
      }
    }
    lds_barrier();
    DEC_STAMP(8);
    ln_rows<true>(YF, VT + tw.g3, VT + tw.e3, a.eps, wave, lane, [&](int row, int col, const float (&y)[4]) {
      f16x4 x3;
#pragma unroll
      for (int e = 0; e < 4; ++e) x3[e] = (f16)y[e];
      st4(XA + row * kSC + col, x3);
      if (HEAD && row0 + row < a.rows) st4(a.x_out + (size_t)(row0 + row) * kC + col, x3);
    });
    lds_barrier();
    if (!HEAD) {
      // the decoder's output norm on the (rounded) last layer output
      ln_rows<false>(XA, VF, VF + kC, a.eps, wave, lane, [&](int row, int col, const float (&y)[4]) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)y[e];
        if (row0 + row < a.rows) st4(a.x_out + (size_t)(row0 + row) * kC + col, o);
      });
    }
    DEC_STAMP(9);
    // ================= box refinement: ref' = ref + reg_branch(x3) =================
    {
      f16x8 xf[8];
      xload(xf, XA, kSC, l15, grp);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      gemm<S, P_WR1, 2>(st, acc, xf);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (wave + t * kWaves) * 16 + 4 * grp;
        const f16x4 b4 = ld4(VT + tw.br1 + n);
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)fmaxf(acc[t][e] + (float)b4[e], 0.f);
        st4(XB + l15 * kSC + n, o);
      }
      lds_barrier();
      xload(xf, XB, kSC, l15, grp);
      acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      gemm<S, P_WR2, 2>(st, acc, xf);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (wave + t * kWaves) * 16 + 4 * grp;
        const f16x4 b4 = ld4(VT + tw.br2 + n);
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)fmaxf(acc[t][e] + (float)b4[e], 0.f);
        st4(XC + l15 * kSC + n, o);
      }
      lds_barrier();
      xload(xf, XC, kSC, l15, grp);
      f32x4 d[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      gemm<S, P_WR3, 1>(st, d, xf);   // (every wave walks the stream; wave 0 holds the 4 real rows)
      if (wave == 0 && grp == 0) {    // lane (row l15, group 0) holds the row's 4 box deltas
        const f16x4 b4 = ld4(VT + tw.br3);
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (f16)((float)(f16)(d[0][e] + (float)b4[e]) + RF[kRows * 4 + l15 * 4 + e]);
          RF[kRows * 4 + l15 * 4 + e] = (float)o[e];
        }
        if (mrow_ok) st4(a.ref_out + (size_t)mrow * 4, o);
      }
    }
    lds_barrier();
  }
  if constexpr (HEAD) {

  DEC_STAMP(10);
  // ================= HEAD of the next layer =================
  // sine embedding of the level-0 reference box (query_sine_embed.hip), fp32 trigonometry: [16][512] -> PJ
  {
    const int F = 2 * kC / 4;   // pos_feat = embed_dims / 2 = 128 channels per coordinate
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + it * kThreads;          // 16 rows x 64 chunks of 8 channels
      const int r = item >> 6, c = item & 63;
      const int j = (c * 8) / F;
      const int coord = j == 0 ? 1 : (j == 1 ? 0 : j);
      const float v = RF[kRows * 4 + r * 4 + coord];
      const float s = 1.0f / (1.0f + expf(-v));
      const float v0 = s * s_vr[r * kMaxL * 2 + (coord & 1)];
      const float e = v0 * 6.283185307179586f;
      const int ch0 = c * 8 - j * F;
      f16x8 o;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int f = (ch0 >> 1) + p;
        const float rev = e * __builtin_amdgcn_exp2f(-a.log2_temperature * (2.0f * (float)f / (float)F)) * 0.15915494309189535f;
        o[2 * p] = (f16)__builtin_amdgcn_sinf(rev);
        o[2 * p + 1] = (f16)__builtin_amdgcn_cosf(rev);
      }
      *reinterpret_cast<f16x8*>(PJ + r * kS2 + c * 8) = o;
    }
  }
  lds_barrier();
  DEC_STAMP(11);
  {
    f16x8 xf[8];
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    xload(xf, PJ, kS2, l15, grp);
    gemm<S, HB_ + 0, 2>(st, acc, xf);
    xload(xf, PJ + kC, kS2, l15, grp);
    gemm<S, HB_ + 1, 2>(st, acc, xf);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = (wave + t * kWaves) * 16 + 4 * grp;
      const f16x4 b4 = ld4(VP + kPosB1 + n);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)fmaxf(acc[t][e] + (float)b4[e], 0.f);
      st4(XB + l15 * kSC + n, o);
    }
    lds_barrier();
    xload(xf, XB, kSC, l15, grp);
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm<S, HB_ + 2, 2>(st, acc, xf);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = (wave + t * kWaves) * 16 + 4 * grp;
      const f16x4 b4 = ld4(VP + kPosB2 + n), x4 = ld4(XA + l15 * kSC + n);
      f16x4 qp, q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qp[e] = (f16)(acc[t][e] + (float)b4[e]);
        q[e] = (f16)((float)x4[e] + (float)qp[e]);
      }
      st4(XD + l15 * kSC + n, q);
      if (mrow_ok) st4(a.qpos_out + (size_t)mrow * kC + n, qp);
    }
  }
  lds_barrier();
  DEC_STAMP(12);
  // in-projections of the next self-attention: [q | k] from x + qpos, v from x
  {
    f16x8 xf[8];
    xload(xf, XD, kSC, l15, grp);
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm<S, HB_ + 3, 4>(st, acc, xf);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = (wave + t * kWaves) * 16 + 4 * grp;
      const f16x4 b4 = ld4(VH + kHeadBqk + n);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[t][e] + (float)b4[e]);
      if (mrow_ok) st4(a.qk_out + (size_t)mrow * (2 * kC) + n, o);
    }
    xload(xf, XA, kSC, l15, grp);
    f32x4 vacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    gemm<S, HB_ + 4, 2>(st, vacc, xf);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = (wave + t * kWaves) * 16 + 4 * grp;
      const f16x4 b4 = ld4(VH + kHeadBv + n);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)(vacc[t][e] + (float)b4[e]);
      if (mrow_ok) st4(a.v_out + (size_t)mrow * kC + n, o);
    }
  }
  DEC_STAMP(13);
  }
}

size_t lds_bytes(int LP) {
  const size_t fixed = (size_t)(5 * kRows * kSC + kRows * kS2 + kMaxTailVec + kHeadNVec + kPosNVec + 2 * kC) * sizeof(f16) +
                       2 * kRows * 4 * sizeof(float) + kMaxL * 4 * sizeof(int) + kRows * kMaxL * 2 * sizeof(float);
  const size_t ent = (size_t)LP * 128 * sizeof(Entry);
  const size_t ffn = 2 * kRows * kSC * sizeof(f16) + kRows * kSF * sizeof(float);
  return fixed + (ent > ffn ? ent : ffn);
}

bool dims_ok(int num_heads, int head_dim, int L, int P, int hidden, int ref_dim, int pos_feat) {
  return num_heads == kM && head_dim == kD && L >= 1 && L <= kMaxL && P >= 1 && L * P <= kMaxLP && ref_dim == 4 &&
         pos_feat == kC / 2 && hidden == kF && (kM * L * P * 3) % 16 == 0 && kM * L * P * 3 <= 512 &&
         lds_bytes(L * P) <= 160 * 1024;   // (entries of every sample point of 128 (row, head) pairs live in LDS)
}

}  // namespace

extern "C" {

#ifndef CODETR_DEC_BF16
#ifdef CODETR_DEC_STAMPS
int codetr_decoder_layer_debug_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dec_stamps), sizeof(unsigned long long) * 32);
}
#endif

int codetr_decoder_layer_supported(int embed_dims, int num_heads, int num_levels, int num_points, int hidden, int ref_dim,
                                   int pos_feat) {
  return embed_dims == kC && dims_ok(num_heads, embed_dims / (num_heads > 0 ? num_heads : 1), num_levels, num_points, hidden,
                                     ref_dim, pos_feat)
             ? 1
             : 0;
}

int64_t codetr_decoder_layer_blob_halfs(int which, int num_levels, int num_points, int hidden) {
  switch (which) {
    case 0: return hidden == kF ? tail_layout(kM * num_levels * num_points * 3).total : CODETR_E_UNSUPPORTED;
    case 1: return kHeadTotal;
    case 2: return kPosTotal;
    case 3: return 2 * kC;
    default: return CODETR_E_BADARG;
  }
}

#endif   // !CODETR_DEC_BF16

int CODETR_DEC_ENTRY(void* stream, const void* x_dev, const void* attn_dev, const void* qpos_dev, const void* ref_dev,
                             const float* valid_ratios32_dev, const void* value_dev, const int64_t* spatial_shapes_dev,
                             const int64_t* level_start_dev, const void* tail_w_dev, const void* pos_w_dev,
                             const void* head_w_dev, const void* final_norm_dev, void* x_out_dev, void* ref_out_dev,
                             void* qpos_out_dev, void* qk_out_dev, void* v_out_dev, int64_t B, int64_t Nq, int64_t S,
                             int num_levels, int num_points, int hidden, float ln_eps, float temperature) {
  const bool tail = attn_dev != nullptr, head = head_w_dev != nullptr;
  if (!x_dev || !ref_dev || !valid_ratios32_dev || B <= 0 || Nq <= 0 || temperature <= 0.f) return CODETR_E_BADARG;
  if (!tail && !head) return CODETR_E_BADARG;
  if (tail && (!qpos_dev || !value_dev || !spatial_shapes_dev || !level_start_dev || !tail_w_dev || !x_out_dev ||
               !ref_out_dev || S <= 0))
    return CODETR_E_BADARG;
  if (head && (!pos_w_dev || !qpos_out_dev || !qk_out_dev || !v_out_dev)) return CODETR_E_BADARG;
  if (tail && !head && !final_norm_dev) return CODETR_E_BADARG;
  if (!dims_ok(kM, kD, num_levels, num_points, hidden, 4, kC / 2)) return CODETR_E_UNSUPPORTED;
  if (B * Nq > 0x7fffffffLL) return CODETR_E_TOO_LARGE;
  if (tail && (double)B * (double)S * (kM * kD * 2) > 4294967295.0) return CODETR_E_TOO_LARGE;   // 32-bit value offsets
  const void* ptrs[] = {x_dev, attn_dev, qpos_dev, value_dev, tail_w_dev, pos_w_dev, head_w_dev, final_norm_dev,
                        x_out_dev, qpos_out_dev, qk_out_dev, v_out_dev};
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 15) return CODETR_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(ref_dev) | reinterpret_cast<uintptr_t>(ref_out_dev)) & 7) return CODETR_E_BADARG;
  DecArgs a{};
  a.x = static_cast<const f16*>(x_dev);
  a.attn = static_cast<const f16*>(attn_dev);
  a.qpos = static_cast<const f16*>(qpos_dev);
  a.ref = static_cast<const f16*>(ref_dev);
  a.vr32 = valid_ratios32_dev;
  a.value = static_cast<const f16*>(value_dev);
  a.shapes = spatial_shapes_dev;
  a.starts = level_start_dev;
  a.tail_w = static_cast<const f16*>(tail_w_dev);
  a.pos_w = static_cast<const f16*>(pos_w_dev);
  a.head_w = static_cast<const f16*>(head_w_dev);
  a.final_norm = static_cast<const f16*>(final_norm_dev);
  a.x_out = static_cast<f16*>(x_out_dev);
  a.ref_out = static_cast<f16*>(ref_out_dev);
  a.qpos_out = static_cast<f16*>(qpos_out_dev);
  a.qk_out = static_cast<f16*>(qk_out_dev);
  a.v_out = static_cast<f16*>(v_out_dev);
  a.rows = (int)(B * Nq);
  a.Nq = (int)Nq;
  a.S = (int)S;
  a.L = num_levels;
  a.P = num_points;
  a.n_ol = kM * num_levels * num_points * 3;
  a.eps = ln_eps;
  a.log2_temperature = log2f(temperature);
  a.tw = tail_layout(a.n_ol);
  const size_t lds = lds_bytes(num_levels * num_points);
  const void* kfn = tail ? (head ? reinterpret_cast<const void*>(decoder_layer_kernel<true, true>)
                                : reinterpret_cast<const void*>(decoder_layer_kernel<true, false>))
                         : reinterpret_cast<const void*>(decoder_layer_kernel<false, true>);
  {
    static std::atomic<uint32_t> done[64];   // bit = instantiation, index = device ordinal
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0, done[0].store(0);
    const uint32_t bit = tail ? (head ? 1u : 2u) : 4u;
    if (!(done[dev].load(std::memory_order_acquire) & bit)) {
      const hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      done[dev].fetch_or(bit, std::memory_order_release);
    }
  }
  const unsigned blocks = (unsigned)((a.rows + kRows - 1) / kRows);
  if (tail && head)
    hipLaunchKernelGGL((decoder_layer_kernel<true, true>), dim3(blocks), dim3(kThreads), lds, static_cast<hipStream_t>(stream), a);
  else if (tail)
    hipLaunchKernelGGL((decoder_layer_kernel<true, false>), dim3(blocks), dim3(kThreads), lds, static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL((decoder_layer_kernel<false, true>), dim3(blocks), dim3(kThreads), lds, static_cast<hipStream_t>(stream), a);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // extern "C"
