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

#include "codetr_hip.h"

// diagnostic builds only (-DCODETR_DEC_ABL=mask; WRONG results by construction, never shipped): 1 = no MSDA gather,
// 2 = no FFN chunks, 4 = no head part, 8 = no box refinement
#ifndef CODETR_DEC_ABL
#define CODETR_DEC_ABL 0
#endif

namespace {

constexpr int kC = 256;          // embed dims
constexpr int kM = 8;            // heads
constexpr int kD = 32;           // head dim
constexpr int kRows = 16;        // query rows per workgroup
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kSC = kC + 8;      // LDS row stride (halfs) of a [16][256] fp16 buffer: 528 B, conflict-free fragment reads
constexpr int kS2 = 2 * kC + 8;  // ... of a [16][512] buffer
constexpr int kSF = kC + 4;      // fp32 row stride of the [16][256] pre-LayerNorm buffer
constexpr int kMaxLP = 32;
constexpr int kMaxL = 8;

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// element offsets into the packed weight blobs (include/codetr_hip.h documents the order)
struct TailW {
  int wo, bo, g1, e1, wol, bol, wout, bout, g2, e2, w1, b1, w2, b2, g3, e3, wr1, br1, wr2, br2, wr3, br3, total;
};
__host__ __device__ inline TailW tail_layout(int n_ol, int F) {
  TailW t;
  int o = 0;
  auto take = [&](int n) { const int at = o; o += n; return at; };
  t.wo = take(kC * kC); t.bo = take(kC); t.g1 = take(kC); t.e1 = take(kC);
  t.wol = take(n_ol * kC); t.bol = take(n_ol);
  t.wout = take(kC * kC); t.bout = take(kC); t.g2 = take(kC); t.e2 = take(kC);
  t.w1 = take(F * kC); t.b1 = take(F); t.w2 = take(kC * F); t.b2 = take(kC); t.g3 = take(kC); t.e3 = take(kC);
  t.wr1 = take(kC * kC); t.br1 = take(kC); t.wr2 = take(kC * kC); t.br2 = take(kC);
  t.wr3 = take(4 * kC); t.br3 = take(8);
  t.total = o;
  return t;
}
constexpr int kHeadWqk = 0, kHeadBqk = 2 * kC * kC, kHeadWv = kHeadBqk + 2 * kC, kHeadBv = kHeadWv + kC * kC,
              kHeadTotal = kHeadBv + kC;
constexpr int kPosW1 = 0, kPosB1 = kC * 2 * kC, kPosW2 = kPosB1 + kC, kPosB2 = kPosW2 + kC * kC, kPosTotal = kPosB2 + kC;

struct DecArgs {
  const f16* x; const f16* attn; const f16* qpos; const f16* ref; const float* vr32; const f16* value;
  const int64_t* shapes; const int64_t* starts;
  const f16* tail_w; const f16* pos_w; const f16* head_w; const f16* final_norm;
  f16* x_out; f16* ref_out; f16* qpos_out; f16* qk_out; f16* v_out;
  int rows, Nq, S, L, P, F, n_ol;
  float eps, log2_temperature;
  TailW tw;
};

struct Entry {
  u32x4 off;
  f32x4 w;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// W fragments of up to NT 16-row tiles (tile0, tile0 + 8, ...) x KS k-steps, straight from global memory.
// rows beyond `nrows` are clamped (their outputs are never used).  Tiles beyond `ntiles` are not touched.
template <int NT, int KS>
__device__ __forceinline__ void wload(f16x8 (&a)[NT][KS], const f16* __restrict__ W, const int K, const int kcol0,
                                      const int tile0, const int ntiles, const int nrows, const int l15, const int grp) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tile = tile0 + t * kWaves;
    if (tile < ntiles) {
      int row = tile * 16 + l15;
      row = row < nrows ? row : nrows - 1;
      const f16* p = W + (size_t)row * K + kcol0 + grp * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[t][ks] = *reinterpret_cast<const f16x8*>(p + ks * 32);
    }
  }
}

// the 16 activation rows of an LDS buffer as MFMA B fragments: lane (row l15, group grp) holds X[row][32 ks + 8 grp ..]
template <int KS>
__device__ __forceinline__ void xload(f16x8 (&xf)[KS], const f16* xs, const int stride, const int l15, const int grp) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const f16x8*>(xs + l15 * stride + ks * 32 + grp * 8);
}

template <int NT, int KS>
__device__ __forceinline__ void wmma(f32x4 (&acc)[NT], const f16x8 (&a)[NT][KS], const f16x8 (&xf)[KS], const int tile0,
                                     const int ntiles) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (tile0 + t * kWaves < ntiles) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc[t] = mfma16(a[t][ks], xf[ks], acc[t]);
    }
}

__device__ __forceinline__ f16x4 ld4(const f16* p) { return *reinterpret_cast<const f16x4*>(p); }
__device__ __forceinline__ void st4(f16* p, f16x4 v) { *reinterpret_cast<f16x4*>(p) = v; }

// LayerNorm of rows 2 * wave, 2 * wave + 1 of an fp16 (SRC32 = false) or fp32 [16][256] LDS buffer; gamma / beta
// from global memory.  Two-pass statistics in fp32 as csrc/layernorm.hip.  `emit(row, col, y[4])` receives the result.
template <bool SRC32, class Emit>
__device__ __forceinline__ void ln_rows(const void* src, const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                        const float eps, const int wave, const int lane, Emit emit) {
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

__global__ __launch_bounds__(kThreads) void decoder_layer_kernel(const DecArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // ---- LDS map ----
  f16* XA = reinterpret_cast<f16*>(smem);                 // [16][kSC]  current activations (x, x1, x2, x3)
  f16* XB = XA + kRows * kSC;                             // [16][kSC]
  f16* XC = XB + kRows * kSC;                             // [16][kSC]
  f16* XD = XC + kRows * kSC;                             // [16][kSC]
  f16* PJ = XD + kRows * kSC;                             // [16][kS2]  (offsets | logits) of the rows; later the sine embedding
  float* RF = reinterpret_cast<float*>(PJ + kRows * kS2); // [16][4] sigmoid(ref) fp32, then [16][4] of the refined boxes
  int* s_meta = reinterpret_cast<int*>(RF + 2 * kRows * 4);   // [kMaxL][4] level table (H, W, start, -)
  float* s_vr = reinterpret_cast<float*>(s_meta + kMaxL * 4); // [16 rows][kMaxL][2] valid ratios of each row's image
  unsigned char* EB = reinterpret_cast<unsigned char*>(s_vr + kRows * kMaxL * 2);   // big region: entries | hidden | fp32 rows
  Entry* entries = reinterpret_cast<Entry*>(EB);          // [LP][128 pairs]
  f16* HB = reinterpret_cast<f16*>(EB);                   // [2][16][kSC] hidden chunks of the FFN
  float* YF = reinterpret_cast<float*>(EB + 2 * kRows * kSC * sizeof(f16));   // [16][kSF] fp32 pre-LayerNorm rows

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, grp = lane >> 4;
  const int row0 = blockIdx.x * kRows;
  const bool tail = a.attn != nullptr, head = a.head_w != nullptr;
  const int L = a.L, P = a.P, LP = L * P;
  // global row of the lane's MFMA output row / of a 32-thread row group (clamped: the last workgroup recomputes the
  // last valid row, its stores are masked)
  const int mrow = row0 + l15 < a.rows ? row0 + l15 : a.rows - 1;
  const bool mrow_ok = row0 + l15 < a.rows;
  const int cr = tid >> 5, cc = (tid & 31) * 8;   // row-copy role: 32 threads x 16 B per row
  const int crow = row0 + cr < a.rows ? row0 + cr : a.rows - 1;

  // ---- level table, valid ratios, reference boxes ----
  if (tail && tid < L) {
    s_meta[4 * tid] = (int)a.shapes[2 * tid];
    s_meta[4 * tid + 1] = (int)a.shapes[2 * tid + 1];
    s_meta[4 * tid + 2] = (int)a.starts[tid];
    s_meta[4 * tid + 3] = 0;
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
    RF[kRows * 4 + tid] = v;                 // unactivated, for the head-only launch
  }

  if (tail) {
    const f16* TW = a.tail_w;
    const TailW& tw = a.tw;
    // ================= out-projection of the self-attention + identity, LN1 =================
    f16x8 wA[2][8];
    wload<2, 8>(wA, TW + tw.wo, kC, 0, wave, 16, kC, l15, grp);
    *reinterpret_cast<f16x8*>(XA + cr * kSC + cc) = *reinterpret_cast<const f16x8*>(a.x + (size_t)crow * kC + cc);
    *reinterpret_cast<f16x8*>(XB + cr * kSC + cc) = *reinterpret_cast<const f16x8*>(a.attn + (size_t)crow * kC + cc);
    __syncthreads();
    {
      f16x8 xf[8];
      xload<8>(xf, XB, kSC, l15, grp);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      wmma<2, 8>(acc, wA, xf, wave, 16);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (wave + t * kWaves) * 16 + 4 * grp;
        const f16x4 b4 = ld4(TW + tw.bo + n), r4 = ld4(XA + l15 * kSC + n);
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)((float)(f16)(acc[t][e] + (float)b4[e]) + (float)r4[e]);
        st4(XC + l15 * kSC + n, o);
      }
    }
    // (offsets | logits) weights: requested now, used after LN1
    const int nt_ol = a.n_ol >> 4;
    f16x8 wB[4][8];
    wload<4, 8>(wB, TW + tw.wol, kC, 0, wave, nt_ol, a.n_ol, l15, grp);
    __syncthreads();
    ln_rows<false>(XC, TW + tw.g1, TW + tw.e1, a.eps, wave, lane, [&](int row, int col, const float (&y)[4]) {
      const int gr = row0 + row < a.rows ? row0 + row : a.rows - 1;
      const f16x4 p4 = ld4(a.qpos + (size_t)gr * kC + col);
      f16x4 x1, q2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x1[e] = (f16)y[e];
        q2[e] = (f16)((float)x1[e] + (float)p4[e]);
      }
      st4(XA + row * kSC + col, x1);
      st4(XB + row * kSC + col, q2);
    });
    __syncthreads();
    // ================= (offsets | logits) projection =================
    {
      f16x8 xf[8];
      xload<8>(xf, XB, kSC, l15, grp);
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      wmma<4, 8>(acc, wB, xf, wave, nt_ol);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (wave + t * kWaves < nt_ol) {
          const int n = (wave + t * kWaves) * 16 + 4 * grp;
          const f16x4 b4 = ld4(TW + tw.bol + n);
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[t][e] + (float)b4[e]);
          st4(PJ + l15 * kS2 + n, o);
        }
    }
    // output projection of the cross-attention: requested now
    wload<2, 8>(wA, TW + tw.wout, kC, 0, wave, 16, kC, l15, grp);
    __syncthreads();
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
      __syncthreads();
      const unsigned lane_byte = (unsigned)sub * 16;
      const unsigned char* vbase = reinterpret_cast<const unsigned char*>(a.value);
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      constexpr int GROUP = 4;
      const Entry* my = entries + pl;
      int i = (CODETR_DEC_ABL & 1) ? LP : 0;
      for (; i + GROUP <= LP; i += GROUP) {
        Entry en[GROUP];
        f16x8 raw[GROUP][4];
#pragma unroll
        for (int g = 0; g < GROUP; ++g) en[g] = my[(i + g) * 128];
#pragma unroll
        for (int g = 0; g < GROUP; ++g)
#pragma unroll
          for (int k = 0; k < 4; ++k) raw[g][k] = *reinterpret_cast<const f16x8*>(vbase + (size_t)(en[g].off[k] + lane_byte));
#pragma unroll
        for (int g = 0; g < GROUP; ++g)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float wk = en[g].w[k];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(wk, (float)raw[g][k][j], acc[j]);
          }
      }
      for (; i < LP; ++i) {
        const Entry en = my[i * 128];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f16x8 raw = *reinterpret_cast<const f16x8*>(vbase + (size_t)(en.off[k] + lane_byte));
          const float wk = en.w[k];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(wk, (float)raw[j], acc[j]);
        }
      }
      f16x8 packed;
#pragma unroll
      for (int j = 0; j < 8; ++j) packed[j] = (f16)acc[j];
      *reinterpret_cast<f16x8*>(XB + r * kSC + m * kD + sub * 8) = packed;
    }
    __syncthreads();
    // ================= output projection + identity, LN2 =================
    {
      f16x8 xf[8];
      xload<8>(xf, XB, kSC, l15, grp);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      wmma<2, 8>(acc, wA, xf, wave, 16);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (wave + t * kWaves) * 16 + 4 * grp;
        const f16x4 b4 = ld4(TW + tw.bout + n), r4 = ld4(XA + l15 * kSC + n);
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)((float)(f16)(acc[t][e] + (float)b4[e]) + (float)r4[e]);
        st4(XC + l15 * kSC + n, o);
      }
    }
    // first hidden chunk of the FFN: requested now
    f16x8 w1f[2][8];
    wload<2, 8>(w1f, TW + tw.w1, kC, 0, wave, 16, kC, l15, grp);
    __syncthreads();
    ln_rows<false>(XC, TW + tw.g2, TW + tw.e2, a.eps, wave, lane, [&](int row, int col, const float (&y)[4]) {
      f16x4 x2;
#pragma unroll
      for (int e = 0; e < 4; ++e) x2[e] = (f16)y[e];
      st4(XA + row * kSC + col, x2);
    });
    __syncthreads();
    // ================= FFN: hidden chunks of 256, Y accumulated in registers, LN3 =================
    {
      f16x8 xf[8];
      xload<8>(xf, XA, kSC, l15, grp);
      f32x4 yacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      const int nchunks = (CODETR_DEC_ABL & 2) ? 0 : a.F >> 8;
      for (int c = 0; c < nchunks; ++c) {
        f16* hb = HB + (c & 1) * kRows * kSC;
        f32x4 hacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        wmma<2, 8>(hacc, w1f, xf, wave, 16);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int n = (wave + t * kWaves) * 16 + 4 * grp;
          const f16x4 b4 = ld4(TW + tw.b1 + c * 256 + n);
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (f16)fmaxf(hacc[t][e] + (float)b4[e], 0.f);
          st4(hb + l15 * kSC + n, o);
        }
        f16x8 w2f[2][8];
        wload<2, 8>(w2f, TW + tw.w2, a.F, c * 256, wave, 16, kC, l15, grp);
        __syncthreads();
        if (c + 1 < nchunks) wload<2, 8>(w1f, TW + tw.w1 + (size_t)(c + 1) * 256 * kC, kC, 0, wave, 16, kC, l15, grp);
        f16x8 hf[8];
        xload<8>(hf, hb, kSC, l15, grp);
        wmma<2, 8>(yacc, w2f, hf, wave, 16);
      }
      // reg branch layer 1: requested now
      wload<2, 8>(wA, TW + tw.wr1, kC, 0, wave, 16, kC, l15, grp);
      __syncthreads();   // (every wave is done with the hidden chunks: YF overlaps nothing of HB, but keep the phases apart)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (wave + t * kWaves) * 16 + 4 * grp;
        const f16x4 b4 = ld4(TW + tw.b2 + n), r4 = ld4(XA + l15 * kSC + n);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = yacc[t][e] + (float)b4[e] + (float)r4[e];
        *reinterpret_cast<f32x4*>(YF + l15 * kSF + n) = o;
      }
    }
    __syncthreads();
    ln_rows<true>(YF, TW + tw.g3, TW + tw.e3, a.eps, wave, lane, [&](int row, int col, const float (&y)[4]) {
      f16x4 x3;
#pragma unroll
      for (int e = 0; e < 4; ++e) x3[e] = (f16)y[e];
      st4(XA + row * kSC + col, x3);
      if (head && row0 + row < a.rows) st4(a.x_out + (size_t)(row0 + row) * kC + col, x3);
    });
    __syncthreads();
    if (!head) {
      // the decoder's output norm on the (rounded) last layer output
      ln_rows<false>(XA, a.final_norm, a.final_norm + kC, a.eps, wave, lane, [&](int row, int col, const float (&y)[4]) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)y[e];
        if (row0 + row < a.rows) st4(a.x_out + (size_t)(row0 + row) * kC + col, o);
      });
    }
    // ================= box refinement: ref' = ref + reg_branch(x3) =================
    if (!(CODETR_DEC_ABL & 8)) {
      f16x8 xf[8];
      xload<8>(xf, XA, kSC, l15, grp);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      wmma<2, 8>(acc, wA, xf, wave, 16);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (wave + t * kWaves) * 16 + 4 * grp;
        const f16x4 b4 = ld4(TW + tw.br1 + n);
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)fmaxf(acc[t][e] + (float)b4[e], 0.f);
        st4(XB + l15 * kSC + n, o);
      }
      wload<2, 8>(wA, TW + tw.wr2, kC, 0, wave, 16, kC, l15, grp);
      __syncthreads();
      xload<8>(xf, XB, kSC, l15, grp);
      acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      wmma<2, 8>(acc, wA, xf, wave, 16);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (wave + t * kWaves) * 16 + 4 * grp;
        const f16x4 b4 = ld4(TW + tw.br2 + n);
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)fmaxf(acc[t][e] + (float)b4[e], 0.f);
        st4(XC + l15 * kSC + n, o);
      }
      f16x8 w3[1][8];
      if (wave == 0) wload<1, 8>(w3, TW + tw.wr3, kC, 0, 0, 1, 4, l15, grp);
      __syncthreads();
      if (wave == 0) {
        xload<8>(xf, XC, kSC, l15, grp);
        f32x4 d[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        wmma<1, 8>(d, w3, xf, 0, 1);
        if (grp == 0) {   // lane (row l15, group 0) holds the row's 4 box deltas
          const f16x4 b4 = ld4(TW + tw.br3);
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (f16)((float)(f16)(d[0][e] + (float)b4[e]) + RF[kRows * 4 + l15 * 4 + e]);
            RF[kRows * 4 + l15 * 4 + e] = (float)o[e];
          }
          if (mrow_ok) st4(a.ref_out + (size_t)mrow * 4, o);
        }
      }
    }
    __syncthreads();
  } else {
    // head-only launch: x is the decoder's input query
    *reinterpret_cast<f16x8*>(XA + cr * kSC + cc) = *reinterpret_cast<const f16x8*>(a.x + (size_t)crow * kC + cc);
    __syncthreads();
  }
  if (!head || (CODETR_DEC_ABL & 4)) return;

  // ================= HEAD of the next layer =================
  const f16* PW = a.pos_w;
  const f16* HW = a.head_w;
  // first half of ref_point_head's first layer (K = 512 in two halves of 256): requested now
  f16x8 wP[2][8];
  wload<2, 8>(wP, PW + kPosW1, 2 * kC, 0, wave, 16, kC, l15, grp);
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
  __syncthreads();
  {
    f16x8 xf[8];
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    xload<8>(xf, PJ, kS2, l15, grp);
    wmma<2, 8>(acc, wP, xf, wave, 16);
    wload<2, 8>(wP, PW + kPosW1, 2 * kC, kC, wave, 16, kC, l15, grp);
    xload<8>(xf, PJ + kC, kS2, l15, grp);
    wmma<2, 8>(acc, wP, xf, wave, 16);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = (wave + t * kWaves) * 16 + 4 * grp;
      const f16x4 b4 = ld4(PW + kPosB1 + n);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)fmaxf(acc[t][e] + (float)b4[e], 0.f);
      st4(XB + l15 * kSC + n, o);
    }
    wload<2, 8>(wP, PW + kPosW2, kC, 0, wave, 16, kC, l15, grp);
    __syncthreads();
    xload<8>(xf, XB, kSC, l15, grp);
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    wmma<2, 8>(acc, wP, xf, wave, 16);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = (wave + t * kWaves) * 16 + 4 * grp;
      const f16x4 b4 = ld4(PW + kPosB2 + n), x4 = ld4(XA + l15 * kSC + n);
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
  // in-projections of the next self-attention: [q | k] from x + qpos, v from x
  f16x8 wQ[4][8];
  wload<4, 8>(wQ, HW + kHeadWqk, kC, 0, wave, 32, 2 * kC, l15, grp);
  __syncthreads();
  {
    f16x8 xf[8];
    xload<8>(xf, XD, kSC, l15, grp);
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    wmma<4, 8>(acc, wQ, xf, wave, 32);
    wload<2, 8>(wP, HW + kHeadWv, kC, 0, wave, 16, kC, l15, grp);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = (wave + t * kWaves) * 16 + 4 * grp;
      const f16x4 b4 = ld4(HW + kHeadBqk + n);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[t][e] + (float)b4[e]);
      if (mrow_ok) st4(a.qk_out + (size_t)mrow * (2 * kC) + n, o);
    }
    xload<8>(xf, XA, kSC, l15, grp);
    f32x4 vacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    wmma<2, 8>(vacc, wP, xf, wave, 16);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = (wave + t * kWaves) * 16 + 4 * grp;
      const f16x4 b4 = ld4(HW + kHeadBv + n);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)(vacc[t][e] + (float)b4[e]);
      if (mrow_ok) st4(a.v_out + (size_t)mrow * kC + n, o);
    }
  }
}

size_t lds_bytes(int LP) {
  const size_t fixed = (size_t)(4 * kRows * kSC + kRows * kS2) * sizeof(f16) + 2 * kRows * 4 * sizeof(float) +
                       kMaxL * 4 * sizeof(int) + kRows * kMaxL * 2 * sizeof(float);
  const size_t ent = (size_t)LP * 128 * sizeof(Entry);
  const size_t ffn = 2 * kRows * kSC * sizeof(f16) + kRows * kSF * sizeof(float);
  return fixed + (ent > ffn ? ent : ffn);
}

bool dims_ok(int num_heads, int head_dim, int L, int P, int hidden, int ref_dim, int pos_feat) {
  return num_heads == kM && head_dim == kD && L >= 1 && L <= kMaxL && P >= 1 && L * P <= kMaxLP && ref_dim == 4 &&
         pos_feat == kC / 2 && hidden >= 256 && hidden % 256 == 0 && (kM * L * P * 3) % 16 == 0 && kM * L * P * 3 <= 512;
}

}  // namespace

extern "C" {

int codetr_decoder_layer_supported(int embed_dims, int num_heads, int num_levels, int num_points, int hidden, int ref_dim,
                                   int pos_feat) {
  return embed_dims == kC && dims_ok(num_heads, embed_dims / (num_heads > 0 ? num_heads : 1), num_levels, num_points, hidden,
                                     ref_dim, pos_feat)
             ? 1
             : 0;
}

int64_t codetr_decoder_layer_blob_halfs(int which, int num_levels, int num_points, int hidden) {
  switch (which) {
    case 0: return tail_layout(kM * num_levels * num_points * 3, hidden).total;
    case 1: return kHeadTotal;
    case 2: return kPosTotal;
    case 3: return 2 * kC;
    default: return CODETR_E_BADARG;
  }
}

int codetr_decoder_layer_f16(void* stream, const void* x_dev, const void* attn_dev, const void* qpos_dev, const void* ref_dev,
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
  a.F = hidden;
  a.n_ol = kM * num_levels * num_points * 3;
  a.eps = ln_eps;
  a.log2_temperature = log2f(temperature);
  a.tw = tail_layout(a.n_ol, hidden);
  const size_t lds = lds_bytes(num_levels * num_points);
  {
    static std::atomic<bool> done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0, done[0].store(false);
    if (!done[dev].load(std::memory_order_acquire)) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_layer_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      done[dev].store(true, std::memory_order_release);
    }
  }
  const unsigned blocks = (unsigned)((a.rows + kRows - 1) / kRows);
  hipLaunchKernelGGL(decoder_layer_kernel, dim3(blocks), dim3(kThreads), lds, static_cast<hipStream_t>(stream), a);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}

}  // extern "C"
