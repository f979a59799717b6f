/*
 * codetr_hip.h -- C ABI of libcodetr_hip.so, the MI355X (gfx950) native library that
 * replaces the reference's codetr/csrc for the Co-DETR inference hot path.
 *
 * Boundary rules (same for every entry point):
 *   - extern "C", plain pointers and sizes; no torch / ATen types cross this line.
 *   - every pointer named *_dev is DEVICE memory owned by the caller (in practice the
 *     PyTorch-ROCm caching allocator); it is borrowed for the duration of the
 *     asynchronous launch and must stay alive until the stream reaches that point.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).
 *     Work is enqueued and the call returns; nothing here synchronises, allocates,
 *     or reads device data on the host, so every entry point is hipGraph-capturable.
 *   - return value: 0 on success; a positive hipError_t value if the launch failed;
 *     a negative CODETR_E_* value if the arguments break the reference's contract
 *     (the reference raises c10::Error via AT_ASSERTM for those; the Python host
 *     turns every non-zero code into a RuntimeError -- nothing fails silently).
 *   - stateless and re-entrant.
 *
 * Which reference interface each entry point replaces is cited per function
 * (paths relative to the reference tree).
 */
#ifndef CODETR_HIP_H_
#define CODETR_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CODETR_E_BADARG (-1)      /* null pointer / non-positive dimension                       */
#define CODETR_E_IM2COL_STEP (-2) /* batch % min(batch, im2col_step) != 0  (ms_deform_attn.cu:924-926) */
#define CODETR_E_TOO_LARGE (-3)   /* a per-image extent exceeds the kernel's 32-bit in-image offsets  */
#define CODETR_E_UNSUPPORTED (-4) /* shape outside what the kernel family implements             */

/* ABI version of this header; bumped on any signature change. */
#define CODETR_HIP_ABI_VERSION 50
int codetr_hip_abi_version(void);
/* Human-readable message for a code returned by any entry point (static storage). */
const char *codetr_hip_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * Multi-scale deformable attention, forward.
 *
 * Replaces:  ms_deformable_im2col_cuda<T>(stream, value, spatial_shapes, level_start_index,
 *            sampling_loc, attn_weight, batch, spatial_size, num_heads, channels, num_levels,
 *            num_query, num_point, out)              codetr/csrc/ms_deform_attn.cu:762-779
 *            and the argument contract of
 *            codetr::ms_deform_attn_forward_reference codetr/csrc/ms_deform_attn.cu:899-956
 *            (also what the TensorRT plugin's enqueue hands over,
 *             codetr/csrc/deformable_attention_plugin.cpp:285-355).
 *
 *   value_dev          [B, S, M, D]          T     flattened multi-level value map
 *   spatial_shapes_dev [L, 2]                int64 (H_l, W_l), read ON DEVICE (cu:236-239)
 *   level_start_dev    [L]                   int64 first row of level l inside S
 *   loc_dev            [B, Nq, M, L, P, 2]   T     (x, y) normalised to [0, 1]
 *   weight_dev         [B, Nq, M, L, P]      T
 *   out_dev            [B, Nq, M*D]          T     every element is written (no pre-zeroing
 *                                                  needed; the reference zero-fills then
 *                                                  overwrites, cu:936)
 *   im2col_step        kept for API parity: validated exactly like cu:924-926, then the whole
 *                      batch is processed by one launch.
 *
 * Semantics: out[b,q,m,c] = sum_l sum_p w[b,q,m,l,p] * bilinear(value_l[b,:,m,c], x*W_l-0.5,
 * y*H_l-0.5) with zero padding per corner and the (-1, size) range gate (cu:31-77, 246-252).
 * Coordinates, bilinear weights and the accumulation are fp32 for f16/bf16/f32 tensors
 * (fp64 for f64) with one rounding at the store.
 * ------------------------------------------------------------------------------------------ */
int codetr_msda_forward_f16(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                            const int64_t *level_start_dev, const void *loc_dev, const void *weight_dev,
                            int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P, int64_t im2col_step,
                            void *out_dev);
int codetr_msda_forward_bf16(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                             const int64_t *level_start_dev, const void *loc_dev, const void *weight_dev,
                             int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P, int64_t im2col_step,
                             void *out_dev);
int codetr_msda_forward_f32(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                            const int64_t *level_start_dev, const void *loc_dev, const void *weight_dev,
                            int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P, int64_t im2col_step,
                            void *out_dev);
int codetr_msda_forward_f64(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                            const int64_t *level_start_dev, const void *loc_dev, const void *weight_dev,
                            int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P, int64_t im2col_step,
                            void *out_dev);

/* The windowed kernel behind codetr_msda_forward_f16 for encoder-shaped calls (round 6, csrc/msda_op4.hip; the same op,
 * reference codetr/csrc/ms_deform_attn.cu:211-261, 762-779): Nq == S, 5 levels x 4 points, 32-channel heads, fp16.  Per
 * (16 x 16 region of the finest level, head) the value rows the region's samples can reach are staged in LDS with a zero
 * border, sampling locations / attention weights reach the lanes through LDS records, samples outside a window take a
 * fix-up path with the reference's gate -- results as the general kernel's for ANY locations.  codetr_msda_forward_f16 calls it
 * by itself; the two entry points are exported for tests and benchmarks:
 *   codetr_msda_op4_supported    the host-visible part of the routing test (types, counts, 32-bit offset ranges); the
 *                                pyramid itself is a device tensor and is judged on the device (csrc/msda_op4_plan.h)
 *   codetr_msda_op4_forward_f16  launches the windowed kernel only: where the device-side plan does not apply its workgroups
 *                                return without writing `out` (codetr_msda_forward_f16 launches the general kernel behind
 *                                it, which evaluates the same plan and returns at once where this kernel did the work) */
int codetr_msda_op4_supported(int elem_bytes, int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P);
int codetr_msda_op4_forward_f16(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                                const int64_t *level_start_dev, const void *loc_dev, const void *weight_dev, int64_t B,
                                int64_t S, int M, int D, int L, int64_t Nq, int P, void *out_dev);

/* ------------------------------------------------------------------------------------------
 * Multi-scale deformable attention with its prologue fused.
 *
 * Same gather as above, but the kernel also performs the steps the reference's module runs between
 * its two projections and the op (codetr/multi_scale_deformable_attention.py:180-196): softmax over
 * the L*P attention logits of each (query, head), and sampling_locations = reference point +
 * normalised offset.  The sampling-location and attention-weight tensors (131 MB + 65 MB per encoder
 * call at 1920x1280 fp16) never exist.  The unfused entry points above remain the op's public form.
 *
 *   offsets_dev [B*Nq rows, offsets_row_stride]  columns [0, M*L*P*2): (x, y) offsets, order (m, l, p, xy)
 *   logits_dev  [B*Nq rows, logits_row_stride]   columns [0, M*L*P): raw attention logits, order (m, l, p)
 *               (both may be column ranges of ONE fused projection output: pass the same row stride
 *                and pointers offset by the column start)
 *   ref_dev     [B, Nq, L, ref_dim]  ref_dim 2: (x, y), loc = ref + off / (W_l, H_l)
 *                                    ref_dim 4: (x, y, w, h), loc = ref_xy + off / P * ref_wh * 0.5
 *   value_head_major  0: value_dev is [B, S, M, D] (the op's layout);
 *                     1: value_dev is [B, M, S, D] (written that way by codetr_linear_*'s head-major
 *                        epilogue): the x0/x1 neighbours of a sample are one 128-byte line, a pair is
 *                        served by twice the lanes and half the loads (D*2 bytes in {64, 128})
 * fp32 softmax / locations; 16-bit storage only; D*2 bytes per head must be 32, 64 or 128.
 * ------------------------------------------------------------------------------------------ */
int codetr_msda_fused_forward_f16(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                                  const int64_t *level_start_dev, const void *offsets_dev, int64_t offsets_row_stride,
                                  const void *logits_dev, int64_t logits_row_stride, const void *ref_dev, int ref_dim,
                                  int value_head_major, int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P,
                                  void *out_dev);
int codetr_msda_fused_forward_bf16(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                                   const int64_t *level_start_dev, const void *offsets_dev, int64_t offsets_row_stride,
                                   const void *logits_dev, int64_t logits_row_stride, const void *ref_dev, int ref_dim,
                                   int value_head_major, int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P,
                                   void *out_dev);
/* The same with ref_dev in fp32 ([B, Nq, L, ref_dim] float, 8-byte aligned) while value / offsets / logits / output keep
 * their 16-bit type: a 16-bit model's reference points (sigmoid outputs in [0, 1]) resolve a coordinate to 1/2048,
 * a quarter pixel on a 480-wide level; codetr_query_sine_embed_f16 writes them unrounded for the decoder's layers. */
int codetr_msda_fused_forward_ref32_f16(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                                        const int64_t *level_start_dev, const void *offsets_dev,
                                        int64_t offsets_row_stride, const void *logits_dev, int64_t logits_row_stride,
                                        const void *ref_dev, int ref_dim, int value_head_major, int64_t B, int64_t S,
                                        int M, int D, int L, int64_t Nq, int P, void *out_dev);
int codetr_msda_fused_forward_ref32_bf16(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                                         const int64_t *level_start_dev, const void *offsets_dev,
                                         int64_t offsets_row_stride, const void *logits_dev, int64_t logits_row_stride,
                                         const void *ref_dev, int ref_dim, int value_head_major, int64_t B, int64_t S,
                                         int M, int D, int L, int64_t Nq, int P, void *out_dev);

/* ------------------------------------------------------------------------------------------
 * Encoder self-attention form of the fused op (LDS-staged gather), for DetrTransformerEncoder
 * (codetr/transformer.py:81-92), where the queries ARE the pixels of the flattened multi-level map (Nq == S, query q =
 * pixel q) and reference_points are 2-d (get_reference_points, codetr/transformer.py:280-305).
 * (The round-3/4 forms codetr_msda_encoder_forward[_win]_* left the library in round 6 with their kernels:
 * tools/micro/experiments/msda_encoder_v3.hip.)
 * ------------------------------------------------------------------------------------------ */
/* Round-5 form of the encoder kernel ("v4", csrc/msda_encoder4.hip): the same op -- ms_deform_attn.cu:31-77, 211-261 with
 * the softmax / sampling-location prologue of multi_scale_deformable_attention.py:180-196 and the reference points of
 * transformer.py:280-305 -- for fp16, L == 5, P == 4, D == 32, with the (offsets | logits) projection handed over in a
 * LANE-MAJOR PACKED layout so that a quad lane of the gather fetches all it needs with two 16-byte loads:
 *   packed_dev        [B * S][packed_row_stride] fp16, packed_row_stride >= 64 M and a multiple of 8; for head m and
 *                     point p (= the quad lane), the 16 halves at columns 64 m + 16 p .. + 15 are
 *                       (x, y) offset of (m, level 0, p), ... , (m, level 4, p)        10 halves   [sampling_offsets rows
 *                       logit of (m, level 0, p), ... , (m, level 4, p)                 5 halves    ((m L + l) P + p) 2 + xy,
 *                       one pad half (ignored)                                                       attention_weights rows (m L + l) P + p]
 *                     i.e. the output of the (sampling_offsets | attention_weights) Linear with its weight rows permuted
 *                     once on the host (codetr_msda_pack_projection_index gives the permutation).
 *   valid_counts_dev  [B][L][2] fp32, required: the reference points are computed in fp32 from the query's pixel centre
 *                     and the valid pixel counts (see codetr_msda_encoder_forward_win_f16).
 *   windows_host      [M][L][4] int8 as above.  A window may exceed the level: it is clamped to the image plus a one-pixel
 *                     ZERO border (corners outside the image read zeros from LDS, ms_deform_attn.cu:52-71), so a wide
 *                     window on a coarse level keeps the whole level resident.
 *   region_w/region_h region of a workgroup in pixels of the finest level; threads: 256 | 512 per workgroup.  A region may
 *                     hold at most 3 * threads / 4 queries (CODETR_E_UNSUPPORTED otherwise).
 *   value_head_major  0: value_dev is the op's [B, S, M, D]; 1: [B, M, S, D] (each head's map contiguous, what
 *                     codetr_linear_* writes with hm_head_dim = D): a staged window row is then one contiguous run.
 * Samples outside the windows are added from global memory with the reference's gate / corner logic: windows change
 * speed, never results beyond the rounding of the packed blend (same tolerance statement as the _win entry).
 * codetr_msda_encoder_packed_lds_bytes: LDS bytes per workgroup such a launch needs, or a negative CODETR_E_* code.
 * codetr_msda_pack_projection_index: idx[64 M] (host): source row of the concatenated [M L P 2 + M L P] projection for
 * every packed output column, -1 for the pad columns. */
int codetr_msda_encoder_forward_packed_f16(void *stream, const void *value_dev, const int64_t *level_shapes_host,
                                           const void *packed_dev, int64_t packed_row_stride,
                                           const float *valid_counts_dev, int64_t B, int64_t S, int M, int D, int L, int P,
                                           const int8_t *windows_host, int region_w, int region_h, int threads,
                                           int value_head_major, void *out_dev);
/* The same for a bf16 model: packed projection and output are bf16, the VALUE MAP IS FP16 (value_f16_dev; written by
 * codetr_linear_bf16_f16out): the blend runs on packed halves either way -- gfx950 has no packed bf16 FMA -- and an fp16
 * value map keeps three more mantissa bits of the projection's fp32 accumulators than a bf16 one would.  Offsets and logits
 * are widened exactly; the result is rounded to bf16 once. */
int codetr_msda_encoder_forward_packed_bf16(void *stream, const void *value_f16_dev, const int64_t *level_shapes_host,
                                            const void *packed_dev, int64_t packed_row_stride,
                                            const float *valid_counts_dev, int64_t B, int64_t S, int M, int D, int L, int P,
                                            const int8_t *windows_host, int region_w, int region_h, int threads,
                                            int value_head_major, void *out_dev);
int64_t codetr_msda_encoder_packed_lds_bytes(const int64_t *level_shapes_host, int M, int L, int P,
                                             const int8_t *windows_host, int region_w, int region_h, int threads);
int codetr_msda_pack_projection_index(int M, int L, int P, int32_t *idx_host);

/* ------------------------------------------------------------------------------------------
 * Patch gather of the Swin stem (mmdet PatchEmbed: Conv2d(C, E, k, stride k) with "corner" zero padding, reference
 * codetr/swin.py:13, 567; equivalent source codetr/transformer_mmcv.py:100-210).  The convolution over
 * non-overlapping patches is a GEMM: this call writes its left operand,
 *   out[(b, ty, tx)][(c, ky, kx)] = x[b, c, k ty + ky, k tx + kx]   (0 beyond the image and for columns >= C k k),
 * in the column order of conv.weight.view(E, C k k), so that codetr_linear_* with the weight padded to kpad columns
 * produces the token-major [B, ceil(H/k) ceil(W/k), E] map directly (no NCHW round trip, no flatten / transpose).
 *   x_dev   [B, C, H, W]  16-bit elements (fp16 or bf16: pure data movement), 8-byte aligned
 *   out_dev [B * ceil(H/k) * ceil(W/k), kpad]  same element type, 16-byte aligned
 * Implemented for k == 4, kpad == 64 (C <= 4): CODETR_E_UNSUPPORTED otherwise.
 * ------------------------------------------------------------------------------------------ */
int codetr_patch_im2col_b16(void *stream, const void *x_dev, int64_t B, int C, int64_t H, int64_t W, int k, int kpad,
                            void *out_dev);

/* k x k / stride / zero-padding patches of a token-major map, for convolutions run as GEMMs on token-major data
 * (the neck's extra level: Conv2d(1536, 256, 3, stride 2, padding 1) of mmdet's ChannelMapper, built at
 * codetr/codetr.py:53-54 from configs lsj:40-47):
 *   out[(b, oy, ox)][(ky, kx, c)] = x[b, oy*stride + ky - pad, ox*stride + kx - pad, c]   (0 outside the map)
 *   x_dev [B, H, W, C] 16-bit elements, C % 8 == 0;  out_dev [B*Ho*Wo, k*k*C], Ho = (H + 2 pad - k) / stride + 1.
 * The GEMM weight is conv.weight permuted to [C_out, (ky, kx, c)]. */
int codetr_im2col_tokens_b16(void *stream, const void *x_dev, int64_t B, int64_t H, int64_t W, int64_t C, int k,
                             int stride, int pad, void *out_dev);

/* Row-wise top-k of 16-bit floats: the two selections of the detection head -- the two-stage proposals
 * (torch.topk(enc_outputs_class.max(-1)[0], 900, dim=1), reference codetr/transformer.py:560-561) and the final
 * detections (cls_score.view(B, -1).topk(300), reference codetr/co_dino_head.py:183-186).
 *   x_dev       [rows, n]  f16 / bf16, contiguous
 *   values_dev  [rows, k]  same type, or NULL;   indices_dev [rows, k] int64
 * Sorted: descending value, ties by ascending index, NaN first (torch's NaN rule; torch leaves the tie order open).
 * k <= 1024, k <= n < 2^24: CODETR_E_UNSUPPORTED otherwise. */
int codetr_topk_f16(void *stream, const void *x_dev, int64_t rows, int64_t n, int k, void *values_dev,
                    int64_t *indices_dev);
int codetr_topk_bf16(void *stream, const void *x_dev, int64_t rows, int64_t n, int k, void *values_dev,
                     int64_t *indices_dev);

/* The same selection with a long row cut over several workgroups (the 204 600-element proposal row of one image):
 *   codetr_topk_chunks         how many equal pieces to cut a row into (1: call codetr_topk_*) and the workspace needed
 *   codetr_topk_chunked_*      pass 1: top k of every piece (rows * chunks workgroups); pass 2: top k of the
 *                              chunks * k candidates, mapped back to row indices.  Same result, same stable order.
 * Caller-owned workspace (16-byte aligned), no allocation, hipGraph-capturable. */
int64_t codetr_topk_chunks(int64_t n, int k, int64_t rows, int64_t *workspace_bytes);
int codetr_topk_chunked_f16(void *stream, const void *x_dev, int64_t rows, int64_t n, int k, int chunks,
                            void *values_dev, int64_t *indices_dev, void *workspace_dev, int64_t workspace_bytes);
int codetr_topk_chunked_bf16(void *stream, const void *x_dev, int64_t rows, int64_t n, int k, int chunks,
                             void *values_dev, int64_t *indices_dev, void *workspace_dev, int64_t workspace_bytes);

/* Name of the kernel variant the arguments above would dispatch to ("tiled_d32x8", "scalar", ...).
 * Pure host function; lets tests assert that the model shape takes the tiled path. */
const char *codetr_msda_variant(int elem_bytes, int M, int D, int L, int P);

/* ------------------------------------------------------------------------------------------
 * Fused linear layer:  y[M,N] = act(x[M,K] . w[N,K]^T + bias[N]) (+ residual[M,N])
 *
 * Replaces, for the nn.Linear layers of the hot path, the ATen call sequence the reference runs
 * per layer (addmm / bias add / ReLU or GELU / residual add as separate kernels):
 *   Swin qkv / proj / MLP           codetr/swin.py:91-115, codetr/transformer_mmcv.py:484-500
 *   MSDA value / offsets / weights / output projections
 *                                   codetr/multi_scale_deformable_attention.py:173-213
 *   encoder / decoder FFN, MHA projections, reference-point MLP, class / box branches
 *                                   codetr/transformer.py:218, 551-557; codetr/co_dino_head.py:169-177
 *
 *   x_dev        [M, K]  row-major, dense            T (f16 / bf16)
 *   w_dev        [N, K]  row-major (nn.Linear.weight) T
 *   bias_dev     [N] or NULL                          T
 *   residual_dev [M, N] or NULL (added AFTER the activation, as `identity + ffn(x)` does)   T
 *   row_mask_dev [M] uint8/bool or NULL: rows with mask value 1 (any non-zero value but 2) are written as
 *                zeros (before the residual) -- `value.masked_fill(key_padding_mask[..., None], 0.0)` of
 *                codetr/multi_scale_deformable_attention.py:174-175 folded into value_proj.  Mask value 2: the
 *                INPUT row counts as zeros, y = act(bias) -- `memory * keep` of apply_mask_to_proposal_and_memory
 *                (codetr/transformer.py:365-380) folded into enc_output
 *   y_dev        [M, N]  (may alias residual_dev)     T
 *   act          0 = none, 1 = ReLU, 2 = GELU (erf form, nn.GELU default)
 *   hm_rows, hm_head_dim   0, 0: y is row-major [M, N].  Otherwise the rows are (batch, position) with
 *                hm_rows positions per batch and the columns (head, channel) with hm_head_dim channels per
 *                head, and y is written HEAD-MAJOR: y[b][head][position][channel] -- the value-map layout
 *                of codetr_msda_fused_forward_*(value_head_major = 1).  Needs N % 8 == 0, no residual.
 *
 * fp32 accumulation on the MFMA units, one rounding at the store.  K must be a multiple of 64
 * (every Linear of the model is; CODETR_E_UNSUPPORTED otherwise); x / w 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
int codetr_linear_f16(void *stream, const void *x_dev, const void *w_dev, const void *bias_dev,
                      const void *residual_dev, const void *row_mask_dev, void *y_dev, int64_t M, int64_t N,
                      int64_t K, int act, int64_t hm_rows, int hm_head_dim);
int codetr_linear_bf16(void *stream, const void *x_dev, const void *w_dev, const void *bias_dev,
                       const void *residual_dev, const void *row_mask_dev, void *y_dev, int64_t M, int64_t N,
                       int64_t K, int act, int64_t hm_rows, int hm_head_dim);
/* bf16 operands, FP16 output (no activation, no residual; row mask and head-major destination as above): the value
 * projection in front of codetr_msda_encoder_forward_packed_bf16, whose staged value map is fp16 whatever the model's
 * type (its blend runs on packed halves; gfx950 has no packed bf16 FMA).  fp32 accumulators rounded once to fp16 -- three
 * more mantissa bits than the bf16 store they replace.  Large short-K problems only (the X-stationary kernel: K in
 * {192, 256}, M >= 32 768, 128 <= N <= 1536); CODETR_E_UNSUPPORTED otherwise. */
int codetr_linear_bf16_f16out(void *stream, const void *x_dev, const void *w_dev, const void *bias_dev,
                              const void *row_mask_dev, void *y_dev, int64_t M, int64_t N, int64_t K,
                              int64_t hm_rows, int hm_head_dim);
/* The encoder self-attention's two projections of one token row as ONE launch (reference
 * codetr/multi_scale_deformable_attention.py:161-162 `query = query + query_pos`, :173-176 value_proj + masked_fill,
 * :177-182 sampling_offsets | attention_weights; in the encoder `value` IS `query`):
 *     value  [M, N_value]  = x @ W[:N_value]^T + b[:N_value]            rows with row_mask != 0 zeroed; head-major
 *                            destination [M / hm_rows][N_value / hm_head_dim][hm_rows][hm_head_dim] if hm_head_dim != 0
 *     packed [M, N_packed] = (x + pos) @ W[N_value:]^T + b[N_value:]    x + pos rounded to the operand type first, exactly
 *                            as the separate add; row-major
 * w_dev [N_value + N_packed, K] and bias_dev are the two layers' parameters concatenated (value rows first).  x and pos
 * are each read once for both products -- as two launches x is read twice (419 MB of 2.5 GB at four 1920x1280 images)
 * and the value projection pays its own prologue.  _bf16: operands, packed output bf16; the value map is FP16 (see
 * codetr_linear_bf16_f16out).  X-stationary kernel: K == 256, N_value % 64 == 0, N_packed % 8 == 0, N_value + N_packed
 * <= 1536, M >= 32 768, 16-byte aligned pointers; CODETR_E_UNSUPPORTED otherwise (the caller then launches the two GEMMs). */
int codetr_encoder_projections_f16(void *stream, const void *x_dev, const void *pos_dev, const void *w_dev,
                                   const void *bias_dev, const void *row_mask_dev, void *value_dev, void *packed_dev,
                                   int64_t M, int64_t N_value, int64_t N_packed, int64_t K, int64_t hm_rows,
                                   int hm_head_dim);
int codetr_encoder_projections_bf16(void *stream, const void *x_dev, const void *pos_dev, const void *w_dev,
                                    const void *bias_dev, const void *row_mask_dev, void *value_f16_dev,
                                    void *packed_dev, int64_t M, int64_t N_value, int64_t N_packed, int64_t K,
                                    int64_t hm_rows, int hm_head_dim);
/* Round 6: the same launch with the positional operand GENERATED in the kernel instead of read.  `pos` of the encoder is the
 * sine positional encoding + level embedding (reference positional_encoding.py:58-93, transformer.py:508-519: what
 * codetr_sine_pos_tokens_* writes): a function of the token's (level, y, x) and of the running sums of the padding mask.  The
 * kernel re-derives a row's two normalised coordinates from the running sums (8 bytes per token instead of the row's 512) and
 * evaluates its channels with the same fp32 operations in the same order as codetr_sine_pos_tokens_*, so value / packed are
 * BIT-IDENTICAL to codetr_encoder_projections_* fed the tensor that entry writes (tests/test_encoder_projections_gpu.py).
 *   ycum0..4 / xcum0..4   per level (nullptr beyond num_levels): running sums of the not-mask along y / x, [B, H_l, W_l] fp32
 *                         (the inputs of codetr_sine_pos_tokens_*);  level_shapes_host [num_levels][2] = (H_l, W_l), sum = S
 *   level_embed_dev       [num_levels, K] in the operands' type, or nullptr;  temperature .. normalize as codetr_sine_pos_tokens_*
 *   M = B * S rows;  the other arguments as codetr_encoder_projections_*. */
int codetr_encoder_projections_posgen_f16(void *stream, const void *x_dev, const float *ycum0, const float *ycum1,
                                          const float *ycum2, const float *ycum3, const float *ycum4, const float *xcum0,
                                          const float *xcum1, const float *xcum2, const float *xcum3, const float *xcum4,
                                          const int64_t *level_shapes_host, int num_levels, const void *level_embed_dev,
                                          float temperature, float scale, float eps, float offset, int normalize,
                                          const void *w_dev, const void *bias_dev, const void *row_mask_dev, void *value_dev,
                                          void *packed_dev, int64_t M, int64_t S, int64_t N_value, int64_t N_packed, int64_t K,
                                          int64_t hm_rows, int hm_head_dim);
int codetr_encoder_projections_posgen_bf16(void *stream, const void *x_dev, const float *ycum0, const float *ycum1,
                                           const float *ycum2, const float *ycum3, const float *ycum4, const float *xcum0,
                                           const float *xcum1, const float *xcum2, const float *xcum3, const float *xcum4,
                                           const int64_t *level_shapes_host, int num_levels, const void *level_embed_dev,
                                           float temperature, float scale, float eps, float offset, int normalize,
                                           const void *w_dev, const void *bias_dev, const void *row_mask_dev,
                                           void *value_f16_dev, void *packed_dev, int64_t M, int64_t S, int64_t N_value,
                                           int64_t N_packed, int64_t K, int64_t hm_rows, int hm_head_dim);

/* Which of the three kernels behind codetr_linear_* serves a (16-byte aligned) problem: "tile128" (128x128 tiles, the
 * general kernel), "tile256" (256x256 tiles, one workgroup per CU), "xs" (X-stationary short-K kernel) or
 * "unsupported" (K % 64 != 0).  Pure function of the arguments and of the CODETR_GEMM_* environment switches; it is the
 * same routine the launcher consults, so tests can assert which kernel a model shape exercises. */
const char *codetr_linear_variant(int64_t M, int64_t N, int64_t K, int act, int has_residual, int hm_head_dim);

/* y = (x + x_add) . w^T + bias with the element-wise add folded into the operand load of the short-K kernel:
 * `query + query_pos` in front of the (offsets | logits) projection of MultiScaleDeformableAttention
 * (codetr/multi_scale_deformable_attention.py:161-162, 177-179) without a separate add kernel or a stored sum.
 * x + x_add is rounded to T first, exactly as the stand-alone fp16 / bf16 add.  x_add_dev [M, K] like x_dev.
 * Only where the X-stationary kernel applies (K in {192, 256}, 128 <= N <= 1536, N % 8 == 0, M >= 32768):
 * CODETR_E_UNSUPPORTED otherwise -- callers then add and call codetr_linear_*. */
int codetr_linear_xadd_f16(void *stream, const void *x_dev, const void *x_add_dev, const void *w_dev,
                           const void *bias_dev, void *y_dev, int64_t M, int64_t N, int64_t K);
int codetr_linear_xadd_bf16(void *stream, const void *x_dev, const void *x_add_dev, const void *w_dev,
                            const void *bias_dev, void *y_dev, int64_t M, int64_t N, int64_t K);

/* y = act(LayerNorm(x) . w^T + bias) with the LayerNorm applied to the rows as they are loaded by the short-K kernel
 * (the whole row is that kernel's stationary operand): Swin's pre-norm blocks, norm1 -> qkv and norm2 -> fc1 (reference
 * codetr/swin.py:345-386), where nothing else reads the normalised tensor.  fp32 two-pass statistics, the normalised
 * row rounded to T before the product (what the separate kernel writes).  gamma / beta [K].
 * Only where the X-stationary kernel applies (K in {192, 256}, 128 <= N <= 1536, N % 8 == 0, M >= 32768):
 * CODETR_E_UNSUPPORTED otherwise -- callers then run codetr_layernorm_* and codetr_linear_*. */
int codetr_linear_ln_f16(void *stream, const void *x_dev, const void *ln_gamma_dev, const void *ln_beta_dev,
                         float ln_eps, const void *w_dev, const void *bias_dev, void *y_dev, int64_t M, int64_t N,
                         int64_t K, int act);
int codetr_linear_ln_bf16(void *stream, const void *x_dev, const void *ln_gamma_dev, const void *ln_beta_dev,
                          float ln_eps, const void *w_dev, const void *bias_dev, void *y_dev, int64_t M, int64_t N,
                          int64_t K, int act);

/* ------------------------------------------------------------------------------------------
 * FP8 (OCP e4m3) linear layer and its activation producers -- BASELINE config 5 ("fp8 weights + activations").
 * No reference counterpart (the reference's dtypes stop at half: codetr/csrc/ms_deform_attn.cu:946,
 * export.py:39-44); the layers served are the nn.Linears of codetr/swin.py:91-115, 345-355.
 *
 *   codetr_linear_fp8      y = act((x8 . w8^T) * x_scale * w_scale[n] + bias[n]) (+ residual)
 *                          x8 [M, K] / w8 [N, K] e4m3 bytes, K contiguous; w_scale [N] fp32 (per output channel);
 *                          x_scale: the static per-tensor scale the producer divided by; bias [N] / residual [M, N]
 *                          fp16 or NULL; y [M, N] fp16, or e4m3 = sat(y / out_scale) when out_is_fp8 (then no
 *                          residual).  fp32 accumulation with v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales).
 *                          K % 128 == 0, N % 8 == 0, 16-byte aligned bases; act as codetr_linear_*.
 *   codetr_cast_fp8_f16    y8[i] = sat(x[i] / scale), n % 8 == 0 elements
 *   codetr_layernorm_fp8_f16  LayerNorm over the last dimension C (fp32 statistics, result rounded to fp16 as the
 *                          fp16 model would) then sat(. / scale) -> e4m3; C % 8 == 0, C <= 4096
 * Saturation: values are clamped to +-448 (the largest finite e4m3) before conversion.
 * ------------------------------------------------------------------------------------------ */
int codetr_linear_fp8(void *stream, const void *x8_dev, const void *w8_dev, const float *w_scale_dev, float x_scale,
                      const void *bias_f16_dev, const void *residual_f16_dev, void *y_dev, int out_is_fp8,
                      float out_scale, int64_t M, int64_t N, int64_t K, int act);
int codetr_cast_fp8_f16(void *stream, const void *x_f16_dev, void *y8_dev, int64_t n, float scale);
int codetr_layernorm_fp8_f16(void *stream, const void *x_f16_dev, const void *gamma_f16_dev, const void *beta_f16_dev,
                             void *y8_dev, int64_t rows, int64_t C, float eps, float scale);

/* ------------------------------------------------------------------------------------------
 * The same path with MX BLOCK SCALES on the activations (no calibration, no static scale): every 32 consecutive
 * K-elements of an activation row share one e8m0 exponent byte (2^(byte - 127)) chosen by the producer from the block's
 * own maximum, and v_mfma_scale_f32_16x16x128_f8f6f4 applies it in hardware -- a lane's scale byte covers exactly the 32
 * operand bytes that lane holds.  Weights keep one fp32 scale per output channel (unit block scales in the instruction).
 *
 * Scale tensors: codetr_mx_scale_bytes(M, K) bytes for an activation [M, K], byte of (row m, block kb = k / 32) at
 *   (((kb >> 2) * MB + (m >> 7)) * 64 + (kb & 3) * 16 + (m & 15)) * 8 + ((m >> 4) & 7),   MB = ceil(M / 128)
 * (the 8 bytes a lane of the GEMM needs per 128-wide k-tile are contiguous: csrc/mx_scale.h).
 *
 *   codetr_linear_fp8mx          y = act((x8 (*) x_scales) . w8^T * w_scale[n] + bias[n]) (+ residual); y fp16, or -- when
 *                                y_scales_dev is given (no residual, N % 128 == 0) -- e4m3 with block scales along N, laid
 *                                out for a consumer GEMM over the same M rows whose K is this N
 *   codetr_cast_fp8mx_f16        x [rows, C] fp16 -> e4m3 + scales (C % 128 == 0)
 *   codetr_layernorm_fp8mx_f16   LayerNorm (as codetr_layernorm_fp8_f16) -> e4m3 + scales (C % 128 == 0)
 *   codetr_window_attention_fp8mx_f16  codetr_window_attention_f16 with its output as e4m3 + scales: one block per
 *                                (token, head) -- head_dim == 32 is the MX block
 * ------------------------------------------------------------------------------------------ */
int64_t codetr_mx_scale_bytes(int64_t M, int64_t K);
int codetr_linear_fp8mx(void *stream, const void *x8_dev, const void *x_scales_dev, const void *w8_dev,
                        const float *w_scale_dev, const void *bias_f16_dev, const void *residual_f16_dev, void *y_dev,
                        void *y_scales_dev, int64_t M, int64_t N, int64_t K, int act);
int codetr_cast_fp8mx_f16(void *stream, const void *x_f16_dev, void *y8_dev, void *y_scales_dev, int64_t rows, int64_t C);
int codetr_layernorm_fp8mx_f16(void *stream, const void *x_f16_dev, const void *gamma_f16_dev, const void *beta_f16_dev,
                               void *y8_dev, void *y_scales_dev, int64_t rows, int64_t C, float eps);
int codetr_window_attention_fp8mx_f16(void *stream, const void *qkv_dev, const void *qkv_bias_dev,
                                      const void *rel_bias_dev, void *out8_dev, void *out_scales_dev, int64_t B, int64_t H,
                                      int64_t W, int num_heads, int head_dim, int window_size, int shift);

/* ------------------------------------------------------------------------------------------
 * Small fp16 element-wise / gather kernels of the decoder and the detection head (csrc/small_ops.hip): the last
 * ATen launches of the fp16 forward, bit-identical to the ATen formulation (one fp16 rounding per operation).
 *   codetr_add_f16           out[i] = a[i % a_period] + b[i]  (`query + query_pos`, reference transformer_mmcv.py:400-404;
 *                            a_period < n broadcasts `a`, e.g. query_embed.weight over the batch); n, a_period % 8 == 0
 *   codetr_sigmoid_f16       out = 1 / (1 + exp(-x))            (co_dino_head.py:181)
 *   codetr_gather_rows_b16   out[b,k,:] = src[b, idx[b,k], :]   (transformer.py:562-566; 16-bit rows, C % 4 == 0)
 *   codetr_decode_boxes_f16  q = idx / C, label = idx % C, sigmoid -> cxcywh -> xyxy -> x (W,H,W,H) -> clamp
 *                            (co_dino_head.py:177-209 + mmdet bbox_cxcywh_to_xyxy); coords_unact [B,Nq,4] = box branch
 *                            output + reference (before the sigmoid); boxes [B,K,4] fp16, labels [B,K] int64
 * Every entry point of this group, codetr_query_sine_embed_* and codetr_encoder_geometry_* / codetr_row_max_* below have a
 * _bf16 twin (same formulas with bfloat16 roundings and finfo(bf16).max): a bf16 model's forward is all C-ABI launches too.
 *   codetr_valid_ratios_f16  out[b,l,j] = fp16(counts[b,l,j]) / level_wh[l,j]   (transformer.py:384-400);
 *                            out32 (may be NULL): counts / level_wh in fp32, unrounded
 * ------------------------------------------------------------------------------------------ */
int codetr_add_f16(void *stream, const void *a_dev, const void *b_dev, void *out_dev, int64_t n, int64_t a_period);
int codetr_add_bf16(void *stream, const void *a_dev, const void *b_dev, void *out_dev, int64_t n, int64_t a_period);
int codetr_sigmoid_f16(void *stream, const void *x_dev, void *out_dev, int64_t n);
int codetr_sigmoid_bf16(void *stream, const void *x_dev, void *out_dev, int64_t n);
int codetr_gather_rows_b16(void *stream, const void *src_dev, const int64_t *idx_dev, void *out_dev, int64_t B, int64_t S,
                           int64_t K, int64_t C);
int codetr_decode_boxes_f16(void *stream, const void *coords_unact_dev, const int64_t *idx_dev, void *boxes_dev,
                            int64_t *labels_dev, int64_t B, int64_t Nq, int64_t K, int num_classes, float img_w,
                            float img_h);
int codetr_decode_boxes_bf16(void *stream, const void *coords_unact_dev, const int64_t *idx_dev, void *boxes_dev,
                            int64_t *labels_dev, int64_t B, int64_t Nq, int64_t K, int num_classes, float img_w,
                            float img_h);
int codetr_valid_ratios_f16(void *stream, const float *counts_dev, const void *level_wh_f16_dev, void *out_dev,
                            float *out32_dev, int64_t B, int L);
int codetr_valid_ratios_bf16(void *stream, const float *counts_dev, const void *level_wh_bf16_dev, void *out_dev,
                            float *out32_dev, int64_t B, int L);

/* Split-K form of the same layer for problems with few output tiles and a long K -- the neck's extra
 * 3x3 / stride-2 level (codetr/codetr.py neck, mmdet ChannelMapper extra_convs) run as a GEMM over unfolded
 * patches is [600, 13824] x [256, 13824]^T: 10 output tiles, one pass leaves 246 CUs idle.
 *   codetr_linear_splitk_plan  returns the number of K ranges the library would use for (M, N, K) -- 1 means
 *                              "call codetr_linear_*" -- and the fp32 workspace it needs (splits * M * N * 4 bytes).
 *   codetr_linear_splitk_*     pass 1 writes one fp32 partial tile per (output tile, K range) into the workspace,
 *                              pass 2 sums the ranges in fp32 and applies bias / activation / row mask / residual
 *                              with the same semantics as codetr_linear_*.  `splits` must come from the plan.
 * The workspace is caller-owned device memory (16-byte aligned) so that the calls stay allocation-free and
 * hipGraph-capturable. */
int codetr_linear_splitk_plan(int64_t M, int64_t N, int64_t K, int64_t *workspace_bytes);
int codetr_linear_splitk_f16(void *stream, const void *x_dev, const void *w_dev, const void *bias_dev,
                             const void *residual_dev, const void *row_mask_dev, void *y_dev, int64_t M, int64_t N,
                             int64_t K, int act, int splits, void *workspace_dev, int64_t workspace_bytes);
int codetr_linear_splitk_bf16(void *stream, const void *x_dev, const void *w_dev, const void *bias_dev,
                              const void *residual_dev, const void *row_mask_dev, void *y_dev, int64_t M, int64_t N,
                              int64_t K, int act, int splits, void *workspace_dev, int64_t workspace_bytes);

/* Persistent form of codetr_linear_* for the large layers (Swin stages 0-3 -- reference codetr/swin.py:92-112,
 * 331-352 -- where TensorRT picked its own GEMM tactic): one resident workgroup per CU walks whole 256 x 256 tiles with
 * its operand stream running across tile boundaries (csrc/gemm_sk.hip); same semantics as codetr_linear_* without row
 * mask / head-major output.
 *   codetr_linear_sk_supported        1 when (M, N, K) is inside the kernel's domain (K % 64 == 0, K >= 128, N % 8 == 0)
 *   codetr_linear_sk_preferred        1 when it measured faster than codetr_linear_* on this class of problem
 *                                     (profiles/r04_gemm_sk.txt): hosts call codetr_linear_sk_* then
 *   codetr_linear_sk_workspace_bytes  size of the caller-owned workspace (fp32 partial tiles + ticket counters of the
 *                                     stream-K split) on the current device; it must be ZERO-FILLED ONCE before its
 *                                     first use -- every launch returns the counters to zero -- and must not be shared
 *                                     by launches that can run concurrently (one per stream); without the 0x40 flag
 *                                     it is not touched and may be NULL / 0
 *   codetr_linear_sk_*                flags: 0 (default); 0x40 = stream-K split of the left-over tiles (correct, measured
 *                                     slower at this model's K); 0x20 = one workgroup per tile (A/B only) */
int64_t codetr_linear_sk_workspace_bytes(void);
int codetr_linear_sk_supported(int64_t M, int64_t N, int64_t K);
int codetr_linear_sk_preferred(int64_t M, int64_t N, int64_t K, int act, int has_residual);
int codetr_linear_sk_f16(void *stream, const void *x_dev, const void *w_dev, const void *bias_dev,
                         const void *residual_dev, void *y_dev, int64_t M, int64_t N, int64_t K, int act,
                         void *workspace_dev, int64_t workspace_bytes, int flags);
int codetr_linear_sk_bf16(void *stream, const void *x_dev, const void *w_dev, const void *bias_dev,
                          const void *residual_dev, void *y_dev, int64_t M, int64_t N, int64_t K, int act,
                          void *workspace_dev, int64_t workspace_bytes, int flags);

/* Fused Swin MLP (round 6, csrc/swin_mlp.hip): y = x + fc2(GELU(fc1(LayerNorm(x)))) in one launch -- the second half of a
 * SwinBlock (reference codetr/swin.py:331-352: x = x + ffn(norm2(x)), FFN = Linear(C, 4C) -> GELU (erf) -> Linear(4C, C)),
 * for the stages whose MLP is bound by the bytes of its hidden activation: C = 192 / 384 (Swin-L stages 0 / 1).  The hidden
 * activation and norm2's output never leave the CU.
 *   x_dev / y_dev        [M, C] row-major, 16-bit storage (y may not alias x: rows are read twice)
 *   ln_gamma / ln_beta   [C], ln_eps: norm2
 *   w1_dev [4C, C], b1_dev [4C];  w2_packed_dev [C, 4C] = codetr_ffn_pack_w2_f16(fc2.weight);  b2_dev [C]
 * CODETR_E_UNSUPPORTED for other C.  Numerics: LayerNorm and GELU in fp32, each rounded to the storage type once; fp32
 * accumulation; fc2's output rounded before the identity is added (the reference's two roundings). */
int codetr_swin_mlp_supported(int64_t M, int64_t C, int64_t hidden);
int codetr_swin_mlp_f16(void *stream, const void *x_dev, const void *ln_gamma_dev, const void *ln_beta_dev, float ln_eps,
                        const void *w1_dev, const void *b1_dev, const void *w2_packed_dev, const void *b2_dev, void *y_dev,
                        int64_t M, int64_t C);
int codetr_swin_mlp_bf16(void *stream, const void *x_dev, const void *ln_gamma_dev, const void *ln_beta_dev, float ln_eps,
                         const void *w1_dev, const void *b1_dev, const void *w2_packed_dev, const void *b2_dev, void *y_dev,
                         int64_t M, int64_t C);

/* Ping-pong form of codetr_linear_sk_* (round 6, csrc/gemm_pp.hip) for the same layers (reference codetr/swin.py:92-112,
 * 331-352): the same persistent 256 x 256 tiles, operand ring and epilogue, but the two waves of every SIMD take turns --
 * one multiplies (32 MFMAs, nothing else) while the other reads its fragments from LDS and issues its LDS-DMA pieces, a
 * workgroup barrier between the segments, the second group of four waves one barrier behind the first.  Same semantics
 * and domain as codetr_linear_sk_* (no workspace, no stream-K split).
 *   codetr_linear_pp_supported   1 when (M, N, K) is inside the kernel's domain (K % 64 == 0, K >= 128, N % 8 == 0)
 *   codetr_linear_pp_preferred   1 where it measured faster than codetr_linear_* and codetr_linear_sk_*
 *                                (profiles/r06_gemm_pp.txt): hosts call codetr_linear_pp_* then
 *   codetr_linear_pp_*           flags: 0 (reserved: the measured variants of the main loop are kept out of the library,
 *                                tools/micro/experiments/gemm_pp_variants.hip) */
int codetr_linear_pp_supported(int64_t M, int64_t N, int64_t K);
int codetr_linear_pp_preferred(int64_t M, int64_t N, int64_t K, int act, int has_residual);
int codetr_linear_pp_f16(void *stream, const void *x_dev, const void *w_dev, const void *bias_dev,
                         const void *residual_dev, void *y_dev, int64_t M, int64_t N, int64_t K, int act, int flags);
int codetr_linear_pp_bf16(void *stream, const void *x_dev, const void *w_dev, const void *bias_dev,
                          const void *residual_dev, void *y_dev, int64_t M, int64_t N, int64_t K, int act, int flags);

/* ------------------------------------------------------------------------------------------
 * Padding-mask pyramid: everything the detection transformer derives from img_masks, one launch.
 *
 * Replaces, per level: F.interpolate(img_masks, size=feat.shape[-2:]).to(bool) (codetr/co_dino_head.py:155),
 * not_mask.cumsum(1) / cumsum(2) (codetr/positional_encoding.py:78-79), the two sums of get_valid_ratio
 * (codetr/transformer.py:384-399) and mask.flatten(1) + cat (codetr/transformer.py:513-520).
 *
 *   img_mask_dev        [B, H_img, W_img], non-zero = padding; mask_elem_bytes = 1 (bool / uint8), 2 (fp16 / bf16)
 *                       or 4 (fp32): the reference hands over a float mask (inferencer.py:354-358) and tests
 *                       `.to(bool)`; -0.0 counts as zero, NaN as non-zero
 *   level_shapes_host   HOST array of 2*num_levels int64: (H_l, W_l) per level (num_levels <= 8)
 *   mask_flat_dev       [B, S] uint8, S = sum H_l*W_l: level masks (nearest-neighbour resize, ATen's index rule),
 *                       levels concatenated -- the transformer's mask_flatten
 *   ycum_dev, xcum_dev  fp32, S*B elements each; level l occupies [B, H_l, W_l] starting at element B*start_l:
 *                       running count of valid pixels down the columns / along the rows (the inputs of
 *                       codetr_sine_pos_tokens_f16)
 *   valid_counts_dev    [B, num_levels, 2] fp32: valid pixels in the first row (w) and first column (h); the host
 *                       divides by (W_l, H_l) in the model dtype to get valid_ratios
 * ------------------------------------------------------------------------------------------ */
int codetr_mask_pyramid(void *stream, const void *img_mask_dev, int64_t B, int64_t H_img, int64_t W_img,
                        int num_levels, const int64_t *level_shapes_host, void *mask_flat_dev, float *ycum_dev,
                        float *xcum_dev, float *valid_counts_dev, int mask_elem_bytes);

/* ------------------------------------------------------------------------------------------
 * Decoder query positions: sigmoid + valid-ratio scaling of the reference boxes and their sine embedding.
 *
 * Replaces the head of each DinoTransformerDecoder layer iteration, codetr/transformer.py:208-217
 * (reference_points_input = reference_points[:, :, None].sigmoid() * cat([valid_ratios, valid_ratios], -1)[:, None];
 * query_sine_embed = gen_sineembed_for_position(reference_points_input[:, :, 0, :], embed_dims // 2)) and
 * gen_sineembed_for_position, codetr/transformer.py:157-190 -- 27 ATen launches per layer.
 *
 *   ref_dev           [B, Nq, ref_dim] f16, ref_dim 2 or 4: reference points, unactivated (apply_sigmoid = 1)
 *                     or already in [0, 1] (apply_sigmoid = 0)
 *   valid_ratios_dev  [B, num_levels, 2] f16 (w, h)
 *   ref_in_dev        out [B, Nq, num_levels, ref_dim] f16: the MSDA reference points of the layer
 *   embed_dev         out [B, Nq, ref_dim * pos_feat] f16: blocks ordered (y, x[, w, h]); within a block channel
 *                     2k = sin, 2k+1 = cos of  v * 2 pi / temperature^(2k / pos_feat),  v = ref_in[b, q, 0, coord]
 * fp32 trigonometry on the f16-rounded ref_in; pos_feat % 8 == 0.
 *   valid_ratios32_dev / ref_in32_dev  NULL, or fp32 valid ratios [B, num_levels, 2] and an fp32 output
 *                     [B, Nq, num_levels, ref_dim]: the same reference points with the sigmoid and the scaling kept in
 *                     fp32 (for codetr_msda_fused_forward_ref32_*); the embedding is then taken from these.
 * ------------------------------------------------------------------------------------------ */
int codetr_query_sine_embed_f16(void *stream, const void *ref_dev, const void *valid_ratios_dev,
                                const float *valid_ratios32_dev, int64_t B, int64_t Nq, int ref_dim, int num_levels,
                                int pos_feat, float temperature, int apply_sigmoid, void *ref_in_dev,
                                float *ref_in32_dev, void *embed_dev);
int codetr_query_sine_embed_bf16(void *stream, const void *ref_dev, const void *valid_ratios_dev,
                                const float *valid_ratios32_dev, int64_t B, int64_t Nq, int ref_dim, int num_levels,
                                int pos_feat, float temperature, int apply_sigmoid, void *ref_in_dev,
                                float *ref_in32_dev, void *embed_dev);

/* ------------------------------------------------------------------------------------------
 * Token geometry of the deformable encoder / two-stage proposal head, one launch (f16 path).
 *
 * Replaces get_reference_points (codetr/transformer.py:280-305), reference_points[:, :, None] * valid_ratios[:, None]
 * (:530), make_encoder_output_proposals_export (:331-339) and the proposal half of
 * apply_mask_to_proposal_and_memory (:351-380); its memory half (`memory * total_mask`) is expressed as row
 * state 2 for codetr_linear_*'s row mask on enc_output.
 *
 *   valid_ratios_dev        [B, L, 2] f16 (w, h)
 *   mask_flat_dev           [B, S] uint8, non-zero = padding (codetr_mask_pyramid's output)
 *   level_shapes_host       HOST array of 2*L int64 (H_l, W_l)
 *   reference_points_dev    out [B, S, 2] f16 (x, y), the reference's own fp16 roundings
 *   reference_by_level_dev  out [B, S, L, 2] f16
 *   proposals_dev           out [B, S, 4] f16: logit of (x, y, 0.05*2^l, 0.05*2^l) where the token is kept,
 *                           finfo(f16).max (NaN where the logit is not finite) where it is dropped
 *   row_state_dev           out [B, S] uint8: 0 = kept, 2 = dropped (padding, or a logit outside (-4.6, 4.6))
 * ------------------------------------------------------------------------------------------ */
int codetr_encoder_geometry_f16(void *stream, const void *valid_ratios_dev, const void *mask_flat_dev, int64_t B,
                                int num_levels, const int64_t *level_shapes_host, void *reference_points_dev,
                                void *reference_by_level_dev, void *proposals_dev, void *row_state_dev);
int codetr_encoder_geometry_bf16(void *stream, const void *valid_ratios_dev, const void *mask_flat_dev, int64_t B,
                                int num_levels, const int64_t *level_shapes_host, void *reference_points_dev,
                                void *reference_by_level_dev, void *proposals_dev, void *row_state_dev);

/* out[r] = max_c x[r, c] (NaN propagates, as torch.max): enc_outputs_class.max(-1)[0], the ranking score of
 * the two-stage top-k (codetr/transformer.py:560).  x [rows, C] f16 dense, out [rows] f16. */
int codetr_row_max_f16(void *stream, const void *x_dev, void *out_dev, int64_t rows, int64_t C);
int codetr_row_max_bf16(void *stream, const void *x_dev, void *out_dev, int64_t rows, int64_t C);

/* ------------------------------------------------------------------------------------------
 * Pre- and post-processing either side of CoDETR.forward (the reference's Inferencer, codetr/inferencer.py).
 *
 * codetr_preprocess_u8_*: one image, uint8 HWC RGB on the device -> normalised CHW in the model dtype + padding mask.
 *   Replaces the CPU test pipeline + DetDataPreprocessor + mask loop (codetr/inferencer.py:439-452, 354-358; configs
 *   co_dino_5scale_swin_l_16xb1_16e_o365tococo.py:88-96): mmdet Resize(keep_ratio) = cv2.resize(INTER_LINEAR) with
 *   OpenCV's 8-bit fixed-point arithmetic, Pad(size, pad_val) to the right / bottom, (x - mean) / std in fp32.
 *     src_dev            [H_src, W_src, 3] uint8
 *     H_resized/W_resized  size of the resized image (host: int(dim * factor + 0.5), mmcv.imrescale)
 *     H_pad/W_pad        output size (>= resized size)
 *     mean_host/std_host/pad_value_host   HOST arrays of 3 (per channel, in 0..255 units; pad value is a uint8 pixel
 *                        value that is normalised like the image, as Pad runs before the normalisation)
 *     dst_dev            [3, H_pad, W_pad]; mask_dev [H_pad, W_pad] (same dtype, 0 inside the image, 1 in the padding)
 *                        or NULL
 * codetr_batched_nms_f32: torchvision.ops.batched_nms of postprocess_predictions (codetr/inferencer.py:388-398):
 *   greedy per-class hard NMS, IoU > iou_threshold suppresses.  boxes_sorted_dev [N, 4] fp32 xyxy and
 *   labels_sorted_dev [N] int64 in DESCENDING SCORE ORDER; keep_dev [N] uint8 out (1 = kept).  One workgroup (N is
 *   <= 300 on this path; up to 60 000 accepted).
 * ------------------------------------------------------------------------------------------ */
int codetr_preprocess_u8_f16(void *stream, const void *src_dev, int64_t H_src, int64_t W_src, int64_t H_resized,
                             int64_t W_resized, int64_t H_pad, int64_t W_pad, const float *mean_host,
                             const float *std_host, const int *pad_value_host, void *dst_dev, void *mask_dev);
int codetr_preprocess_u8_f32(void *stream, const void *src_dev, int64_t H_src, int64_t W_src, int64_t H_resized,
                             int64_t W_resized, int64_t H_pad, int64_t W_pad, const float *mean_host,
                             const float *std_host, const int *pad_value_host, void *dst_dev, void *mask_dev);
int codetr_batched_nms_f32(void *stream, const float *boxes_sorted_dev, const int64_t *labels_sorted_dev, int64_t N,
                           float iou_threshold, void *keep_dev);

/* ------------------------------------------------------------------------------------------
 * Backward of multi-scale deformable attention (training path).
 *
 * Replaces ms_deformable_col2im_cuda<T> / ms_deform_attn_backward (codetr/csrc/ms_deform_attn.cu:781-897, 975-1028;
 * arithmetic of ms_deform_attn_col2im_bilinear :79-146): same operands as the forward plus
 *   grad_output_dev        [B, Nq, M*D]          T
 *   grad_value_dev         [B, S, M, D]          T   accumulated with atomics -- the CALLER zero-fills all three
 *   grad_sampling_loc_dev  [B, Nq, M, L, P, 2]   T   gradients first (reference codetr/ops.py:94-96)
 *   grad_attn_weight_dev   [B, Nq, M, L, P]      T
 * T in {f16, f32, f64}; f16 computes in fp32 and accumulates grad_value with packed f16 atomics.  D must be a
 * power of two <= 64 (>= 2 for f16).  im2col_step: the reference's contract (batch % min(batch, step) == 0).
 * ------------------------------------------------------------------------------------------ */
int codetr_msda_backward_f16(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                             const int64_t *level_start_dev, const void *sampling_loc_dev, const void *attn_weight_dev,
                             const void *grad_output_dev, int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P,
                             int64_t im2col_step, void *grad_value_dev, void *grad_sampling_loc_dev,
                             void *grad_attn_weight_dev);
int codetr_msda_backward_f32(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                             const int64_t *level_start_dev, const void *sampling_loc_dev, const void *attn_weight_dev,
                             const void *grad_output_dev, int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P,
                             int64_t im2col_step, void *grad_value_dev, void *grad_sampling_loc_dev,
                             void *grad_attn_weight_dev);
int codetr_msda_backward_f64(void *stream, const void *value_dev, const int64_t *spatial_shapes_dev,
                             const int64_t *level_start_dev, const void *sampling_loc_dev, const void *attn_weight_dev,
                             const void *grad_output_dev, int64_t B, int64_t S, int M, int D, int L, int64_t Nq, int P,
                             int64_t im2col_step, void *grad_value_dev, void *grad_sampling_loc_dev,
                             void *grad_attn_weight_dev);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension: y[r,:] = (x[r,:] - mean) * rsqrt(var + eps) * gamma + beta
 *
 * Replaces at::native::layer_norm behind every nn.LayerNorm of the hot path (built through mmcv's
 * build_norm_layer at codetr/swin.py:331,345,627 and codetr/transformer_mmcv.py:647; nn.LayerNorm at
 * codetr/transformer.py:153,452).  x, y [rows, C] dense row-major (y may alias x); gamma, beta [C];
 * fp32 statistics (two-pass), one rounding at the store.  C % 8 == 0, C <= 4096.
 * ------------------------------------------------------------------------------------------ */
int codetr_layernorm_f16(void *stream, const void *x_dev, const void *gamma_dev, const void *beta_dev, void *y_dev,
                         int64_t rows, int64_t C, float eps);
int codetr_layernorm_bf16(void *stream, const void *x_dev, const void *gamma_dev, const void *beta_dev, void *y_dev,
                          int64_t rows, int64_t C, float eps);

/* Swin PatchMerging's gather + LayerNorm in one pass (codetr/transformer_mmcv.py:213-316: nn.Unfold(2, stride 2) on
 * the token map, then LayerNorm(4C)): x_dev [B, H, W, C] f16 token map, y_dev [B, ceil(H/2)*ceil(W/2), 4C] f16 with
 * the 4C axis ordered (ky, kx, c) -- gamma / beta (and the reduction Linear's columns) must be in that order, i.e.
 * permuted from nn.Unfold's (c, ky, kx); zeros beyond an odd H / W.  C % 8 == 0, 4C <= 4096. */
int codetr_patch_merge_layernorm_f16(void *stream, const void *x_dev, const void *gamma_dev, const void *beta_dev,
                                     void *y_dev, int64_t B, int64_t H, int64_t W, int64_t C, float eps);
int codetr_patch_merge_layernorm_bf16(void *stream, const void *x_dev, const void *gamma_dev, const void *beta_dev,
                                     void *y_dev, int64_t B, int64_t H, int64_t W, int64_t C, float eps);  /* bf16 storage, same contract */

/* ------------------------------------------------------------------------------------------
 * Fused (shifted-)window multi-head self-attention of one Swin block.
 *
 * Replaces the tensor program of ShiftWindowMSA.forward + WindowMSA.forward between the qkv and
 * the proj Linear (reference codetr/swin.py:191-252 and :92-112): pad-to-window, roll(-shift),
 * window_partition, q*scale, q@k^T, + relative position bias, + shift mask (-100), softmax,
 * attn@v, head merge, window_reverse, roll(+shift), crop.
 *
 *   qkv_dev      [B, H*W, 3*C]  f16  output of the qkv Linear on the UNPADDED, unrolled token map;
 *                                    channel order (q | k | v), each head-major (C = num_heads*head_dim)
 *   qkv_bias_dev [3*C]          f16  bias of that Linear (what a zero pad token turns into; zeros
 *                                    if the layer has no bias) -- pad tokens take part in the
 *                                    softmax as keys, exactly as in the reference
 *   rel_bias_dev [num_heads, N, N] f16  gathered relative position bias, N = window_size^2
 *   out_dev      [B, H*W, C]    f16  attention output in the same spatial token order
 *   shift        0 (W-MSA) or window_size/2 (SW-MSA); head_dim must be 32; window_size in {4,7,8,12}
 *
 * fp32 scores / softmax / accumulation; probabilities rounded to f16 (bf16 in the _bf16 form, all tensors bf16) for
 * the P.V product.
 * ------------------------------------------------------------------------------------------ */
int codetr_window_attention_f16(void *stream, const void *qkv_dev, const void *qkv_bias_dev,
                                const void *rel_bias_dev, void *out_dev, int64_t B, int64_t H, int64_t W,
                                int num_heads, int head_dim, int window_size, int shift);
int codetr_window_attention_bf16(void *stream, const void *qkv_dev, const void *qkv_bias_dev,
                                 const void *rel_bias_dev, void *out_dev, int64_t B, int64_t H, int64_t W,
                                 int num_heads, int head_dim, int window_size, int shift);
/* The same attention with the output emitted as OCP e4m3 bytes, out8[i] = sat(f16(o[i]) / out_scale): the operand of
 * the fp8 proj GEMM (BASELINE config 5) without the 16-bit tensor's round trip: codetr_cast_fp8_f16 applied to the
 * kernel's own fp16 result (tests: equal to the two-kernel path but for ~1 element per million on a rounding boundary).  out8_dev [B, H*W, C] bytes, 4-byte aligned. */
int codetr_window_attention_fp8out_f16(void *stream, const void *qkv_dev, const void *qkv_bias_dev,
                                       const void *rel_bias_dev, void *out8_dev, float out_scale, int64_t B,
                                       int64_t H, int64_t W, int num_heads, int head_dim, int window_size, int shift);
/* Round 6: every form above through ONE entry, plus the layout of the relative-position bias.
 *   elem        0: fp16 tensors, 1: bf16 (out_mode 0 only)
 *   out_mode    0: 16-bit output (codetr_window_attention_f16 / _bf16); 1: e4m3 = sat(f16(o) / out_scale)
 *               (codetr_window_attention_fp8out_f16); 2: e4m3 + MX block scales in out_scales_dev
 *               (codetr_window_attention_fp8mx_f16).  out_scales_dev / out_scale are ignored by the modes that do not use them.
 *   bias_layout 0: rel_bias_dev is the reference's gathered table [nH, N, N] (swin.py:92-112: table[index].view(N, N, nH)
 *               permuted), as above; 1: the same values in the LANE ORDER of the kernel's score tiles -- row q of head h
 *               holds at position j the bias of key idx[j], idx from codetr_window_attention_bias_index (window sizes 12,
 *               8, 4: whole 16-key tiles).  A lane's values of a query tile are then contiguous (72 bytes for the 12 x 12
 *               window: five loads instead of nine 8-byte pieces 32 bytes apart), results are bit-identical to layout 0,
 *               launches 4-12 % shorter (profiles/r06_window_attention.txt).  The table is a per-layer constant: the host
 *               permutes it once (codetr/swin.py: WindowMSA.relative_position_bias).
 * codetr_window_attention_bias_index: idx_host[window_size^2]; CODETR_E_UNSUPPORTED for window sizes without the lane order. */
int codetr_window_attention_ex(void *stream, const void *qkv_dev, const void *qkv_bias_dev, const void *rel_bias_dev,
                               void *out_dev, void *out_scales_dev, float out_scale, int64_t B, int64_t H, int64_t W,
                               int num_heads, int head_dim, int window_size, int shift, int elem, int out_mode,
                               int bias_layout);
int codetr_window_attention_bias_index(int window_size, int32_t *idx_host);

/* ------------------------------------------------------------------------------------------
 * Dense multi-head softmax attention, head_dim 32: the core of nn.MultiheadAttention(256, 8) in the decoder's
 * self-attention (reference codetr/transformer_mmcv.py:394-428, called from DetrTransformerDecoderLayer,
 * codetr/transformer.py:233-277) between its in- and out-projections, which are codetr_linear_* calls:
 *   out[b, i, h*32 + c] = sum_j softmax_j(q[b,i,h,:] . k[b,j,h,:] / sqrt(32)) * v[b, j, h*32 + c]
 *   q_dev   [B, Nq, >= H*32]  rows q_row_stride elements apart (q and k may be column ranges of one fused projection)
 *   k_dev, v_dev [B, Nk, ...] likewise;  out_dev [B, Nq, ...] rows out_row_stride apart
 * No attention / key-padding mask (the inference path has none).  fp32 scores, online softmax over 128-key chunks,
 * fp32 accumulation.  Nk <= 1024 (K and V of one head live in LDS), head_dim == 32: CODETR_E_UNSUPPORTED otherwise.
 * ------------------------------------------------------------------------------------------ */
int codetr_mha_attention_f16(void *stream, const void *q_dev, const void *k_dev, const void *v_dev, void *out_dev,
                             int64_t B, int64_t Nq, int64_t Nk, int num_heads, int head_dim, int64_t q_row_stride,
                             int64_t k_row_stride, int64_t v_row_stride, int64_t out_row_stride);
int codetr_mha_attention_bf16(void *stream, const void *q_dev, const void *k_dev, const void *v_dev, void *out_dev,
                              int64_t B, int64_t Nq, int64_t Nk, int num_heads, int head_dim, int64_t q_row_stride,
                              int64_t k_row_stride, int64_t v_row_stride, int64_t out_row_stride);

/* ------------------------------------------------------------------------------------------
 * One DINO decoder layer's row-local work in one launch (csrc/decoder_layer.hip), f16, embed_dims 256, 8 heads of 32,
 * 4-d reference boxes, post-norm layer (self_attn, norm, cross_attn, norm, ffn, norm) with box refinement.
 * Replaces, per layer, the ~19 launches of reference codetr/transformer.py:193-230 (DinoTransformerDecoder.forward:
 * reference-point scaling, gen_sineembed_for_position :157-190, ref_point_head, the layer, reg_branches[lid]) and
 * :233-277 / transformer_mmcv.py:583-749 (the layer's operation_order) -- everything except softmax(QK^T)V of the
 * self-attention, which stays codetr_mha_attention_f16 between two of these launches:
 *   TAIL (attn_dev != NULL): x1 = LN1(x + attn.Wo^T + bo); (offsets | logits) = (x1 + qpos).Wol^T + bol;
 *        s = MSDA(value, softmax(logits), ref_xy + off / P * ref_wh / 2) with ref = sigmoid(ref_dev) * valid ratios (fp32);
 *        x2 = LN2(x1 + s.Wout^T + bout); x3 = LN3(x2 + relu(x2.W1^T + b1).W2^T + b2); ref_out = ref + reg_branch(x3)
 *   HEAD (head_w_dev != NULL): qpos_out = ref_point_head(sine_embed(sigmoid(ref') * valid_ratios[level 0]));
 *        qk_out = (x3 + qpos_out).Wqk^T + bqk; v_out = x3.Wv^T + bv       (ref' = ref_out, or ref_dev without a TAIL;
 *        x3 = x_dev without a TAIL)
 *   x_out = x3 with a HEAD, else final_norm(x3) (the decoder's output LayerNorm, final_norm_dev = gamma | beta).
 * Tensors: x, attn, qpos, x_out, qpos_out, v_out [B*Nq, 256]; qk_out [B*Nq, 512]; ref, ref_out [B*Nq, 4] unactivated;
 * valid_ratios32 [B, L, 2] fp32; value [B, S, 8, 32] (the layer's projected, masked value map); spatial_shapes [L, 2],
 * level_start [L] int64 on the device.  Packed f16 weights: matrices first, FRAGMENT-MAJOR -- a [N][K] nn.Linear weight
 * is stored as N/16 x K/32 blocks of 64 x 8 halfs, block (tile, ks) at ((tile * K/32 + ks) * 64 + lane) * 8 with
 * lane = 16 g + r holding W[16 tile + r][32 ks + 8 g .. + 7] (one wave load = 1 KB of whole lines, the MFMA operand
 * layout) -- then the small vectors, which the kernel copies to LDS once; in this order:
 *   tail_w: Wo, Wol (offsets rows then logits rows: 8*L*P*2 + 8*L*P, zero rows up to 512), Wout, W1 [2048][256],
 *           W2 [256][2048], Wr1, Wr2, Wr3 (4 rows + 12 zero rows) | bo, LN1 gamma, beta, bol, bout, LN2 gamma, beta, b1,
 *           b2, LN3 gamma, beta, br1, br2, br3 (4 values + 4 of padding)
 *   head_w: Wqk [512][256], Wv [256][256] | bqk, bv       pos_w: Wp1 [256][512], Wp2 | bp1, bp2      final_norm: gamma, beta
 * codetr_decoder_layer_blob_halfs(which = 0 tail | 1 head | 2 pos | 3 final norm, ...) returns the element counts.
 * Rounding points are those of the separate kernels (f16 wherever they materialise a tensor; fp32 statistics, softmax,
 * sampling arithmetic and accumulation); the results differ from them only by summation order.
 * CODETR_E_UNSUPPORTED outside L*P <= 32, hidden == 2048, (8*L*P*3) % 16 == 0 and <= 512.
 * ------------------------------------------------------------------------------------------ */
int codetr_decoder_layer_supported(int embed_dims, int num_heads, int num_levels, int num_points, int hidden, int ref_dim,
                                   int pos_feat);
int64_t codetr_decoder_layer_blob_halfs(int which, int num_levels, int num_points, int hidden);
int codetr_decoder_layer_f16(void *stream, const void *x_dev, const void *attn_dev, const void *qpos_dev,
                             const void *ref_dev, const float *valid_ratios32_dev, const void *value_dev,
                             const int64_t *spatial_shapes_dev, const int64_t *level_start_dev, const void *tail_w_dev,
                             const void *pos_w_dev, const void *head_w_dev, const void *final_norm_dev, void *x_out_dev,
                             void *ref_out_dev, void *qpos_out_dev, void *qk_out_dev, void *v_out_dev, int64_t B,
                             int64_t Nq, int64_t S, int num_levels, int num_points, int hidden, float ln_eps,
                             float temperature);
/* bf16 twin: bf16 activations / weights / outputs, v_mfma_f32_16x16x32_bf16, the same fp32 arithmetic in between (the
 * same source compiled with bf16 storage, csrc/decoder_layer_bf16.hip). */
int codetr_decoder_layer_bf16(void *stream, const void *x_dev, const void *attn_dev, const void *qpos_dev,
                             const void *ref_dev, const float *valid_ratios32_dev, const void *value_dev,
                             const int64_t *spatial_shapes_dev, const int64_t *level_start_dev, const void *tail_w_dev,
                             const void *pos_w_dev, const void *head_w_dev, const void *final_norm_dev, void *x_out_dev,
                             void *ref_out_dev, void *qpos_out_dev, void *qk_out_dev, void *v_out_dev, int64_t B,
                             int64_t Nq, int64_t S, int num_levels, int num_points, int hidden, float ln_eps,
                             float temperature);

/* ------------------------------------------------------------------------------------------
 * GroupNorm on token-major activations, written into a slice of the flattened multi-level map.
 *
 * Replaces nn.GroupNorm(32, 256) of the ChannelMapper neck (mmdet v3.3.0, built at
 * codetr/codetr.py:53-54 from configs lsj:40-47) together with the flatten / transpose / cat copies
 * that turn the NCHW levels into the encoder's [B, S, C] input (codetr/transformer.py:508-519).
 *
 *   x_dev   [B, HW, C] f16 (output of the level's 1x1 conv run as a linear over tokens)
 *   out_dev points at row `level_start` of image 0 in the [B, S, C] destination;
 *           out_batch_stride = S*C elements between images
 *   workspace_dev: codetr_groupnorm_tokens_workspace_bytes(B, HW, C) bytes of device scratch
 *   groups must equal C/8 (8 channels = one 16-byte chunk per group: GN(32) on 256 channels)
 * fp32 partial sums per 512-row slab, fp64 final reduction, statistics over (HW x 8) per (image, group).
 * ------------------------------------------------------------------------------------------ */
int64_t codetr_groupnorm_tokens_workspace_bytes(int64_t B, int64_t HW, int64_t C);
int codetr_groupnorm_tokens_f16(void *stream, const void *x_dev, const void *gamma_dev, const void *beta_dev,
                                void *out_dev, int64_t out_batch_stride, void *workspace_dev, int64_t B, int64_t HW,
                                int64_t C, int groups, float eps);
int codetr_groupnorm_tokens_bf16(void *stream, const void *x_dev, const void *gamma_dev, const void *beta_dev,
                                void *out_dev, int64_t out_batch_stride, void *workspace_dev, int64_t B, int64_t HW,
                                int64_t C, int groups, float eps);  /* bf16 storage, same contract */

/* ------------------------------------------------------------------------------------------
 * Sine positional encoding of one pyramid level, written into a slice of lvl_pos_embed [B, S, 2*num_feats].
 *
 * Replaces SinePositionalEncoding.forward (codetr/positional_encoding.py:58-93) and the flatten / transpose /
 * "+ level_embeds[lvl]" / cat steps of CoDinoTransformer.forward (codetr/transformer.py:508-519).
 *   ycum_dev, xcum_dev [B, H, W] f32: running sums of (1 - mask) along y and along x (host: two cumsum calls)
 *   level_embed_dev    [2*num_feats] f16 or NULL
 *   out_dev            row `level_start` of image 0 in the [B, S, 2*num_feats] destination; out_batch_stride =
 *                      S * 2*num_feats elements between images
 * channel layout [pos_y | pos_x], inside each: channel 2f = sin(e / T^(2f/num_feats)), 2f+1 = cos(same),
 * e = (cum + offset) / (last + eps) * scale when `normalize`, else cum.  fp32 arithmetic, f16 result.
 * ------------------------------------------------------------------------------------------ */
int codetr_sine_pos_tokens_f16(void *stream, const float *ycum_dev, const float *xcum_dev, const void *level_embed_dev,
                               void *out_dev, int64_t out_batch_stride, int64_t B, int64_t H, int64_t W, int num_feats,
                               float temperature, float scale, float eps, float offset, int normalize);
int codetr_sine_pos_tokens_bf16(void *stream, const float *ycum_dev, const float *xcum_dev, const void *level_embed_dev,
                               void *out_dev, int64_t out_batch_stride, int64_t B, int64_t H, int64_t W, int num_feats,
                               float temperature, float scale, float eps, float offset, int normalize);  /* bf16 storage, same contract */

/* ------------------------------------------------------------------------------------------
 * Fused transformer FFN:  y = x + relu(x . w1^T + b1) . w2^T + b2      (hidden activation never leaves the CU)
 *
 * Replaces FFN.forward (codetr/transformer_mmcv.py:484-500: Linear, ReLU, Linear, + identity) of the deformable
 * encoder / DINO decoder layers (configs lsj:72-79, 92-99: embed_dims 256, feedforward_channels 2048).
 *   x_dev [M, 256]; w1_dev [hidden, 256]; b1_dev [hidden]; b2_dev [256]; y_dev [M, 256]; all f16
 *   w2_packed_dev [256, hidden]: the second Linear's weight passed ONCE through codetr_ffn_pack_w2_f16 (a column
 *   permutation inside every 64-block that puts the hidden units in the order the first product's accumulators
 *   hold them, so that those accumulators feed the second product without leaving registers)
 * C_in must be 256, hidden a multiple of 64; all device pointers 16-byte aligned (CODETR_E_BADARG otherwise).  fp32
 * accumulation; the hidden activation is rounded to f16 between the two products and the FFN output once more before
 * the residual add (as the two-kernel f16 path does).
 * ------------------------------------------------------------------------------------------ */
int codetr_ffn_relu_f16(void *stream, const void *x_dev, const void *w1_dev, const void *b1_dev,
                        const void *w2_packed_dev, const void *b2_dev, void *y_dev, int64_t M, int64_t C_in,
                        int64_t hidden);

/* The same kernel with the layer's next two steps folded into its epilogue (post-norm encoder layer,
 * operation_order (..., 'ffn', 'norm'), codetr/transformer_mmcv.py:649-749, and the `query + query_pos` at the head
 * of the next layer's attention, codetr/multi_scale_deformable_attention.py:161-162):
 *   y      = LayerNorm(x + ffn(x)) * gamma + beta        ln_gamma_dev / ln_beta_dev [256] f16, both or neither
 *   y_plus_pos = y + pos                                  pos_dev / y_plus_pos_dev [M, 256] f16, both or neither
 * The LayerNorm arithmetic is that of codetr_layernorm_f16 (fp32 two-pass statistics over the f16-rounded sum), summed
 * in the accumulators' lane order: equal to running the three kernels one after another up to the last f16 bit of a
 * few outputs (< 2 %); y_plus_pos is exactly y + pos. */
int codetr_ffn_relu_ln_f16(void *stream, const void *x_dev, const void *w1_dev, const void *b1_dev,
                           const void *w2_packed_dev, const void *b2_dev, void *y_dev, int64_t M, int64_t C_in,
                           int64_t hidden, const void *ln_gamma_dev, const void *ln_beta_dev, float ln_eps,
                           const void *pos_dev, void *y_plus_pos_dev);

/* ... and with the layer's PRECEDING LayerNorm folded in as well (operation_order (..., 'norm', 'ffn', 'norm')): the
 * kernel normalises its input rows in registers (fp32 two-pass statistics, f16 result = the MFMA operand and the
 * identity), so that norm's output is never written:
 *   x1 = LayerNorm(x) * ln_in_gamma + ln_in_beta;  y = LayerNorm(x1 + ffn(x1)) * gamma + beta;  y_plus_pos = y + pos
 * ln_in_gamma_dev / ln_in_beta_dev [256] f16, both or neither (NULL: codetr_ffn_relu_ln_f16). */
int codetr_ffn_relu_ln2_f16(void *stream, const void *x_dev, const void *w1_dev, const void *b1_dev,
                            const void *w2_packed_dev, const void *b2_dev, void *y_dev, int64_t M, int64_t C_in,
                            int64_t hidden, const void *ln_in_gamma_dev, const void *ln_in_beta_dev, float ln_in_eps,
                            const void *ln_gamma_dev, const void *ln_beta_dev, float ln_eps, const void *pos_dev,
                            void *y_plus_pos_dev);
/* The fused FFN on the e4m3 matrix path (BASELINE config 5; no reference counterpart for 8-bit arithmetic -- the layer
 * is the same FFN, codetr/transformer_mmcv.py:484-500, inside the same post-norm encoder layer :709-749):
 *   x1 = LayerNorm_in(x) (or x);  xq = sat(x1 / x_scale) -> e4m3
 *   h  = relu((xq . w1q^T) * w1_scale[j] * x_scale + b1[j]);  hq = sat(h / h_scale) -> e4m3
 *   y  = LayerNorm(x1 + (hq . w2q^T) * w2_scale[n] * h_scale + b2[n]) * gamma + beta;  y_plus_pos = y + pos
 * x / y / pos / biases / LayerNorm parameters f16 as in codetr_ffn_relu_ln2_f16; w1q_dev [hidden, 256] e4m3 bytes with
 * per-hidden-unit scales w1_scale_dev [hidden] fp32; w2q_packed_dev [256, hidden] e4m3 bytes with per-output-channel
 * scales w2_scale_dev [256] fp32, its columns permuted inside every 128-block: packed column 32 g + 4 t + r holds
 * hidden unit 16 t + 4 g + r of the block (g < 4, t < 8, r < 4 -- the order in which the first product's accumulators
 * become the second product's operand).  x_scale / h_scale: static per-tensor activation scales from calibration.
 * fp32 accumulation with v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales); conversions saturate at +-448.
 * C_in must be 256, hidden a multiple of 128 and <= 2048 (CODETR_E_UNSUPPORTED otherwise); x, y, pos, the weights and
 * the LayerNorm parameters 16-byte aligned (CODETR_E_BADARG). */
int codetr_ffn_fp8(void *stream, const void *x_f16_dev, const void *w1q_dev, const float *w1_scale_dev,
                   const void *b1_f16_dev, const void *w2q_packed_dev, const float *w2_scale_dev,
                   const void *b2_f16_dev, void *y_f16_dev, int64_t M, int64_t C_in, int64_t hidden, float x_scale,
                   float h_scale, const void *ln_in_gamma_dev, const void *ln_in_beta_dev, float ln_in_eps,
                   const void *ln_gamma_dev, const void *ln_beta_dev, float ln_eps, const void *pos_dev,
                   void *y_plus_pos_dev);

/* bf16 storage form of the same kernel (all tensors bf16; the packed W2 comes from codetr_ffn_pack_w2_f16, which only
 * moves 16-bit elements) */
int codetr_ffn_relu_ln2_bf16(void *stream, const void *x_dev, const void *w1_dev, const void *b1_dev,
                            const void *w2_packed_dev, const void *b2_dev, void *y_dev, int64_t M, int64_t C_in,
                            int64_t hidden, const void *ln_in_gamma_dev, const void *ln_in_beta_dev, float ln_in_eps,
                            const void *ln_gamma_dev, const void *ln_beta_dev, float ln_eps, const void *pos_dev,
                            void *y_plus_pos_dev);
/* one-time weight pre-pack for the call above: w2_dev [C_out, hidden] f16 -> w2_packed_dev (same shape) */
int codetr_ffn_pack_w2_f16(void *stream, const void *w2_dev, void *w2_packed_dev, int64_t C_out, int64_t hidden);
/* The post-norm encoder layer from the attention output to the layer output as ONE launch (reference
 * codetr/transformer_mmcv.py: BaseTransformerLayer.forward, operation_order self_attn - norm - ffn - norm, with the tail of
 * MultiScaleDeformableAttention.forward, multi_scale_deformable_attention.py: `output_proj(output) + identity`):
 *     x0 = identity + E(attn @ Wo^T + bo)       (E: rounded to the storage type, as the separate GEMM's epilogue does)
 *     x1 = LayerNorm_in(x0)                      (ln_in_* NULL: x1 = x0)
 *     y  = LayerNorm(x1 + relu(x1 @ W1^T + b1) @ W2^T + b2),   y_plus_pos = y + pos (optional, as above)
 * attn_dev / identity_dev / y_dev [M, 256]; wo_dev [256, 256], bo_dev [256]: nn.Linear parameters as they are;
 * w1_perm_dev [hidden, 256]: W1 with its COLUMNS reordered by codetr_ffn_oproj_w1_index (column j <- column idx[j]; the
 * kernel produces x1 in its epilogue's lane layout and multiplies it as it is); w2_packed_dev from
 * codetr_ffn_pack_w2_f16.  Replaces a GEMM launch that reads attn and identity and writes x0, and the FFN kernel's read of
 * x0: 1.7 GB -> 0.84 GB of HBM traffic per layer at four 1920x1280 images, for 6 % more MFMA work.  Same argument checks
 * as codetr_ffn_relu_ln2_f16. */
int codetr_ffn_oproj_relu_ln2_f16(void *stream, const void *attn_dev, const void *wo_dev, const void *bo_dev,
                                  const void *identity_dev, const void *w1_perm_dev, const void *b1_dev,
                                  const void *w2_packed_dev, const void *b2_dev, void *y_dev, int64_t M, int64_t C_in,
                                  int64_t hidden, const void *ln_in_gamma_dev, const void *ln_in_beta_dev,
                                  float ln_in_eps, const void *ln_gamma_dev, const void *ln_beta_dev, float ln_eps,
                                  const void *pos_dev, void *y_plus_pos_dev);
int codetr_ffn_oproj_relu_ln2_bf16(void *stream, const void *attn_dev, const void *wo_dev, const void *bo_dev,
                                   const void *identity_dev, const void *w1_perm_dev, const void *b1_dev,
                                   const void *w2_packed_dev, const void *b2_dev, void *y_dev, int64_t M, int64_t C_in,
                                   int64_t hidden, const void *ln_in_gamma_dev, const void *ln_in_beta_dev,
                                   float ln_in_eps, const void *ln_gamma_dev, const void *ln_beta_dev, float ln_eps,
                                   const void *pos_dev, void *y_plus_pos_dev);
/* idx_host[C_in] (C_in = 256): source column of W1 for every column of w1_perm_dev.  Host-only. */
int codetr_ffn_oproj_w1_index(int64_t C_in, int32_t *idx_host);

#ifdef __cplusplus
}
#endif
#endif /* CODETR_HIP_H_ */
