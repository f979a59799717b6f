/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement, in plain C, of the reference's multi-scale deformable attention
 * forward.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the product (co-detr-tensorrt_amd/) never does.
 *
 * What it follows (all paths relative to the reference tree):
 *   codetr/csrc/ms_deform_attn.cu:31-77    bilinear sampler: floor, 4 corner reads with
 *                                          per-corner bounds tests, weights hh*hw, hh*lw,
 *                                          lh*hw, lh*lw, value layout [S, M, D]
 *   codetr/csrc/ms_deform_attn.cu:211-261  per-output-scalar loop: index decode, level
 *                                          loop, point loop, pixel coords
 *                                          h_im = loc_y*H - 0.5, w_im = loc_x*W - 0.5,
 *                                          range gate (-1, size), accumulate w * sample
 *   codetr/csrc/ms_deform_attn.cu:899-956  dims from tensor shapes, im2col_step chunking
 *                                          (batch % min(batch, step) == 0), zero-filled out
 *
 * Pinning: the reference's CUDA source cannot be built or run in the build image (no
 * nvcc, no NVIDIA GPU), so this restatement is pinned against the reference's own
 * Python formulation of the same op (codetr/ops.py:129-186, F.grid_sample based),
 * imported live from /root/reference by tests/golden/make_golden.py, on the recipes of
 * the reference's tests (tests/test_multi_scale_deformable_attention.py:14-62, 229-364,
 * 417-501).  The reference tests assert CUDA == Python within 1e-15 rel (double), so
 * agreeing with the Python path to that level pins this file to the CUDA kernel too.
 *
 * Arithmetic is carried out in the tensor's own type (float or double), exactly as the
 * templated reference does; half inputs are widened to float by the Python wrapper
 * (the north star judges fp16 against the fp32 path).  One deliberate deviation: the
 * reference calls floorf() on every scalar type (cu:35-36), i.e. a double coordinate is
 * narrowed to float before the floor; here the double instantiation uses floor().  The two
 * differ only when a coordinate sits within 1 float-ulp below an integer, where the
 * bilinear form is continuous, so results agree to ~1e-11 relative in that corner case.
 *
 * Build: see oracle/Makefile  (gcc -O2 -fPIC -shared [-fopenmp]).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define DEFINE_MSDA(NAME, T, FLOOR)                                                              \
  static T NAME##_bilinear(const T *bottom, int height, int width, int nheads, int channels,     \
                           T h, T w, int m, int c) {                                              \
    const int h_low = (int)FLOOR(h);                                                              \
    const int w_low = (int)FLOOR(w);                                                              \
    const int h_high = h_low + 1;                                                                 \
    const int w_high = w_low + 1;                                                                 \
    const T lh = h - (T)h_low, lw = w - (T)w_low;                                                 \
    const T hh = (T)1 - lh, hw = (T)1 - lw;                                                       \
    const int64_t w_stride = (int64_t)nheads * channels;                                          \
    const int64_t h_stride = (int64_t)width * w_stride;                                           \
    const int64_t base = (int64_t)m * channels + c;                                               \
    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                             \
    if (h_low >= 0 && w_low >= 0) v1 = bottom[h_low * h_stride + w_low * w_stride + base];        \
    if (h_low >= 0 && w_high <= width - 1) v2 = bottom[h_low * h_stride + w_high * w_stride + base]; \
    if (h_high <= height - 1 && w_low >= 0) v3 = bottom[h_high * h_stride + w_low * w_stride + base]; \
    if (h_high <= height - 1 && w_high <= width - 1)                                              \
      v4 = bottom[h_high * h_stride + w_high * w_stride + base];                                  \
    const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                               \
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;                                                 \
  }                                                                                               \
                                                                                                  \
  /* returns 0 on success, 1 on the reference's "batch must divide im2col_step" error */         \
  int NAME(const T *value, const int64_t *spatial_shapes, const int64_t *level_start_index,       \
           const T *sampling_loc, const T *attn_weight, int64_t batch, int64_t spatial_size,      \
           int num_heads, int channels, int num_levels, int64_t num_query, int num_point,         \
           int64_t im2col_step, T *out) {                                                         \
    const int64_t step = batch < im2col_step ? batch : im2col_step;                               \
    if (step <= 0 || batch % step != 0) return 1;                                                 \
    const int64_t n = batch * num_query * num_heads * channels;                                   \
    memset(out, 0, (size_t)n * sizeof(T));                                                        \
    const int64_t qid_stride = (int64_t)num_heads * channels;                                     \
    _Pragma("omp parallel for schedule(static)")                                                  \
    for (int64_t index = 0; index < n; ++index) {                                                 \
      int64_t tmp = index;                                                                        \
      const int c_col = (int)(tmp % channels);                                                    \
      tmp /= channels;                                                                            \
      const int64_t sampling_index = tmp;                                                         \
      const int m_col = (int)(tmp % num_heads);                                                   \
      tmp /= num_heads;                                                                           \
      tmp /= num_query;                                                                           \
      const int64_t b_col = tmp;                                                                  \
      int64_t w_ptr = sampling_index * num_levels * num_point;                                    \
      int64_t loc_ptr = w_ptr << 1;                                                               \
      const T *value_b = value + b_col * spatial_size * qid_stride;                               \
      T col = 0;                                                                                  \
      for (int l = 0; l < num_levels; ++l) {                                                      \
        const int64_t start = level_start_index[l];                                               \
        const int sh = (int)spatial_shapes[2 * l];                                                \
        const int sw = (int)spatial_shapes[2 * l + 1];                                            \
        const T *value_l = value_b + start * qid_stride;                                          \
        for (int p = 0; p < num_point; ++p) {                                                     \
          const T loc_w = sampling_loc[loc_ptr];                                                  \
          const T loc_h = sampling_loc[loc_ptr + 1];                                              \
          const T weight = attn_weight[w_ptr];                                                    \
          const T h_im = loc_h * (T)sh - (T)0.5;                                                  \
          const T w_im = loc_w * (T)sw - (T)0.5;                                                  \
          if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)sh && w_im < (T)sw) {                     \
            col += NAME##_bilinear(value_l, sh, sw, num_heads, channels, h_im, w_im, m_col, c_col) * weight; \
          }                                                                                       \
          w_ptr += 1;                                                                             \
          loc_ptr += 2;                                                                           \
        }                                                                                         \
      }                                                                                           \
      out[index] = col;                                                                           \
    }                                                                                             \
    return 0;                                                                                     \
  }

DEFINE_MSDA(msda_ref_forward_f32, float, floorf)
DEFINE_MSDA(msda_ref_forward_f64, double, floor)
