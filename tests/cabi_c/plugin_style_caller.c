/* Plain-C caller in the shape of the reference's TensorRT plugin enqueue (deformable_attention_plugin.cpp:285-355):
 * raw device pointers + a stream in, no torch, no Python.  Runs one fp32 MSDA forward on the GPU through the C ABI
 * and checks it against the CPU oracle (oracle/msda_ref.c, linked as test infrastructure). */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "codetr_hip.h"

/* oracle/msda_ref.c */
int msda_ref_forward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                         const float *sampling_loc, const float *attn_weight, int64_t batch, int64_t spatial_size,
                         int num_heads, int channels, int num_levels, int64_t num_query, int num_point,
                         int64_t im2col_step, float *out);

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      return 10;                                                                  \
    }                                                                             \
  } while (0)

static unsigned rng = 12345u;
static float urand(void) {
  rng = rng * 1664525u + 1013904223u;
  return (float)(rng >> 8) / 16777216.0f;
}

int main(void) {
  enum { B = 2, M = 8, D = 32, L = 3, P = 4, Nq = 77 };
  const int64_t shapes[L][2] = {{12, 20}, {6, 10}, {3, 5}};
  int64_t starts[L], S = 0;
  for (int l = 0; l < L; ++l) {
    starts[l] = S;
    S += shapes[l][0] * shapes[l][1];
  }
  const size_t nv = (size_t)B * S * M * D, nl = (size_t)B * Nq * M * L * P * 2, nw = nl / 2, no = (size_t)B * Nq * M * D;
  float *value = malloc(nv * 4), *loc = malloc(nl * 4), *w = malloc(nw * 4), *out = malloc(no * 4), *ref = malloc(no * 4);
  for (size_t i = 0; i < nv; ++i) value[i] = urand() - 0.5f;
  for (size_t i = 0; i < nl; ++i) loc[i] = urand() * 1.2f - 0.1f; /* some samples leave the image */
  for (size_t i = 0; i < nw; ++i) w[i] = urand() / (L * P);

  void *dv, *dl, *dw, *dout, *dss, *dls;
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  CHECK(hipMalloc(&dv, nv * 4));
  CHECK(hipMalloc(&dl, nl * 4));
  CHECK(hipMalloc(&dw, nw * 4));
  CHECK(hipMalloc(&dout, no * 4));
  CHECK(hipMalloc(&dss, sizeof(shapes)));
  CHECK(hipMalloc(&dls, sizeof(starts)));
  CHECK(hipMemcpy(dv, value, nv * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dl, loc, nl * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw, w, nw * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dss, shapes, sizeof(shapes), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dls, starts, sizeof(starts), hipMemcpyHostToDevice));

  const int rc = codetr_msda_forward_f32((void *)stream, dv, (const int64_t *)dss, (const int64_t *)dls, dl, dw, B, S, M, D,
                                         L, Nq, P, /*im2col_step*/ 64, dout);
  if (rc) {
    fprintf(stderr, "codetr_msda_forward_f32: %s\n", codetr_hip_strerror(rc));
    return 11;
  }
  CHECK(hipStreamSynchronize(stream));
  CHECK(hipMemcpy(out, dout, no * 4, hipMemcpyDeviceToHost));

  if (msda_ref_forward_f32(value, &shapes[0][0], starts, loc, w, B, S, M, D, L, Nq, P, 64, ref)) return 12;
  double worst = 0.0;
  for (size_t i = 0; i < no; ++i) {
    const double d = fabs((double)out[i] - (double)ref[i]);
    if (d > worst) worst = d;
  }
  printf("max |gpu - oracle| = %.3e over %zu outputs\n", worst, no);
  return worst < 1e-5 ? 0 : 13;
}
