// Second probe of v_mfma_scale_f32_16x16x128_f8f6f4: WHICH 32 k-elements of a row does the scale byte of lane (row + 16 b) cover?
// All-ones operands; scale_a x2 in lane 5 + 16 b (row 5 -> 160); then the 16 bytes of lane group g, register half h of row 5 are
// zeroed: the row drops by 32 if those 16 elements belong to block b, by 16 otherwise.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mx_scale_probe2.hip -o tools/micro/_bin/mx_scale_probe2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned* sa, int zero_lane, int zero_half, float* out) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (lane == zero_lane && (i >> 2) == zero_half) ? 0 : 0x38383838;
    b[i] = 0x38383838;
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)sa[lane], 0, 0x7f7f7f7f);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

int main() {
  unsigned hsa[64], *dsa;
  float *dout, hout[256];
  hipMalloc(&dsa, 256);
  hipMalloc(&dout, 1024);
  for (int b = 0; b < 4; ++b) {
    for (int l = 0; l < 64; ++l) hsa[l] = l == 5 + 16 * b ? 0x80808080u : 0x7f7f7f7fu;
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice);
    printf("scale x2 in lane %2d (row 5, block %d): row 5 after zeroing (lane group g, half h):", 5 + 16 * b, b);
    for (int g = 0; g < 4; ++g)
      for (int h = 0; h < 2; ++h) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dsa, 5 + 16 * g, h, dout);
        hipMemcpy(hout, dout, 1024, hipMemcpyDeviceToHost);
        // row 5, column 0: lane (5 >> 2) * 16 + 0, register 5 & 3
        printf("  g%d h%d: %3.0f", g, h, hout[((5 >> 2) * 16 + 0) * 4 + (5 & 3)]);
      }
    printf("\n");
  }
  return 0;
}
