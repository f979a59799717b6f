// Probe of v_mfma_scale_f32_16x16x128_f8f6f4's block-scale operands (no ISA manual in this image): which lane's scale
// byte applies to which (row, 32-wide k block) of A and B, what the e8m0 byte means, what op_sel selects.
// One wave: A[16 x 128] = B[16 x 128] = 1.0 (e4m3 0x38) -> every D element is 128 at unit scales.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mx_scale_probe.hip -o tools/micro/_bin/mx_scale_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: scales given as literal zeros (the "unscaled" form the product kernels use)
// mode 1: scale_a = sa[lane], scale_b = sb[lane], op_sel 0
// mode 2: same with op_sel_a = opa, op_sel_b = opb (byte select?)
template <int OPA, int OPB>
__global__ void probe(const unsigned* sa, const unsigned* sb, const unsigned char* afill, float* out, int mode) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  const unsigned av = afill[lane] * 0x01010101u;
  for (int i = 0; i < 8; ++i) {
    a[i] = (int)av;
    b[i] = 0x38383838;
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (mode == 0)
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
  else
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OPA, (int)sa[lane], OPB, (int)sb[lane]);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

static void show(const char* what, const float* h) {
  printf("%s\n", what);
  // D layout of the 16x16 f32 result: lane l holds rows 4*(l>>4)+r, column l&15 (as for the other 16x16 MFMAs)
  for (int row = 0; row < 16; ++row) {
    printf("  row %2d:", row);
    for (int col = 0; col < 16; ++col) printf(" %6.0f", h[(((row >> 2) << 4) + col) * 4 + (row & 3)]);
    printf("\n");
  }
}

int main() {
  unsigned hsa[64], hsb[64];
  unsigned char haf[64];
  unsigned *dsa, *dsb;
  unsigned char* daf;
  float *dout, hout[256];
  hipMalloc(&dsa, 256);
  hipMalloc(&dsb, 256);
  hipMalloc(&daf, 64);
  hipMalloc(&dout, 1024);
  auto run = [&](int mode, int opa, int opb, const char* what) {
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice);
    hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    hipMemcpy(daf, haf, 64, hipMemcpyHostToDevice);
    if (opa == 0 && opb == 0) hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, dsa, dsb, daf, dout, mode);
    else if (opa == 1 && opb == 0) hipLaunchKernelGGL((probe<1, 0>), dim3(1), dim3(64), 0, 0, dsa, dsb, daf, dout, mode);
    else if (opa == 2 && opb == 0) hipLaunchKernelGGL((probe<2, 0>), dim3(1), dim3(64), 0, 0, dsa, dsb, daf, dout, mode);
    else hipLaunchKernelGGL((probe<3, 0>), dim3(1), dim3(64), 0, 0, dsa, dsb, daf, dout, mode);
    hipMemcpy(hout, dout, 1024, hipMemcpyDeviceToHost);
    show(what, hout);
  };
  for (int l = 0; l < 64; ++l) haf[l] = 0x38;
  for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 0;
  run(0, 0, 0, "literal zero scales (expect 128 everywhere)");
  for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 0x7f7f7f7f;
  run(1, 0, 0, "scale bytes 0x7f (e8m0 1.0?) both");
  for (int l = 0; l < 64; ++l) hsa[l] = 0x7f7f7f80;   // low byte 0x80 (x2?)
  run(1, 0, 0, "scale_a low byte 0x80, op_sel 0 (expect 256 if byte 0 is used and 0x80 = 2.0)");
  run(1, 1, 0, "scale_a low byte 0x80, op_sel_a 1");
  for (int l = 0; l < 64; ++l) hsa[l] = 0x7f7f807f;   // byte 1 = 0x80
  run(1, 1, 0, "scale_a byte 1 = 0x80, op_sel_a 1");
  run(1, 2, 0, "scale_a byte 1 = 0x80, op_sel_a 2");
  // which lane's scale applies to which row / k block: only lane t has x2
  for (int t : {0, 5, 16, 21, 37, 63}) {
    for (int l = 0; l < 64; ++l) hsa[l] = l == t ? 0x80808080u : 0x7f7f7f7fu;
    char buf[128];
    snprintf(buf, sizeof buf, "scale_a x2 in lane %d only (row = lane & 15, k block = lane >> 4 => that row +32)", t);
    run(1, 0, 0, buf);
  }
  for (int l = 0; l < 64; ++l) hsa[l] = 0x7f7f7f7f;
  for (int t : {3, 19}) {
    for (int l = 0; l < 64; ++l) hsb[l] = l == t ? 0x80808080u : 0x7f7f7f7fu;
    char buf[128];
    snprintf(buf, sizeof buf, "scale_b x2 in lane %d only (column = lane & 15)", t);
    run(1, 0, 0, buf);
  }
  // confirm the k block: zero A in the elements of lane 21 (row 5, k block 1) and scale lane 21 by 2: nothing changes
  for (int l = 0; l < 64; ++l) { hsb[l] = 0x7f7f7f7f; hsa[l] = l == 21 ? 0x80808080u : 0x7f7f7f7fu; haf[l] = l == 21 ? 0 : 0x38; }
  run(1, 0, 0, "A of lane 21 zeroed and its scale x2 (expect row 5 = 96)");
  return 0;
}
