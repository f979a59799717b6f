// Element-type helpers shared by the fp16 / bf16 GEMM kernels (gemm_sk.hip; gemm_f16.hip keeps its own copy of the same
// definitions inside its anonymous namespace): storage <-> fp32 conversion, the 16x16x32 MFMA of the type, and nn.GELU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace codetr_gemm {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct HalfT {
  using frag = f16x8;
  __device__ static f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  __device__ static float to_f32(unsigned short bits) {
    _Float16 h;
    __builtin_memcpy(&h, &bits, 2);
    return (float)h;
  }
  __device__ static unsigned short from_f32(float v) {
    _Float16 h = (_Float16)v;
    unsigned short bits;
    __builtin_memcpy(&bits, &h, 2);
    return bits;
  }
  // two fp32 -> one dword of two halves (round to nearest even: v_cvt_pk_f16_f32 on gfx950)
  __device__ static unsigned pack2(float lo, float hi) {
    f16x2 h = {(_Float16)lo, (_Float16)hi};
    unsigned o;
    __builtin_memcpy(&o, &h, 4);
    return o;
  }
};

struct BFloatT {
  using frag = bf16x8;
  __device__ static f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  __device__ static float to_f32(unsigned short bits) { return __uint_as_float(((unsigned)bits) << 16); }
  // (the hardware converter: v_cvt_pk_bf16_f32, round to nearest even, quiet NaN)
  __device__ static unsigned short from_f32(float v) {
    const __bf16 h = (__bf16)v;
    return __builtin_bit_cast(unsigned short, h);
  }
  __device__ static unsigned pack2(float lo, float hi) {
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    const bf16x2v h = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, h);
  }
};

// nn.GELU (erf form) = 0.5 x (1 + erf(x / sqrt 2)), erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): the sign of erf
// folds into |x| and the halves into the coefficients -- 11 plain operations + v_rcp + v_exp.  Same code as gemm_f16.hip.
__device__ __forceinline__ float gelu_erf(float x) {
  const float u = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, u, 1.0f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  const float ez = __builtin_amdgcn_exp2f(u * u * (-0.5f * 1.4426950408889634f));
  return fmaf(u, fmaf(-(p * t), ez, 0.5f), 0.5f * x);
}

// the same function on two values at once: the 11 plain operations as 6 packed ones (v_pk_fma_f32 / v_pk_mul_f32 are full
// rate on gfx950 when no MFMA competes for the issue slot -- an epilogue), v_rcp / v_exp per element.  Same operations in
// the same order as gelu_erf, so the results are bit-identical.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 u = {fabsf(x.x), fabsf(x.y)};
  const f32x2 d = __builtin_elementwise_fma(f32x2{0.3275911f * 0.70710678118654752f, 0.3275911f * 0.70710678118654752f}, u,
                                            f32x2{1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2 p = __builtin_elementwise_fma(f32x2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, t,
                                      f32x2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
  const f32x2 e = (u * u) * f32x2{-0.5f * 1.4426950408889634f, -0.5f * 1.4426950408889634f};
  const f32x2 ez = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
  const f32x2 h = __builtin_elementwise_fma(-(p * t), ez, f32x2{0.5f, 0.5f});
  return __builtin_elementwise_fma(u, h, x * f32x2{0.5f, 0.5f});
}

}  // namespace codetr_gemm
