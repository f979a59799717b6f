// MX block scales of the e4m3 path (BASELINE config 5): every 32 consecutive K-elements of an activation row share one
// e8m0 exponent byte (value 2^(byte - 127)) that v_mfma_scale_f32_16x16x128_f8f6f4 applies in hardware -- lane l's scale
// byte covers exactly the 32 bytes of the operand row that lane holds (row l & 15, k block l >> 4; op_sel picks the byte
// of the 32-bit scale register; probed in tools/micro/mx_scale_probe.hip, profiles/r03_mx_scale_probe.txt).
//
// Layout of a scale tensor for an activation X[M, K] that a 256-row-tile GEMM consumes: indexed so that the 8 bytes one
// lane needs for one 128-wide k-tile -- its k block (kb & 3) of the rows j * 16 + (l & 15), j = 0..7, of a 128-row block
// -- are contiguous (one 8-byte load per lane and k-tile, straight into the MFMA's scale operands):
//   byte((m, kb)) = (((kb >> 2) * MB + (m >> 7)) * 64 + (kb & 3) * 16 + (m & 15)) * 8 + ((m >> 4) & 7),  MB = ceil(M / 128)
// Size: (K / 128) * MB * 512 bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__host__ __device__ inline size_t mx_index(int64_t m, int kb, int64_t MB) {
  return ((((size_t)(kb >> 2) * (size_t)MB + (size_t)(m >> 7)) * 64 + (size_t)((kb & 3) * 16 + (int)(m & 15))) * 8) +
         (size_t)((m >> 4) & 7);
}
__host__ __device__ inline int64_t mx_blocks128(int64_t M) { return (M + 127) >> 7; }
__host__ __device__ inline size_t mx_bytes(int64_t M, int64_t K) { return (size_t)(K / 128) * (size_t)mx_blocks128(M) * 512; }

// smallest power of two 2^e with amax * 2^-e <= 448 (the largest finite e4m3), as the e8m0 byte e + 127 in [1, 253];
// amax == 0 (or denormal dust) -> byte 1
__device__ __forceinline__ unsigned mx_e8m0(float amax) {
  const unsigned u = __float_as_uint(amax * (1.0f / 448.0f));
  unsigned e = (u + 0x7fffffu) >> 23;   // ceil(log2) + 127 (exact powers of two stay)
  e = e < 1u ? 1u : e;
  return e > 253u ? 253u : e;
}
// 2^-(byte - 127): the factor that brings the block into e4m3 range
__device__ __forceinline__ float mx_inv_scale(unsigned byte) { return __uint_as_float((254u - byte) << 23); }
