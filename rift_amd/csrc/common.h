// Shared device helpers for the RIFT gfx950 kernels (wave64, MFMA fragments, RNG).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rift {

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs (MFMA 16x16x32 A/B operand)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // MFMA 16x16 accumulator

__device__ __forceinline__ unsigned short f2bf(float f) {   // round-to-nearest-even fp32 -> bf16
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

// counter-based RNG for dropout / drop-path / state-dropout: one 32-bit hash per element
// (two rounds of the lowbias32 integer finaliser over (seed, stream, element index); 32-bit multiplies only).
__device__ __forceinline__ uint32_t hash32(uint32_t seed, uint32_t stream, uint32_t idx) {
  uint32_t x = idx * 0x9E3779B1u + (seed ^ (stream * 0x85EBCA6Bu));
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  x += seed * 0xC2B2AE35u + stream;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t stream, uint32_t idx) {
  return (float)(hash32(seed, stream, idx) >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7, far below bf16 resolution):
// one exp + one rcp instead of libm erff's ~60 instructions.  bf16 paths only; fp32 mode keeps erff.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfz = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erfz, x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace rift
