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

// counter-based RNG for dropout / drop-path / state-dropout: one 32-bit hash per element.
__device__ __forceinline__ uint32_t hash32(uint32_t seed, uint32_t stream, uint32_t idx) {
  uint64_t z = ((uint64_t)(seed ^ (stream * 0x9E3779B9u)) << 32) | idx;
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= (z >> 31);
  return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t stream, uint32_t idx) {
  return (float)(hash32(seed, stream, idx) >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

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
