// Can VALU instructions issue in the shadow of an MFMA?  Per iteration: M independent v_mfma_f32_16x16x32_bf16 (eight accumulators, no
// dependency stalls) interleaved with K independent full-rate VALU instructions (v_fma_f32 chains over 8 registers), one asm statement.
// Cases: one wave per SIMD (a wave overlapping its own MFMAs) and two waves per SIMD where wave A issues only MFMAs and wave B only VALU
// (overlap ACROSS waves).  Prints ticks per iteration (s_memtime) -- if the pipes overlap, the MFMA-only time hides the VALU time.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap.bin && tools/ubench/mfma_valu_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// one wave's loop: per iteration 8 independent MFMAs (DO_M) with KV independent full-rate VALU instructions after each (DO_V)
template <int KV, bool DO_M, bool DO_V>
__device__ __forceinline__ void body(int iters, int lane, float* sink_slot) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3F80 + lane); b[i] = (short)(0x3F80 + i); }
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane + i;
  const float m = 0.999f;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (DO_M) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
      if (DO_V) {
#pragma unroll
        for (int j = 0; j < KV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(i * KV + j) & 7]) : "v"(m));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3] + v[i];
  *sink_slot = s;
}

// SPLIT: waves 0-3 (the first wave of each SIMD) issue only the MFMAs, waves 4-7 only the VALU instructions
template <int KV, bool DO_M, bool DO_V, bool SPLIT>
__global__ __launch_bounds__(512) void k(long long* out, float* sink, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  const long long t0 = clock64();
  if (SPLIT) {
    if (wave < 4) body<KV, true, false>(iters, lane, sink + blockIdx.x * blockDim.x + threadIdx.x);
    else body<KV, false, true>(iters, lane, sink + blockIdx.x * blockDim.x + threadIdx.x);
  } else body<KV, DO_M, DO_V>(iters, lane, sink + blockIdx.x * blockDim.x + threadIdx.x);
  const long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KV, bool DO_M, bool DO_V, bool SPLIT>
void run(const char* label, int waves, long long* dout, float* sink) {
  const int iters = 4000;
  hipLaunchKernelGGL((k<KV, DO_M, DO_V, SPLIT>), dim3(256), dim3(64 * waves), 0, 0, dout, sink, iters);
  std::vector<long long> h(256 * 8);
  (void)hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) mx = mx > (double)h[b * 8 + w] ? mx : (double)h[b * 8 + w];
  printf("%-56s %d waves/CU : %7.1f ticks per iteration\n", label, waves, mx / iters);
}

int main() {
  long long* dout; float* sink;
  (void)hipMalloc(&dout, 256 * 8 * 8); (void)hipMalloc(&sink, 256 * 512 * 4);
  printf("clock64 ticks (s_memtime); 4 waves per CU = one wave per SIMD; an iteration = 8 MFMA and / or 8 KV VALU\n");
  run<0, true, false, false>("8 MFMA", 4, dout, sink);
  run<1, false, true, false>("8 VALU", 4, dout, sink);
  run<3, false, true, false>("24 VALU", 4, dout, sink);
  run<6, false, true, false>("48 VALU", 4, dout, sink);
  run<1, true, true, false>("same wave: 8 MFMA + 8 VALU", 4, dout, sink);
  run<2, true, true, false>("same wave: 8 MFMA + 16 VALU", 4, dout, sink);
  run<3, true, true, false>("same wave: 8 MFMA + 24 VALU", 4, dout, sink);
  run<4, true, true, false>("same wave: 8 MFMA + 32 VALU", 4, dout, sink);
  run<6, true, true, false>("same wave: 8 MFMA + 48 VALU", 4, dout, sink);
  run<0, true, false, false>("two waves per SIMD, each 8 MFMA", 8, dout, sink);
  run<3, false, true, false>("two waves per SIMD, each 24 VALU", 8, dout, sink);
  run<3, true, true, false>("two waves per SIMD, each 8 MFMA + 24 VALU", 8, dout, sink);
  run<3, true, true, true>("two waves per SIMD: A 8 MFMA | B 24 VALU", 8, dout, sink);
  run<6, true, true, true>("two waves per SIMD: A 8 MFMA | B 48 VALU", 8, dout, sink);
  return 0;
}
