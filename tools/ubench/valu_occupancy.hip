// How fast does a SIMD issue VALU / mixed MFMA + VALU work at 1, 2, 3 and 4 waves per SIMD?  (round 6)
// The wave-private kernels (NAT levels, scene encoder, decoder) run 2 waves per SIMD at 210 - 250 VGPRs and are VALU-issue bound; the
// round-2 micro-benchmark (mfma_valu_overlap.hip) stopped at two waves: one wave issues a full-rate VALU instruction every ~4.7 cycles, two
// waves together one every ~2.8 -- the SIMD-32 pipe's own rate is one per 2 cycles.  This one measures what 3 (<= 168 VGPRs) and
// 4 (<= 128 VGPRs) waves per SIMD would buy, per kind of instruction stream:
//   plain  : independent v_fma_f32
//   pk16   : v_pk_fma_f16 (the packed GELU's instruction)
//   mixed  : per 1 MFMA (16x16x32 bf16) KV plain VALU instructions, KV = 5, 8, 12, 20 (the kernels' measured ratios)
// Prints cycles per VALU instruction PER SIMD (total VALU instructions of the SIMD's waves / elapsed s_memtime ticks).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_occupancy.hip -o tools/ubench/valu_occupancy.bin && tools/ubench/valu_occupancy.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int KV, bool DO_M, int KIND>      // KIND 0: v_fma_f32, 1: v_pk_fma_f16, 2: v_exp_f32 every 8th, 3: v_pk_mul_f32
__device__ __forceinline__ void body(int iters, int lane, float* sink_slot) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3F80 + lane); b[i] = (short)(0x3F80 + i); }
  f32x4 c[4];
  for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane + i;
  const float m = 0.999f;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (DO_M) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < KV; ++j) {
        const int r = (i * KV + j) & 7;
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(m));
        if (KIND == 1) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(v[r]) : "v"(m));
        if (KIND == 2) { if ((j & 7) == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r])); else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(m)); }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  *sink_slot = s;
}

template <int KV, bool DO_M, int KIND, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(long long* out, float* sink, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  const long long t0 = clock64();
  body<KV, DO_M, KIND>(iters, lane, sink + blockIdx.x * blockDim.x + threadIdx.x);
  const long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int KV, bool DO_M, int KIND, int WAVES>
double run1(long long* dout, float* sink) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<KV, DO_M, KIND, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, dout, sink, iters);
  std::vector<long long> h(256 * 16);
  (void)hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < WAVES; ++w) mx = mx > (double)h[b * 16 + w] ? mx : (double)h[b * 16 + w];
  // VALU instructions of one SIMD = (WAVES / 4) waves x iters x 4 x KV
  return mx / ((double)(WAVES / 4) * iters * 4 * KV);
}

template <int KV, bool DO_M, int KIND>
void row(const char* label, long long* dout, float* sink) {
  const double a = run1<KV, DO_M, KIND, 4>(dout, sink), b = run1<KV, DO_M, KIND, 8>(dout, sink), c = run1<KV, DO_M, KIND, 12>(dout, sink), d = run1<KV, DO_M, KIND, 16>(dout, sink);
  printf("%-44s %6.2f %6.2f %6.2f %6.2f   (2 -> 3 waves: x%.2f, 2 -> 4: x%.2f)\n", label, a, b, c, d, b / c, b / d);
}

int main() {
  long long* dout; float* sink;
  (void)hipMalloc(&dout, 256 * 16 * 8); (void)hipMalloc(&sink, 256 * 1024 * 4);
  printf("s_memtime ticks per VALU instruction of a SIMD, at 1 / 2 / 3 / 4 waves per SIMD (256 workgroups, one per CU)\n");
  printf("%-44s %6s %6s %6s %6s\n", "stream", "1", "2", "3", "4");
  row<8, false, 0>("v_fma_f32 only", dout, sink);
  row<8, false, 1>("v_pk_fma_f16 only", dout, sink);
  row<8, false, 2>("7 v_fma_f32 + 1 v_exp_f32", dout, sink);
  row<5, true, 0>("1 MFMA : 5 v_fma_f32  (decoder, pass B)", dout, sink);
  row<8, true, 0>("1 MFMA : 8 v_fma_f32  (level 2, encoder)", dout, sink);
  row<12, true, 0>("1 MFMA : 12 v_fma_f32 (level 1)", dout, sink);
  row<20, true, 0>("1 MFMA : 20 v_fma_f32 (level 0)", dout, sink);
  row<12, true, 1>("1 MFMA : 12 v_pk_fma_f16", dout, sink);
  return 0;
}
