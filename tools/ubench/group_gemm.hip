// What one group GEMM of the wave-private kernels costs (decw_gemm of wp_stream.h: 32 fragments from LDS under 32 MFMAs, eight reads in
// flight) against the number of waves of a workgroup that run it at once (per cell: wave 0 / the slowest working wave) -- separates the MFMA pipe (16 cycles per instruction and SIMD),
// the LDS read path and the per-wave latency chain.  Prints s_memtime ticks per GEMM of wave 0, for 1..8 working waves, with and without a
// workgroup barrier between GEMMs, and for the 16-fragment halves (wp_gemm_half.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRIFT_OP_F16=0 -Irift_amd/csrc tools/ubench/group_gemm.hip -o tools/ubench/group_gemm.bin && tools/ubench/group_gemm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "wp_stream.h"
#include "wp_gemm_half.h"
using namespace RIFT_NS;

template <int MODE>     // 0: full group GEMM, 1: n-half, 2: k-half
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(long long* out, float* sink, int reps, int nwork, int barrier) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 65536 / 4; i += 512) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u;
  __syncthreads();
  h16x8 x[4];
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 8; ++j) x[k][j] = (short)0x3c00;
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint32_t base = (uint32_t)(uintptr_t)smem + (uint32_t)lane * 16u;
  const long long t0 = clock64();
  if (MODE == 4) {          // the first eight reads of GEMM r + 1 issued ahead of the barrier behind GEMM r (a ring that has the next group complete one barrier early)
    h16x8 w[8];
    if (wv < nwork) decw_gemm_pre(base, w);
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
      if (wv < nwork) {
        decw_gemm_post(base + (uint32_t)(r & 1) * 32768u, w, x, c);
        decw_gemm_pre(base + (uint32_t)((r + 1) & 1) * 32768u, w);
      }
      if (barrier) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    if (MODE >= 6 && wv >= 4) {          // the second wave of every SIMD starts late: 6: s_sleep 1 (64 cycles), 7: s_sleep 3, 8: s_sleep 6
      if (MODE == 6) asm volatile("s_sleep 1");
      if (MODE == 7) asm volatile("s_sleep 3");
      if (MODE == 8) asm volatile("s_sleep 6");
    }
    if (wv < nwork) {
      const uint32_t a = base + (uint32_t)(r & 1) * 32768u;
      if (MODE == 0 || MODE >= 5) decw_gemm<false>(a, x, c);
      if (MODE == 1) { f32x4 (&c4)[4] = reinterpret_cast<f32x4 (&)[4]>(c); decw_gemm_nhalf(a, x, c4); }
      if (MODE == 2) { h16x8 (&x2)[2] = reinterpret_cast<h16x8 (&)[2]>(x); decw_gemm_khalf(a, x2, c); }
      if (MODE == 3) {        // the shared-FFN arrangement: waves 0-3 a full GEMM, waves 4-7 a k-half each
        if (wv < 4) decw_gemm<false>(a, x, c);
        else { h16x8 (&x2)[2] = reinterpret_cast<h16x8 (&)[2]>(x); decw_gemm_khalf(a + (uint32_t)((wv >> 1) & 1) * 16384u, x2, c); }
      }
    }
    if (barrier && (MODE != 5 || (r & 1))) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (MODE 5: a barrier per TWO group GEMMs)
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  sink[blockIdx.x * 512 + tid] = s;
  if (lane == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
}

int main() {
  long long* out; float* sink;
  hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 4);
  const int reps = 2000;
  auto run = [&](auto kern, const char* name) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int barrier = 0; barrier < 2; ++barrier) {
      printf("%s%s:", name, barrier ? " + barrier" : "          ");
      for (int nwork = 1; nwork <= 8; ++nwork) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536, 0, out, sink, reps, nwork, barrier);
        hipDeviceSynchronize();
        std::vector<long long> h(8);
        hipMemcpy(h.data(), out, 64, hipMemcpyDeviceToHost);
        long long mx = 0;
        for (int w = 0; w < nwork; ++w) mx = h[w] > mx ? h[w] : mx;      // (without a barrier the waves of a SIMD do not finish together: the older one is served first)
        printf("  %d: %4.0f / %4.0f", nwork, (double)h[0] / reps, (double)mx / reps);
      }
      printf("\n");
    }
  };
  run(k<0>, "full 32 fragments ");
  run(k<1>, "n-half 16 fragments");
  run(k<2>, "k-half 16 fragments");
  run(k<3>, "4 full + k-halves  ");
  run(k<4>, "full, head prefetched");
  run(k<5>, "full, barrier per 2  ");
  run(k<6>, "full, skew s_sleep 1 ");
  run(k<7>, "full, skew s_sleep 3 ");
  run(k<8>, "full, skew s_sleep 6 ");
  return 0;
}
