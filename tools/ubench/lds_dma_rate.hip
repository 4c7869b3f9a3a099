// How fast does a CU fill a 32 KiB LDS ring slot from an L2-resident image by LDS-DMA (global_load_lds_dwordx4, 4 x 1 KiB per wave and group,
// s_waitcnt vmcnt + s_barrier per group), with every CU of the chip doing the same (the weight stream of the wave-private kernels)?
// AHEAD = 1: a two-slot ring, the next group requested at the barrier that opens the current one; AHEAD = 2: three slots, two groups in flight.
// Prints ticks (s_memtime) per group and bytes / tick / CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_rate.hip -o tools/ubench/lds_dma_rate.bin && tools/ubench/lds_dma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <stdint.h>

__device__ __forceinline__ void glds(const void* src, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  const uint64_t a = reinterpret_cast<uint64_t>(src);
  const uint64_t sa = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t sd = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sa), "s"(sd) : "memory");
}

// four consecutive fragments behind ONE M0 write: the instruction offset moves the global AND the LDS address
__device__ __forceinline__ void glds4(const void* src, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  const uint64_t a = reinterpret_cast<uint64_t>(src);
  const uint64_t sa = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t sd = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sa), "s"(sd) : "memory");
}

template <int AHEAD, bool ONE_M0>
__global__ __launch_bounds__(512) void k(const unsigned char* img, int ngroups, int total_groups, long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
  constexpr int NS = AHEAD + 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ring);
  float acc = 0.f;
  auto req = [&](int g) {
    const unsigned char* src = img + (size_t)(g % ngroups) * 32768;
    if (ONE_M0) glds4(src + (size_t)wv * 4096, lane * 16u, lds0 + (uint32_t)(g % NS) * 32768u + wv * 4096u);
    else for (int f = wv; f < 32; f += 8) glds(src + (size_t)f * 1024, lane * 16u, lds0 + (uint32_t)(g % NS) * 32768u + f * 1024u);
  };
  __syncthreads();
  const long long t0 = clock64();
  for (int g = 0; g < AHEAD; ++g) req(g);
  for (int g = 0; g < total_groups; ++g) {
    // group g has landed when at most 4 (AHEAD - 1) of my requests are outstanding (in-order returns)
    if (AHEAD == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    req(g + AHEAD);
    acc += *reinterpret_cast<const float*>(ring + (g % NS) * 32768 + tid * 4);
  }
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  // check: the last group's slot holds that group's bytes (image byte = its group index)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  const int gl = total_groups - 1;
  int bad = 0;
  for (int i = tid; i < 32768; i += 512) bad += ring[(gl % NS) * 32768 + i] != (unsigned char)(gl % ngroups + 1);
  if (bad) atomicAdd(reinterpret_cast<int*>(sink) + 1, bad);
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  const int ngroups = 88, total = 880;           // a 2.75 MiB image (fits every XCD's L2), walked ten times; byte value = group index + 1
  unsigned char* img; long long* cyc; float* sink;
  (void)hipMalloc(&img, (size_t)ngroups * 32768);
  for (int g = 0; g < ngroups; ++g) (void)hipMemset(img + (size_t)g * 32768, g + 1, 32768);
  (void)hipMalloc(&cyc, 256 * 8); (void)hipMalloc(&sink, 8); (void)hipMemset(sink, 0, 8);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  for (int grid : {1, 256}) {
    for (int v = 0; v < 4; ++v) {
      const int ahead = 1 + (v & 1); const bool one = v >= 2;
      for (int it = 0; it < 2; ++it) {
        if (v == 0) hipLaunchKernelGGL((k<1, false>), dim3(grid), dim3(512), 65536, 0, img, ngroups, total, cyc, sink);
        if (v == 1) hipLaunchKernelGGL((k<2, false>), dim3(grid), dim3(512), 98304, 0, img, ngroups, total, cyc, sink);
        if (v == 2) hipLaunchKernelGGL((k<1, true>), dim3(grid), dim3(512), 65536, 0, img, ngroups, total, cyc, sink);
        if (v == 3) hipLaunchKernelGGL((k<2, true>), dim3(grid), dim3(512), 98304, 0, img, ngroups, total, cyc, sink);
        (void)hipDeviceSynchronize();
      }
      std::vector<long long> h(grid);
      (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
      int chk[2]; (void)hipMemcpy(chk, sink, 8, hipMemcpyDeviceToHost);
      double mx = 0; for (auto x : h) mx = mx > (double)x ? mx : (double)x;
      printf("grid %3d, %d group(s) in flight, %s: %6.0f ticks per 32 KiB group -> %5.1f B/tick/CU   (wrong bytes so far: %d)\n", grid, ahead,
             one ? "one M0 write per 4 fragments" : "M0 written per fragment   ", mx / total, 32768.0 * total / mx, chk[1]);
    }
  }
  return 0;
}
