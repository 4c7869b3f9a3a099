// Microbenchmark: per-CU rate at which 8 waves stream an L2-resident bf16 weight matrix into registers,
// (a) MFMA-fragment pattern over a row-major [N][K] image (16 rows x 64 B per wave instruction),
// (b) fragment-major image (1 KiB contiguous per wave instruction).  Prints bytes / cycle / CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;

template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void stream_kernel(const unsigned short* W, int N, int K, int reps, int depth, long long* cyc, float* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int ntiles = N / 16, ksteps = K / 32;
  float acc = 0.f;
  __syncthreads();
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    for (int nt = wave; nt < ntiles; nt += NW) {
      for (int ks0 = 0; ks0 < ksteps; ks0 += 8) {
        bf16x8 f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ks = ks0 + u;
          if (MODE == 0) f[u] = *reinterpret_cast<const bf16x8*>(W + (size_t)(nt * 16 + l15) * K + ks * 32 + l4 * 8);
          else f[u] = *reinterpret_cast<const bf16x8*>(W + ((size_t)(nt * ksteps + ks) * 64 + lane) * 8);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (float)f[u][0] + (float)f[u][7];
      }
    }
  }
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  const int N = 256, K = 256;
  unsigned short* W; long long* cyc; float* sink;
  hipMalloc(&W, (size_t)N * K * 2); hipMemset(W, 0, (size_t)N * K * 2);
  hipMalloc(&cyc, 4096 * 8); hipMalloc(&sink, 4);
  const int reps = 20;
  for (int grid : {1, 256, 1024}) {
    for (int mode = 0; mode < 2; ++mode) {
      for (int nw : {4, 8, 16}) {
        for (int it = 0; it < 2; ++it) {
          if (mode == 0) { if (nw == 4) stream_kernel<0, 4><<<grid, 256>>>(W, N, K, reps, 0, cyc, sink); else if (nw == 8) stream_kernel<0, 8><<<grid, 512>>>(W, N, K, reps, 0, cyc, sink); else stream_kernel<0, 16><<<grid, 1024>>>(W, N, K, reps, 0, cyc, sink); }
          else { if (nw == 4) stream_kernel<1, 4><<<grid, 256>>>(W, N, K, reps, 0, cyc, sink); else if (nw == 8) stream_kernel<1, 8><<<grid, 512>>>(W, N, K, reps, 0, cyc, sink); else stream_kernel<1, 16><<<grid, 1024>>>(W, N, K, reps, 0, cyc, sink); }
          hipDeviceSynchronize();
        }
        std::vector<long long> h(grid);
        hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= grid;
        printf("grid %4d mode %d waves %2d: %.0f cycles per WG for %d x 128 KiB -> %.1f B/cycle/WG\n", grid, mode, nw, avg, reps, (double)reps * N * K * 2 / avg);
      }
    }
  }
  return 0;
}
