// How many wait states does gfx950 need between v_cvt_pk_bf16_f32 writing a VGPR and an MFMA reading it as an operand?
// One asm statement per case (nothing the compiler can pad): B operand dword 0 is poisoned, then written by the conversion, then N states
// later the MFMA reads it.  A = ones, so D[i][j] = sum_k B[k][j]; a stale read shows up as the poison value in the sums.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/cvt_mfma_hazard.hip -o tools/ubench/cvt_mfma_hazard.bin && tools/ubench/cvt_mfma_hazard.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int N>
__global__ void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float bad = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float x = 1.0f + (float)((lane + it) & 7), y = 2.0f;
    u32x4 a = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};   // bf16 ones
    u32x4 b = {0x43004300u, 0u, 0u, 0u};                              // dword 0 poisoned with (128, 128)
    f32x4 d;
    if (N == 0) asm volatile("v_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v100, %4\n\ts_nop 7\n\tv_cvt_pk_bf16_f32 v100, %2, %3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, v[100:103], 0\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(x), "v"(y), "v"(b[0]) : "v100", "v101", "v102", "v103");
    if (N == 1) asm volatile("v_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v100, %4\n\ts_nop 7\n\tv_cvt_pk_bf16_f32 v100, %2, %3\n\ts_nop 0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, v[100:103], 0\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(x), "v"(y), "v"(b[0]) : "v100", "v101", "v102", "v103");
    if (N == 2) asm volatile("v_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v100, %4\n\ts_nop 7\n\tv_cvt_pk_bf16_f32 v100, %2, %3\n\ts_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, v[100:103], 0\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(x), "v"(y), "v"(b[0]) : "v100", "v101", "v102", "v103");
    if (N == 3) asm volatile("v_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v100, %4\n\ts_nop 7\n\tv_cvt_pk_bf16_f32 v100, %2, %3\n\ts_nop 2\n\tv_mfma_f32_16x16x32_bf16 %0, %1, v[100:103], 0\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(x), "v"(y), "v"(b[0]) : "v100", "v101", "v102", "v103");
    if (N == 4) asm volatile("v_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v100, %4\n\ts_nop 7\n\tv_cvt_pk_bf16_f32 v100, %2, %3\n\ts_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, v[100:103], 0\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(x), "v"(y), "v"(b[0]) : "v100", "v101", "v102", "v103");
    if (N == 6) asm volatile("v_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v100, %4\n\ts_nop 7\n\tv_cvt_pk_bf16_f32 v100, %2, %3\n\ts_nop 5\n\tv_mfma_f32_16x16x32_bf16 %0, %1, v[100:103], 0\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(x), "v"(y), "v"(b[0]) : "v100", "v101", "v102", "v103");
    // expected: column j = lane & 15 sums B[k][j] over k; only k slots 0,1 of lane group l4 = lane >> 4 are non-zero: x(lane') + 2 for the 4 lanes of the column
    float want = 0.f;
    for (int g = 0; g < 4; ++g) want += 1.0f + (float)(((lane & 15) + 16 * g + it) & 7) + 2.0f;
    for (int r = 0; r < 4; ++r) bad += (d[r] != want) ? 1.f : 0.f;
    acc += d;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = bad;
}

template <int N> void run(float* dout, std::vector<float>& h) {
  hipLaunchKernelGGL(k<N>, dim3(1024), dim3(512), 0, 0, dout, 2000);
  hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
  double s = 0; for (float v : h) s += v;
  printf("wait states %d: %.0f wrong results of %.0f\n", N, s, (double)h.size() * 2000 * 4);
}
int main() {
  float* dout; std::vector<float> h(1024 * 512);
  hipMalloc(&dout, h.size() * 4);
  run<0>(dout, h); run<1>(dout, h); run<2>(dout, h); run<3>(dout, h); run<4>(dout, h); run<6>(dout, h);
  return 0;
}
