// VALU-rate microbenchmark for the GELU epilogue variants (8 waves per CU-resident workgroup, values in registers).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gelu_rate.hip -o gpurun_out/gelu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float gelu_as(float x) {   // current: A&S 7.1.25 (rcp + exp2)
  const float ax = fabsf(x);
  const float z = ax * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.47047f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, 0.7478556f, -0.0958798f), 0.3480242f);
  const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
  const float erfz = fmaf(-poly, e, 1.0f);
  return 0.5f * fmaf(ax, erfz, x);
}
#define P0 1.12838531f
#define P1 0.153424003f
#define P2 0.0432474986f
#define P3 0.000753648848f
#define Q1 0.469360935f
#define Q2 0.0945981576f
#define Q3 0.00932609519f
__device__ __forceinline__ float gelu_rat(float x) {  // erf(z) = z P(z^2) / Q(z^2), z clamped to +-3.3 (one rcp)
  const float z = __builtin_amdgcn_fmed3f(x * 0.70710678118654752440f, -3.3f, 3.3f);
  const float t = z * z;
  const float p = fmaf(t, fmaf(t, fmaf(t, P3, P2), P1), P0);
  const float q = fmaf(t, fmaf(t, fmaf(t, Q3, Q2), Q1), 1.0f);
  const float e = z * p * __builtin_amdgcn_rcpf(q);
  const float hx = 0.5f * x;
  return fmaf(hx, e, hx);
}
__device__ __forceinline__ f2 gelu_rat2(f2 x) {       // the same on two elements with packed fp32 math
  f2 z = x * 0.70710678118654752440f;
  z.x = __builtin_amdgcn_fmed3f(z.x, -3.3f, 3.3f); z.y = __builtin_amdgcn_fmed3f(z.y, -3.3f, 3.3f);
  const f2 t = z * z;
  const f2 p = __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, (f2)P3, (f2)P2), (f2)P1), (f2)P0);
  const f2 q = __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, (f2)Q3, (f2)Q2), (f2)Q1), (f2)1.0f);
  f2 r; r.x = __builtin_amdgcn_rcpf(q.x); r.y = __builtin_amdgcn_rcpf(q.y);
  const f2 e = z * p * r;
  const f2 hx = x * 0.5f;
  return __builtin_elementwise_fma(hx, e, hx);
}

template <int V>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters, long long* cyc) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x * 16 + i];
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      if (V == 0) { v[i] = gelu_as(v[i]) + 0.3f; v[i + 1] = gelu_as(v[i + 1]) + 0.3f; }
      if (V == 1) { v[i] = gelu_rat(v[i]) + 0.3f; v[i + 1] = gelu_rat(v[i + 1]) + 0.3f; }
      if (V == 2) { f2 a; a.x = v[i]; a.y = v[i + 1]; a = gelu_rat2(a) + 0.3f; v[i] = a.x; v[i + 1] = a.y; }
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[V] = t1 - t0;
}
__global__ void acc(const float* in, float* o) {   // accuracy: o[0..2] = max abs error of the three variants against erff
  __shared__ float m[3];
  if (threadIdx.x < 3) m[threadIdx.x] = 0.f;
  __syncthreads();
  float e0 = 0, e1 = 0, e2 = 0;
  for (int i = threadIdx.x; i < 200001; i += blockDim.x) {
    const float x = -10.f + i * 1e-4f;
    const float ref = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    f2 a; a.x = x; a.y = x; a = gelu_rat2(a);
    e0 = fmaxf(e0, fabsf(gelu_as(x) - ref)); e1 = fmaxf(e1, fabsf(gelu_rat(x) - ref)); e2 = fmaxf(e2, fabsf(a.y - ref));
  }
  atomicMax((int*)&m[0], __float_as_int(e0)); atomicMax((int*)&m[1], __float_as_int(e1)); atomicMax((int*)&m[2], __float_as_int(e2));
  __syncthreads();
  if (threadIdx.x < 3) o[threadIdx.x] = m[threadIdx.x];
}
int main() {
  float *in, *out, *eo; long long* cyc;
  hipMalloc(&in, 512 * 16 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64); hipMalloc(&eo, 16);
  std::vector<float> h(512 * 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = -3.f + 6.f * (float)(i % 977) / 977.f;
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    k<0><<<256, 512>>>(out, in, iters, cyc); k<1><<<256, 512>>>(out, in, iters, cyc); k<2><<<256, 512>>>(out, in, iters, cyc);
  }
  acc<<<1, 1024>>>(in, eo);
  hipDeviceSynchronize();
  long long c[3]; float e[3];
  hipMemcpy(c, cyc, 24, hipMemcpyDeviceToHost); hipMemcpy(e, eo, 12, hipMemcpyDeviceToHost);
  const char* nm[3] = {"A&S rcp+exp2", "rational scalar", "rational packed"};
  // per wave-instruction-stream: 8 waves on 4 SIMDs -> 2 waves/SIMD; cycles per element per wave
  for (int i = 0; i < 3; ++i) printf("%-16s %.2f cycles per gelu (wave-level, 2 waves/SIMD) max|err| %.2e\n", nm[i], (double)c[i] / (iters * 16.0 * 2), e[i]);
  return 0;
}
