// What does a hipGraph replay of a small-batch update step cost on this stack, against issuing the same launches one by one?
// The step of tools/host_time.py at 32 scenes is ~50 launches on three streams joined by events, each kernel a few microseconds long: the
// host's issue time IS the step.  This builds that shape synthetically -- NK kernels of SPIN ticks each, a main chain
// with two side chains forked and joined by events -- and prints, per step: host issue time and wall time, for direct launches and for
// hipGraphLaunch of the stream-captured step (one graph, relaunched), with the kernel arguments unchanged between replays and with ONE
// by-value argument per node rewritten before every replay (hipGraphExecKernelNodeSetParams: what changing seeds would cost).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/graph_launch.hip -o tools/ubench/graph_launch.bin && tools/ubench/graph_launch.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void spin_kernel(float* out, int spin, unsigned seed) {
  const long long t0 = clock64();
  float a = (float)seed;
  while (clock64() - t0 < spin) a = a * 1.0001f + 1.f;
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int NK = argc > 1 ? atoi(argv[1]) : 50, SPIN = argc > 2 ? atoi(argv[2]) : 200, STEPS = 300, BLOCKS = 32;
  hipStream_t s0, s1, s2;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t fork, j1, j2;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j2, hipEventDisableTiming));
  float* out; CK(hipMalloc(&out, 4096));
  // one step: 6 kernels on s0, fork; NK/3 on each of s1 / s2 beside the rest of s0; join; 4 more on s0
  auto issue = [&](unsigned seed) -> int {
    int n = 0;
    for (int i = 0; i < 6; ++i, ++n) hipLaunchKernelGGL(spin_kernel, dim3(BLOCKS), dim3(256), 0, s0, out, SPIN, seed + n);
    CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); CK(hipStreamWaitEvent(s2, fork, 0));
    const int side = NK / 3;
    for (int i = 0; i < side; ++i, ++n) hipLaunchKernelGGL(spin_kernel, dim3(BLOCKS), dim3(256), 0, s1, out + 64, SPIN, seed + n);
    for (int i = 0; i < side; ++i, ++n) hipLaunchKernelGGL(spin_kernel, dim3(BLOCKS), dim3(256), 0, s2, out + 128, SPIN, seed + n);
    for (int i = 0; i < NK - 10 - 2 * side; ++i, ++n) hipLaunchKernelGGL(spin_kernel, dim3(BLOCKS), dim3(256), 0, s0, out, SPIN, seed + n);
    CK(hipEventRecord(j1, s1)); CK(hipEventRecord(j2, s2)); CK(hipStreamWaitEvent(s0, j1, 0)); CK(hipStreamWaitEvent(s0, j2, 0));
    for (int i = 0; i < 4; ++i, ++n) hipLaunchKernelGGL(spin_kernel, dim3(BLOCKS), dim3(256), 0, s0, out, SPIN, seed + n);
    return 0;
  };
  // ---- direct
  for (int w = 0; w < 20; ++w) if (issue(w)) return 1;
  CK(hipDeviceSynchronize());
  double t0 = now_us(), host = 0;
  for (int k = 0; k < STEPS; ++k) { const double a = now_us(); if (issue(k)) return 1; host += now_us() - a; }
  CK(hipDeviceSynchronize());
  double wall = now_us() - t0;
  printf("direct   : %d launches/step, host issue %.1f us/step, wall %.1f us/step\n", NK, host / STEPS, wall / STEPS);
  // ---- captured
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
  if (issue(7)) return 1;
  CK(hipStreamEndCapture(s0, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
  for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, s0));
  CK(hipStreamSynchronize(s0));
  t0 = now_us(); host = 0;
  for (int k = 0; k < STEPS; ++k) { const double a = now_us(); CK(hipGraphLaunch(ge, s0)); host += now_us() - a; }
  CK(hipStreamSynchronize(s0));
  wall = now_us() - t0;
  printf("graph    : %zu nodes, host issue %.1f us/step, wall %.1f us/step\n", nn, host / STEPS, wall / STEPS);
  // ---- captured, one by-value argument of every kernel node rewritten per replay
  std::vector<hipGraphNode_t> kn; std::vector<hipKernelNodeParams> kp;
  for (auto n : nodes) { hipGraphNodeType t; CK(hipGraphNodeGetType(n, &t)); if (t == hipGraphNodeTypeKernel) { hipKernelNodeParams p; CK(hipGraphKernelNodeGetParams(n, &p)); kn.push_back(n); kp.push_back(p); } }
  std::vector<float*> a0(kn.size()); std::vector<int> a1(kn.size(), SPIN); std::vector<unsigned> a2(kn.size());
  std::vector<void*> argv3(kn.size() * 3);
  for (size_t i = 0; i < kn.size(); ++i) { a0[i] = *reinterpret_cast<float**>(kp[i].kernelParams[0]); argv3[3 * i] = &a0[i]; argv3[3 * i + 1] = &a1[i]; argv3[3 * i + 2] = &a2[i]; kp[i].kernelParams = &argv3[3 * i]; }
  t0 = now_us(); host = 0;
  for (int k = 0; k < STEPS; ++k) {
    const double a = now_us();
    for (size_t i = 0; i < kn.size(); ++i) { a2[i] = k + (unsigned)i; CK(hipGraphExecKernelNodeSetParams(ge, kn[i], &kp[i])); }
    CK(hipGraphLaunch(ge, s0)); host += now_us() - a;
  }
  CK(hipStreamSynchronize(s0));
  wall = now_us() - t0;
  printf("graph+set: %zu kernel nodes rewritten per replay, host issue %.1f us/step, wall %.1f us/step\n", kn.size(), host / STEPS, wall / STEPS);
  // ---- single-stream chain of the same launches, for the device-side floor
  t0 = now_us();
  for (int k = 0; k < STEPS; ++k) for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(spin_kernel, dim3(BLOCKS), dim3(256), 0, s0, out, SPIN, k);
  CK(hipStreamSynchronize(s0));
  printf("one queue: wall %.1f us/step (%d launches)\n", (now_us() - t0) / STEPS, NK);
  return 0;
}
