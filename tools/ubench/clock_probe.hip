// Effective shader clock from inside a kernel: s_memtime (shader-clock ticks) against s_memrealtime (the constant 100 MHz reference) over a
// ~20 us window of one wave.  The RATIO does not depend on how fast the probe's own instructions issue, so the wave may share its SIMD with
// whatever the step is running.  tools/ramp_trace.py launches one probe per update step on a stream of its own (round 6: is the slow first
// 20-step region after an idle period a clock ramp?).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/clock_probe.hip -o tools/ubench/libclock_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void clock_probe_kernel(unsigned long long* out, int window_ticks_100mhz) {
  if (threadIdx.x != 0) return;
  const unsigned long long r0 = wall_clock64();
  const unsigned long long c0 = clock64();
  unsigned long long r1 = r0;
  while (r1 - r0 < (unsigned long long)window_ticks_100mhz) { __builtin_amdgcn_s_sleep(8); r1 = wall_clock64(); }
  const unsigned long long c1 = clock64();
  r1 = wall_clock64();
  out[0] = r0; out[1] = r1 - r0; out[2] = c1 - c0;
}

extern "C" int clock_probe_launch(void* out3, int window_ticks_100mhz, void* stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out3, window_ticks_100mhz);
  return (int)hipGetLastError();
}
