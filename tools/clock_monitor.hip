// Shader-clock monitor: one wave samples s_memtime (shader clock ticks) against s_memrealtime (100 MHz) while a
// workload runs on other streams.  Built as a tiny shared library and driven from tools/clock_trace.py.
//   hipcc -O3 -shared -fPIC --offload-arch=gfx950 tools/clock_monitor.hip -o tools/libclockmon.so
#include <hip/hip_runtime.h>

__global__ void clock_monitor_kernel(long long* buf, int nsamples, int interval_ticks, volatile int* stop) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < nsamples; ++i) {
    const long long c0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    while ((long long)wall_clock64() - w0 < interval_ticks) __builtin_amdgcn_s_sleep(32);
    const long long c1 = __builtin_readcyclecounter();
    const long long w1 = wall_clock64();
    buf[3 * i] = w0;
    buf[3 * i + 1] = w1 - w0;
    buf[3 * i + 2] = c1 - c0;
    if (*stop) { for (int j = i + 1; j < nsamples; ++j) buf[3 * j + 1] = 0; break; }
  }
}

// marks wall-clock time on the workload's stream (begin / end of the region of interest)
__global__ void clock_mark_kernel(long long* slot) {
  if (threadIdx.x == 0) *slot = wall_clock64();
}

extern "C" int clockmon_launch(void* stream, long long* buf, int nsamples, int interval_ticks, int* stop) {
  static hipStream_t own = nullptr;
  if (!stream) {
    if (!own && hipStreamCreateWithFlags(&own, hipStreamNonBlocking) != hipSuccess) return -1;
    stream = own;
  }
  hipLaunchKernelGGL(clock_monitor_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, buf, nsamples, interval_ticks, stop);
  return (int)hipGetLastError();
}
extern "C" int clockmon_mark(void* stream, long long* slot) {
  hipLaunchKernelGGL(clock_mark_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slot);
  return (int)hipGetLastError();
}
