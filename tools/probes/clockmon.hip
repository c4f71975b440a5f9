// clockmon.hip -- the monitor kernel of clock_probe.hip as a tiny shared library, so that Python tools
// (tools/probe_clock.py) can watch the shader clock while torch-launched graphs of the library's
// kernels replay on another stream.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probes/bin/libclockmon.so tools/probes/clockmon.hip
#include <hip/hip_runtime.h>

__global__ void clockmon_kernel(unsigned long long* out, int n, int sleeps) {
  for (int i = 0; i < n; ++i) {
    const unsigned long long c = __builtin_amdgcn_s_memtime();      // shader-clock counter
    const unsigned long long w = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    out[2 * i] = c;
    out[2 * i + 1] = w;
    for (int j = 0; j < sleeps; ++j) __builtin_amdgcn_s_sleep(127);
  }
}

extern "C" __attribute__((visibility("default"))) int clockmon_launch(void* out, int n, int sleeps, void* stream) {
  hipLaunchKernelGGL(clockmon_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)out, n, sleeps);
  return (int)hipGetLastError();
}
