// grid_barrier_probe.hip -- what a device-wide barrier between the stages of a persistent kernel costs on gfx950
// next to what a kernel boundary inside a hipGraph costs (round 6: would chaining the small-M GEMVs of a decoder
// layer in ONE launch pay?).
//   (a) G workgroups x T threads, B barriers in a row (sense-free monotonic counter, agent-scope atomics, one poller
//       per workgroup): wall time / B = the barrier's latency with every workgroup arriving together;
//   (b) the same G x T kernel doing NOTHING, launched B times inside one hipGraph: time / B = a kernel boundary.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/grid_barrier_probe tools/probes/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, unsigned long long& spins) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned n = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++n > (1u << 22)) break;   // bounded: a workgroup that never arrives must not hang the box
    }
    spins += n;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(512) barrier_loop(unsigned* counter, int n_barriers, unsigned long long* spins_out) {
  unsigned long long spins = 0;
  for (int b = 0; b < n_barriers; ++b) grid_barrier(counter, (unsigned)(b + 1) * gridDim.x, spins);
  if (threadIdx.x == 0 && blockIdx.x == 0) *spins_out = spins;
}

__global__ void __launch_bounds__(512) empty_kernel(unsigned* p) {
  if (p == nullptr && threadIdx.x == 12345) *p = 1;
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 512, B = argc > 3 ? atoi(argv[3]) : 200;
  unsigned* counter; unsigned long long* spins;
  CK(hipMalloc(&counter, 256)); CK(hipMalloc(&spins, 8));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemsetAsync(counter, 0, 256, st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(barrier_loop, dim3(G), dim3(T), 0, st, counter, B, spins);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long sp; CK(hipMemcpy(&sp, spins, 8, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"grid_barrier\", \"wgs\": %d, \"threads\": %d, \"barriers\": %d, \"us_per_barrier\": %.3f, \"polls_wg0_per_barrier\": %.1f}\n",
           G, T, B, ms * 1e3 / B, (double)sp / B);
  }
  // kernel boundaries inside a graph
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int b = 0; b < B; ++b) hipLaunchKernelGGL(empty_kernel, dim3(G), dim3(T), 0, st, counter);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"probe\": \"graph_kernel_boundary\", \"wgs\": %d, \"threads\": %d, \"launches\": %d, \"us_per_launch\": %.3f}\n", G, T, B, ms * 1e3 / B);
  }
  return 0;
}
