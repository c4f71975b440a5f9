// stream_probe.hip -- standalone micro-benchmark (hipcc, no torch): what does the MEMORY SIDE give
// a small-M int4 GEMM?  Streams a packed weight matrix with exactly the access pattern of the
// kt-major layout (wq[K/64][N/32][64 lanes][4] u32: one contiguous KiB per (kt, column tile), one
// wave per column tile walking kt) and nothing else -- no unpack, no MFMA, no LDS -- for a range of
// ring depths / waves per workgroup / K splits, on rotating buffers larger than the Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe tools/probes/stream_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } \
  } while (0)

// one wave per column tile; blockDim = 64 * WPB; grid = (n_tiles / WPB) * split
template <int RING, bool NT>
__global__ void __launch_bounds__(512) stream_kernel(const uint32_t* __restrict__ wq, int n_tiles, int n_kt,
                                                     int split, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int nb = blockIdx.x % (n_tiles / wpb), ks = blockIdx.x / (n_tiles / wpb);
  const int nt = nb * wpb + wave;
  const int per = (n_kt + split - 1) / split;
  const int k0 = ks * per, k1 = min(n_kt, k0 + per);
  const char* base = reinterpret_cast<const char*>(wq) + ((size_t)nt * 64 + lane) * 16;
  const size_t stride = (size_t)n_tiles * 1024;
  u32x4 ring[RING];
  u32x4 acc = {0, 0, 0, 0};
  auto ld = [&](int kt) {
    const int kc = kt < k1 ? kt : k1 - 1;
    const u32x4* p = reinterpret_cast<const u32x4*>(base + (size_t)kc * stride);
    return NT ? __builtin_nontemporal_load(p) : *p;
  };
#pragma unroll
  for (int r = 0; r < RING; ++r) ring[r] = ld(k0 + r);
  for (int kt = k0; kt < k1; kt += RING) {
#pragma unroll
    for (int r = 0; r < RING; ++r) {
      acc ^= ring[r];
      ring[r] = ld(kt + RING + r);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x;
}

template <int RING, bool NT>
float run(const std::vector<uint32_t*>& bufs, int n_tiles, int n_kt, int wpb, int split, uint32_t* out, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const dim3 grid((n_tiles / wpb) * split), blk(64 * wpb);
  for (size_t i = 0; i < bufs.size(); ++i)
    hipLaunchKernelGGL((stream_kernel<RING, NT>), grid, blk, 0, 0, bufs[i], n_tiles, n_kt, split, out);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0));
    for (size_t i = 0; i < bufs.size(); ++i)
      hipLaunchKernelGGL((stream_kernel<RING, NT>), grid, blk, 0, 0, bufs[i], n_tiles, n_kt, split, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms * 1e3f / bufs.size());
  }
  return best;
}

int main() {
  struct Shape { const char* name; int K, N; };
  const Shape shapes[] = {{"qkv", 4096, 6144}, {"o", 4096, 4096}, {"gate_up", 4096, 28672}, {"down", 14336, 4096}};
  uint32_t* out;
  CHECK(hipMalloc(&out, 64 << 20));
  for (const Shape& s : shapes) {
    const size_t bytes = (size_t)s.K * s.N / 2;
    const int n_rot = (int)((400u << 20) / bytes) + 1;
    std::vector<uint32_t*> bufs(n_rot);
    for (auto& b : bufs) {
      CHECK(hipMalloc(&b, bytes));
      CHECK(hipMemset(b, 0x5a, bytes));
    }
    const int n_tiles = s.N / 32, n_kt = s.K / 64;
    for (int wpb : {4, 8}) {
      for (int split : {1, 2, 4, 8, 16}) {
        if ((n_tiles / wpb) * split > 4096 || n_kt / split < 8) continue;
        const float t4 = run<4, true>(bufs, n_tiles, n_kt, wpb, split, out, 5);
        const float t8 = run<8, true>(bufs, n_tiles, n_kt, wpb, split, out, 5);
        const float t16 = run<16, true>(bufs, n_tiles, n_kt, wpb, split, out, 5);
        const float t8p = run<8, false>(bufs, n_tiles, n_kt, wpb, split, out, 5);
        printf("{\"probe\": \"stream\", \"shape\": \"%s\", \"MB\": %.1f, \"wpb\": %d, \"split\": %d, \"wgs\": %d, "
               "\"us_ring4\": %.2f, \"us_ring8\": %.2f, \"us_ring16\": %.2f, \"us_ring8_plain\": %.2f, \"gbps_best\": %.0f}\n",
               s.name, bytes / 1e6, wpb, split, (n_tiles / wpb) * split, t4, t8, t16, t8p,
               bytes / fminf(fminf(t4, t8), t16) / 1e3);
        fflush(stdout);
      }
    }
    for (auto& b : bufs) CHECK(hipFree(b));
  }
  return 0;
}
