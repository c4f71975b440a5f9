// clock_probe.hip -- what does the shader clock do while the GEMMs run?
//
// The fraction-of-peak figures in DESIGN.md price every MFMA kernel against 2.5 PFLOP/s, i.e. against
// 256 CUs x 4 SIMDs x one 32x32x16 MFMA per 32 cycles at the 2.4 GHz boost clock.  This harness
// measures the clock the chip actually sustains under each load: a one-wave MONITOR kernel on its own
// stream samples (s_memtime = shader-clock counter, s_memrealtime = 100 MHz constant clock) every few
// microseconds while the load runs on another stream; MHz = d(memtime) / d(memrealtime) x 100.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/probes/bin/clock_probe \
//         tools/probes/clock_probe.hip -Lscalellm_amd/csrc -lslm_hip -Wl,-rpath,$PWD/scalellm_amd/csrc
//   tools/probes/bin/clock_probe            (prints one JSON line per load)
//
// Loads: idle; a register-only MFMA loop on ZERO operands and on RANDOM operands (same instruction
// stream, different toggle activity); a device-to-device copy (HBM stream); the library's int4 GEMM
// at the headline gate_up shape, M = 256 (wave-specialised kernel) and M = 4096 (256 x 256 kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "slm_hip.h"

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void monitor_kernel(unsigned long long* out, int n, int sleeps) {
  for (int i = 0; i < n; ++i) {
    const unsigned long long c = __builtin_amdgcn_s_memtime();
    const unsigned long long w = __builtin_amdgcn_s_memrealtime();
    out[2 * i] = c;
    out[2 * i + 1] = w;
    for (int j = 0; j < sleeps; ++j) __builtin_amdgcn_s_sleep(127);
  }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// register-only MFMA loop: 4 independent accumulator chains per wave, 8 waves per workgroup
__global__ void __launch_bounds__(512) mfma_loop_kernel(const uint32_t* seed, float* sink, int iters) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 ra, rb;
  const uint32_t s = seed[threadIdx.x & 63];
  ra.x = s * 2654435761u; ra.y = s * 40503u + 17u; ra.z = s ^ 0x9e3779b9u; ra.w = s * 977u;
  rb.x = s * 31u + 7u; rb.y = s * 2246822519u; rb.z = s ^ 0x85ebca6bu; rb.w = s * 131u;
  // keep the bf16 exponents small (values ~1): clear the top exponent bits of every half
  const uint32_t m = s == 0u ? 0u : 0x3fff3fffu;
  ra.x &= m; ra.y &= m; ra.z &= m; ra.w &= m;
  rb.x &= m; rb.y &= m; rb.z &= m; rb.w &= m;
  const bf16x8 a = __builtin_bit_cast(bf16x8, ra), b = __builtin_bit_cast(bf16x8, rb);
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
  }
  float r = 0.f;
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 16; ++i) r += acc[q][i];
  if (r == 12345.678f) sink[0] = r;
}


// read-only HBM stream: every wave walks its own contiguous region 1 KiB per instruction, DEPTH loads
// in flight per lane (nt: touched once), xor-folded so nothing is optimised away
template <int DEPTH>
__global__ void __launch_bounds__(512) read_stream_kernel(const uint32_t* src, size_t words_per_wave, uint32_t* sink) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
  const u32x4* p = reinterpret_cast<const u32x4*>(src + wave * words_per_wave) + (threadIdx.x & 63);
  const size_t n = words_per_wave / 256;  // 1-KiB steps
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (size_t i = 0; i + DEPTH <= n; i += DEPTH) {
    u32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_nontemporal_load(p + (i + d) * 64);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

// the same stream, grid-strided: consecutive waves read consecutive KiB (wave w takes KiB w, w + W, ...)
template <int DEPTH>
__global__ void __launch_bounds__(512) read_stream_strided_kernel(const uint32_t* src, size_t total_kib, uint32_t* sink) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const size_t W = (size_t)gridDim.x * 8;
  const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
  const u32x4* p = reinterpret_cast<const u32x4*>(src) + (threadIdx.x & 63);
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (size_t i = wave; i + (DEPTH - 1) * W < total_kib; i += DEPTH * W) {
    u32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_nontemporal_load(p + (i + d * W) * 64);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

struct Stats {
  double mhz_med, mhz_min, mhz_max, window_us;
  int n;
};

static Stats analyse(const std::vector<unsigned long long>& s, double t0_ticks, double t1_ticks) {
  // samples whose wall time lies inside [t0, t1] (100 MHz ticks)
  std::vector<double> mhz;
  for (size_t i = 1; i * 2 + 1 < s.size(); ++i) {
    const double w0 = (double)s[2 * i - 1], w1 = (double)s[2 * i + 1];
    if (w0 < t0_ticks || w1 > t1_ticks || w1 <= w0) continue;
    const double dc = (double)(s[2 * i] - s[2 * i - 2]);
    mhz.push_back(dc / (w1 - w0) * 100.0);
  }
  Stats st{0, 0, 0, (t1_ticks - t0_ticks) / 100.0, (int)mhz.size()};
  if (mhz.empty()) return st;
  std::sort(mhz.begin(), mhz.end());
  st.mhz_med = mhz[mhz.size() / 2];
  st.mhz_min = mhz[mhz.size() / 20];      // 5th percentile
  st.mhz_max = mhz[mhz.size() * 19 / 20];  // 95th percentile
  return st;
}

template <typename F>
static void run_load(const char* name, double work, const char* work_unit, F&& launch_on, int warm, int reps) {
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const int n = 6000, sleeps = 1;  // ~127 x 64 cycles = 3-4 us per sample
  unsigned long long* dmon;
  CK(hipMalloc(&dmon, sizeof(unsigned long long) * 2 * n));
  CK(hipMemset(dmon, 0, sizeof(unsigned long long) * 2 * n));
  for (int i = 0; i < warm; ++i) launch_on(sa);
  CK(hipStreamSynchronize(sa));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // wall-clock stamps of the load window from the device side too: tiny kernels before / after
  unsigned long long* dwin;
  CK(hipMalloc(&dwin, 4 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(monitor_kernel, dim3(1), dim3(1), 0, sb, dmon, n, sleeps);
  // let the monitor collect ~2 ms of idle samples first
  hipLaunchKernelGGL(monitor_kernel, dim3(1), dim3(1), 0, sa, dwin, 1, 600);
  CK(hipEventRecord(e0, sa));
  for (int i = 0; i < reps; ++i) launch_on(sa);
  CK(hipEventRecord(e1, sa));
  hipLaunchKernelGGL(monitor_kernel, dim3(1), dim3(1), 0, sa, dwin + 2, 1, 0);
  CK(hipStreamSynchronize(sa));
  CK(hipStreamSynchronize(sb));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> s(2 * n), w(4);
  CK(hipMemcpy(s.data(), dmon, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(w.data(), dwin, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost));
  // window: from the end of the idle spacer (its stamp is taken at its START, so add its sleep
  // time by using the load's own end stamp minus the event time) to the end stamp
  const double t1 = (double)w[3];
  const double t0 = t1 - (double)ms * 1e-3 * 1e8;
  const double skip = (t1 - t0) * 0.15;  // drop the ramp at both ends
  const Stats busy = analyse(s, t0 + skip, t1 - skip);
  const Stats idle = analyse(s, (double)s[1], t0 - 20000.0);
  const double us_per = (double)ms * 1e3 / reps;
  printf("{\"load\": \"%s\", \"reps\": %d, \"us_per_launch\": %.2f, \"%s\": %.1f, \"clock_mhz_median\": %.0f, "
         "\"clock_mhz_p05\": %.0f, \"clock_mhz_p95\": %.0f, \"samples\": %d, \"idle_before_mhz_median\": %.0f, "
         "\"idle_samples\": %d}\n",
         name, reps, us_per, work_unit, work / us_per, busy.mhz_med, busy.mhz_min, busy.mhz_max, busy.n,
         idle.mhz_med, idle.n);
  fflush(stdout);
  CK(hipFree(dmon));
  CK(hipFree(dwin));
  CK(hipStreamDestroy(sa));
  CK(hipStreamDestroy(sb));
}

int main() {
  CK(hipSetDevice(0));
  // ---- idle
  run_load("idle (empty launches)", 0.0, "none", [&](hipStream_t st) {
    hipLaunchKernelGGL(monitor_kernel, dim3(1), dim3(1), 0, st, (unsigned long long*)nullptr, 0, 0);
  }, 2, 2000);

  // ---- register-only MFMA loops
  uint32_t hseed[64];
  float* sink;
  uint32_t* dseed;
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&dseed, sizeof(hseed)));
  const int iters = 2048;
  const double mfma_flop = 2048.0 * 8 * 4.0 * iters * 2.0 * 32 * 32 * 16;  // workgroups x waves x chains x iters
  for (int variant = 0; variant < 2; ++variant) {
    for (int i = 0; i < 64; ++i) hseed[i] = variant == 0 ? 0u : (uint32_t)(i * 2654435761u + 12345u) | 1u;
    CK(hipMemcpy(dseed, hseed, sizeof(hseed), hipMemcpyHostToDevice));
    run_load(variant == 0 ? "mfma 32x32x16 bf16 loop, ZERO operands" : "mfma 32x32x16 bf16 loop, RANDOM operands",
             mfma_flop / 1e6, "tflops", [&](hipStream_t st) {
               hipLaunchKernelGGL(mfma_loop_kernel, dim3(2048), dim3(512), 0, st, dseed, sink, iters);
             }, 1, 6);
  }


  // ---- HBM read stream: 8 GiB walked once per launch (far beyond the 256 MiB Infinity Cache)
  {
    const size_t bytes = (size_t)8 << 30;
    uint32_t* src;
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(src, 1, bytes));
    for (int wgs : {256, 512, 1024, 2048}) {
      const size_t waves = (size_t)wgs * 8;
      const size_t words_per_wave = bytes / 4 / waves;
      char name[128];
      snprintf(name, sizeof(name), "read-only stream 8 GiB, %d workgroups x 8 waves, 8 x 16 B in flight per lane", wgs);
      run_load(name, (double)bytes / 1e6, "tbps", [&](hipStream_t st) {
        hipLaunchKernelGGL(read_stream_kernel<8>, dim3(wgs), dim3(512), 0, st, src, words_per_wave, (uint32_t*)sink);
      }, 1, 8);
    }
    for (int wgs : {256, 512, 2048}) {
      char name[160];
      snprintf(name, sizeof(name), "read-only stream 8 GiB, grid-strided (adjacent waves read adjacent KiB), %d workgroups, 8 in flight", wgs);
      run_load(name, (double)bytes / 1e6, "tbps", [&](hipStream_t st) {
        hipLaunchKernelGGL(read_stream_strided_kernel<8>, dim3(wgs), dim3(512), 0, st, src, bytes / 1024, (uint32_t*)sink);
      }, 1, 8);
    }
    for (int wgs : {256, 512}) {
      const size_t waves = (size_t)wgs * 8;
      const size_t words_per_wave = bytes / 4 / waves;
      char name[160];
      snprintf(name, sizeof(name), "read-only stream 8 GiB, %d workgroups x 8 waves, 16 x 16 B in flight per lane", wgs);
      run_load(name, (double)bytes / 1e6, "tbps", [&](hipStream_t st) {
        hipLaunchKernelGGL(read_stream_kernel<16>, dim3(wgs), dim3(512), 0, st, src, words_per_wave, (uint32_t*)sink);
      }, 1, 8);
    }
    CK(hipFree(src));
  }

  // ---- HBM stream: device-to-device copy of 1 GiB
  {
    const size_t bytes = (size_t)1 << 30;
    void *src, *dst;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&dst, bytes));
    CK(hipMemset(src, 1, bytes));
    run_load("device-to-device copy 1 GiB", 2.0 * bytes / 1e6, "tbps", [&](hipStream_t st) {
      CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    }, 2, 20);
    CK(hipFree(src));
    CK(hipFree(dst));
  }

  // ---- the library's int4 GEMM, gate_up 4096 x 28672
  {
    const int64_t K = 4096, N = 28672, Mmax = 4096;
    const size_t wbytes = (size_t)K * N / 2, szbytes = (size_t)(K / 128) * N * 4;
    const int n_rot = 6;  // 6 x 59 MB of distinct weights: larger than the Infinity Cache
    std::vector<void*> wq(n_rot), sz(n_rot);
    std::vector<uint32_t> h(wbytes / 4);
    uint32_t x = 123456789u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    std::vector<uint32_t> hs(szbytes / 4);
    for (auto& v : hs) {
      x = x * 1664525u + 1013904223u;
      const uint32_t scale = 0x3c00u + (x >> 26);          // bf16 ~ 0.0078 .. 0.0155
      const uint32_t zm = 0x4300u + ((x >> 8) & 15u);      // bf16 128 + zero, zero in 0..15
      v = scale | (zm << 16);
    }
    for (int r = 0; r < n_rot; ++r) {
      CK(hipMalloc(&wq[r], wbytes));
      CK(hipMalloc(&sz[r], szbytes));
      CK(hipMemcpy(wq[r], h.data(), wbytes, hipMemcpyHostToDevice));
      CK(hipMemcpy(sz[r], hs.data(), szbytes, hipMemcpyHostToDevice));
    }
    void *a, *c, *ws;
    CK(hipMalloc(&a, (size_t)Mmax * K * 2));
    CK(hipMalloc(&c, (size_t)Mmax * N * 2));
    std::vector<uint16_t> ha((size_t)Mmax * K);
    for (auto& v : ha) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00u | (x >> 25) | ((x >> 9) & 0x8000u)); }
    CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    const size_t ws_bytes = (size_t)1 << 30;
    CK(hipMalloc(&ws, ws_bytes));
    for (int64_t M : {256, 1024, 4096}) {
      int rot = 0;
      slm_w4_gemm_args g;
      memset(&g, 0, sizeof(g));
      g.a = a; g.c = c; g.M = M; g.K = K; g.N = N; g.lda = K; g.ldc = N; g.group_size = 128;
      g.dtype = SLM_BF16; g.workspace = ws; g.workspace_bytes = ws_bytes;
      char name[128];
      snprintf(name, sizeof(name), "slm_w4a16_gemm gate_up 4096x28672 bf16, M = %lld", (long long)M);
      const double flop = 2.0 * M * K * N;
      run_load(name, flop / 1e6, "tflops", [&](hipStream_t st) {
        g.wq = wq[rot]; g.sz = sz[rot];
        rot = (rot + 1) % n_rot;
        const int rc = slm_w4a16_gemm(&g, st);
        if (rc != 0) { fprintf(stderr, "slm_w4a16_gemm rc %d\n", rc); exit(1); }
      }, 6, M == 256 ? 120 : (M == 1024 ? 40 : 12));
    }
  }
  return 0;
}
