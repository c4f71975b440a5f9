// valu_rate_probe.hip -- issue rate of the VALU instructions the int4 dequant is made of (gfx950).
// One workgroup per CU; W waves per SIMD; every wave runs a long unrolled chain-free sequence of ONE
// instruction kind (16 independent destinations) and the wall time gives cycles per instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/valu_rate_probe tools/probes/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ void __launch_bounds__(1024) probe(uint32_t* out, int iters, uint32_t seed, float scale) {
  uint32_t r[16];
  f32x2 f[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    r[i] = seed + threadIdx.x * 17 + i;
    f[i] = f32x2{(float)i, scale};
  }
  uint32_t src = seed ^ threadIdx.x;
  f32x2 s2 = {scale, scale}, c2 = {1.f, 2.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if constexpr (KIND == 0) {  // v_cvt_scalef32_pk_bf16_fp8
          bf16x2 v;
          asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2" : "=v"(v) : "v"(src), "v"(scale));
          r[i] = __builtin_bit_cast(uint32_t, v);
        } else if constexpr (KIND == 1) {  // v_cvt_f32_ubyte0
          float v;
          asm volatile("v_cvt_f32_ubyte0_e32 %0, %1" : "=v"(v) : "v"(src));
          r[i] = __builtin_bit_cast(uint32_t, v);
        } else if constexpr (KIND == 2) {  // v_pk_fma_f32
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(s2), "v"(c2), "v"(c2));
        } else if constexpr (KIND == 3) {  // v_cvt_pk_bf16_f32
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r[i]) : "v"(scale), "v"(scale));
        } else if constexpr (KIND == 4) {  // v_perm_b32
          asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(src), "v"(seed), "v"(src));
        } else if constexpr (KIND == 5) {  // v_and_b32
          asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(r[i]) : "v"(src), "v"(seed));
        } else if constexpr (KIND == 6) {  // v_and_or_b32
          asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(src), "v"(seed), "v"(src));
        } else if constexpr (KIND == 7) {  // v_fma_f32
          float v;
          asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(scale), "v"(scale), "v"(scale));
          r[i] = __builtin_bit_cast(uint32_t, v);
        } else if constexpr (KIND == 8) {  // v_dot2_f32_bf16
          float v;
          asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(v) : "v"(src), "v"(seed), "v"(scale));
          r[i] = __builtin_bit_cast(uint32_t, v);
        } else if constexpr (KIND == 9) {  // v_cvt_scalef32_pk_bf16_fp8, op_sel high half
          bf16x2 v;
          asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2 op_sel:[1,0,0]" : "=v"(v) : "v"(src), "v"(scale));
          r[i] = __builtin_bit_cast(uint32_t, v);
        }
      }
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc ^= r[i] ^ __builtin_bit_cast(uint32_t, f[i].x);
  if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// value check of the fp8 route: byte 0x0q (q = 0..15) as E4M3 = q * 2^-9, so scale = s * 512 gives bf16(q * s)
__global__ void check(uint32_t* out, float s) {
  const uint32_t q = threadIdx.x & 15;
  const uint32_t src = q | ((15 - q) << 8) | (q << 16) | ((15 - q) << 24);
  bf16x2 lo = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src, s * 512.f, false);
  bf16x2 hi = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src, s * 512.f, true);
  out[threadIdx.x * 2] = __builtin_bit_cast(uint32_t, lo);
  out[threadIdx.x * 2 + 1] = __builtin_bit_cast(uint32_t, hi);
}

static float bf16_to_f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f_to_bf16_rn(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

template <int KIND>
static void run(const char* name, int waves_per_simd, uint32_t* d_out, int n_cu, double mhz) {
  const int iters = 4000;
  const int threads = 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(n_cu), dim3(threads), 0, 0, d_out, 10, 1u, 1.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(n_cu), dim3(threads), 0, 0, d_out, iters, 1u, 1.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)iters * 64 * waves_per_simd;
  const double cyc = ms * 1e-3 * mhz * 1e6 / instr_per_simd;
  printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"cycles_per_instr_per_simd_at_%.0fMHz\": %.2f}\n",
         name, waves_per_simd, ms, mhz, cyc);
}

int main(int argc, char** argv) {
  double mhz = argc > 1 ? atof(argv[1]) : 2400.0;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  uint32_t* d_out; hipMalloc(&d_out, 1 << 20);
  // value check
  const float scales[3] = {0.0123f, 1.0f, 0.00037f};
  int bad = 0;
  for (float s0 : scales) {
    const float s = bf16_to_f(f_to_bf16_rn(s0));
    hipLaunchKernelGGL(check, dim3(1), dim3(16), 0, 0, d_out, s);
    std::vector<uint32_t> h(32);
    hipMemcpy(h.data(), d_out, 128, hipMemcpyDeviceToHost);
    for (int q = 0; q < 16; ++q) {
      const uint16_t e0 = f_to_bf16_rn(q * s), e1 = f_to_bf16_rn((15 - q) * s);
      const uint16_t g0 = h[2 * q] & 0xffff, g1 = h[2 * q] >> 16, g2 = h[2 * q + 1] & 0xffff, g3 = h[2 * q + 1] >> 16;
      if (g0 != e0 || g1 != e1 || g2 != e0 || g3 != e1) { ++bad; printf("mismatch q=%d s=%g: got %04x %04x %04x %04x want %04x %04x\n", q, s, g0, g1, g2, g3, e0, e1); }
    }
  }
  printf("{\"check\": \"fp8 byte 0x0q * (512 s) == bf16_rn(q s)\", \"mismatches\": %d}\n", bad);
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_cvt_scalef32_pk_bf16_fp8", w, d_out, n_cu, mhz);
    run<9>("v_cvt_scalef32_pk_bf16_fp8 op_sel hi", w, d_out, n_cu, mhz);
    run<1>("v_cvt_f32_ubyte0", w, d_out, n_cu, mhz);
    run<2>("v_pk_fma_f32", w, d_out, n_cu, mhz);
    run<3>("v_cvt_pk_bf16_f32", w, d_out, n_cu, mhz);
    run<4>("v_perm_b32", w, d_out, n_cu, mhz);
    run<5>("v_and_b32", w, d_out, n_cu, mhz);
    run<6>("v_and_or_b32", w, d_out, n_cu, mhz);
    run<7>("v_fma_f32", w, d_out, n_cu, mhz);
    run<8>("v_dot2_f32_bf16", w, d_out, n_cu, mhz);
  }
  return 0;
}
