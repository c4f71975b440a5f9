// common.h -- shared device helpers for libslm_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slm_hip.h"

namespace slm {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

struct bf16_tag {};
struct f16_tag {};

// ---- packed 16-bit pair <-> fp32 ------------------------------------------------
template <typename T>
__device__ __forceinline__ float lo_f32(uint32_t w);
template <typename T>
__device__ __forceinline__ float hi_f32(uint32_t w);

template <>
__device__ __forceinline__ float lo_f32<bf16_tag>(uint32_t w) {
  return __builtin_bit_cast(float, w << 16);
}
template <>
__device__ __forceinline__ float hi_f32<bf16_tag>(uint32_t w) {
  return __builtin_bit_cast(float, w & 0xffff0000u);
}
template <>
__device__ __forceinline__ float lo_f32<f16_tag>(uint32_t w) {
  return (float)__builtin_bit_cast(f16x2_t, w)[0];
}
template <>
__device__ __forceinline__ float hi_f32<f16_tag>(uint32_t w) {
  return (float)__builtin_bit_cast(f16x2_t, w)[1];
}

// fp32 pair -> packed 16-bit pair, round-to-nearest-even
template <typename T>
__device__ __forceinline__ uint32_t pack2(float a, float b);
template <>
__device__ __forceinline__ uint32_t pack2<bf16_tag>(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <>
__device__ __forceinline__ uint32_t pack2<f16_tag>(float a, float b) {
  f16x2_t h = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(uint32_t, h);
}

template <typename T>
__device__ __forceinline__ uint16_t pack1(float a) {
  return (uint16_t)(pack2<T>(a, 0.f) & 0xffffu);
}

// acc += a.lo*b.lo + a.hi*b.hi  (v_dot2c_f32_bf16 / v_dot2c_f32_f16, fp32 accumulate)
template <typename T>
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc);
template <>
__device__ __forceinline__ float dot2<bf16_tag>(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a),
                                         __builtin_bit_cast(bf16x2_t, b), acc, false);
}
template <>
__device__ __forceinline__ float dot2<f16_tag>(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b),
                                acc, false);
}

// ---- cross-lane (wave64) --------------------------------------------------------
// DPP lane exchange inside a row of 16 lanes; full-rate VALU, no LDS traffic.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // i <-> 7-i inside each 8 lanes
constexpr int DPP_ROW_MIRROR = 0x140;       // i <-> 15-i inside each 16 lanes

// all-reduce sum across aligned groups of W lanes (W = 4, 8, 16, 32)
template <int W>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_f32<DPP_QUAD_XOR1>(v);
  v += dpp_f32<DPP_QUAD_XOR2>(v);
  if constexpr (W >= 8) v += dpp_f32<DPP_ROW_HALF_MIRROR>(v);
  if constexpr (W >= 16) v += dpp_f32<DPP_ROW_MIRROR>(v);
  if constexpr (W >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (W >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// all-reduce N independent values across aligned groups of W lanes, one DPP step at a time over
// all N values (keeps N independent dependency chains adjacent in program order)
template <int W, int N>
__device__ __forceinline__ void group_sum_many(float* v) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_f32<DPP_QUAD_XOR1>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_f32<DPP_QUAD_XOR2>(v[i]);
  if constexpr (W >= 8) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f32<DPP_ROW_HALF_MIRROR>(v[i]);
  }
  if constexpr (W >= 16) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f32<DPP_ROW_MIRROR>(v[i]);
  }
  if constexpr (W >= 32) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __shfl_xor(v[i], 16, 64);
  }
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// silu(gate) * up in fp32 on values that are already exact in T: THE formula of the SiLU*mul glue
// kernel (kernel::act_and_mul, reference src/kernels/activation_kernels.cu:84), shared with the
// GEMM epilogues that fuse it so that both paths give identical bits
__device__ __forceinline__ float silu_mul1(const float g, const float u) {
  const float sig = __builtin_amdgcn_rcpf(1.0f + fast_exp2(-g * 1.4426950408889634f));
  float r = g * sig * u;
  // the product is an fp32 value in a register before any conversion to T: without this the
  // compiler may fold "multiply, then round to fp16" into one v_fma_mixlo_f16 (single rounding) in
  // some callers and not in others (v_pk_mul_f32 + v_cvt), and the fused / unfused paths would
  // differ in the last bit on ties
  asm("" : "+v"(r));
  return r;
}

// the fused-epilogue form: fp32 accumulators -> rounded to T (what the unfused GEMM would have
// stored) -> silu_mul1
template <typename T>
__device__ __forceinline__ float silu_mul_acc(const float gate_acc, const float up_acc) {
  return silu_mul1(lo_f32<T>((uint32_t)pack1<T>(gate_acc)), lo_f32<T>((uint32_t)pack1<T>(up_acc)));
}

// 8 x silu_mul1, rounded RN to T
template <typename T>
__device__ __forceinline__ u32x4 silu_mul8(const u32x4 g, const u32x4 u) {
  const float gf[8] = {lo_f32<T>(g.x), hi_f32<T>(g.x), lo_f32<T>(g.y), hi_f32<T>(g.y),
                       lo_f32<T>(g.z), hi_f32<T>(g.z), lo_f32<T>(g.w), hi_f32<T>(g.w)};
  const float uf[8] = {lo_f32<T>(u.x), hi_f32<T>(u.x), lo_f32<T>(u.y), hi_f32<T>(u.y),
                       lo_f32<T>(u.z), hi_f32<T>(u.z), lo_f32<T>(u.w), hi_f32<T>(u.w)};
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = silu_mul1(gf[j], uf[j]);
  u32x4 r;
  r.x = pack2<T>(o[0], o[1]); r.y = pack2<T>(o[2], o[3]);
  r.z = pack2<T>(o[4], o[5]); r.w = pack2<T>(o[6], o[7]);
  return r;
}

// sum over split-K partial slabs of 4 consecutive columns, in THE order of the split-K reduce
// kernel (w4.hip) -- shared with the RMSNorm that can absorb the reduction (glue.hip), so that
// "GEMM -> reduce -> norm" and "GEMM (deferred) -> norm" give identical bits
__device__ __forceinline__ f32x4 splitk_sum4(const float* __restrict__ src, const int64_t slab,
                                             const int split_k) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= split_k; k += 8) {  // 8 independent 16-B loads in flight per thread
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + (k + i) * slab);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  for (; k + 2 <= split_k; k += 2) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + k * slab);
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + (k + 1) * slab);
    s += a + b;
  }
  if (k < split_k) s += *reinterpret_cast<const f32x4*>(src + k * slab);
  return s;
}

// RMSNorm row arithmetic shared by rms_norm_kernel (glue.hip) and the fused all-reduce
// (allreduce.hip) -- normalization.h:17-52.  The fma is EXPLICIT: left to -ffp-contract the compiler
// fuses or not per kernel, and the two paths would differ in the last bit of the sum of squares.
__device__ __forceinline__ float rms_sumsq8(const float (&f)[8], float ss) {
#pragma unroll
  for (int j = 0; j < 8; ++j) ss = __builtin_fmaf(f[j], f[j], ss);
  return ss;
}
// y = T(T(h * rs) * w): the normalised value is rounded to T before the weight (normalization.h:27-29)
template <typename T>
__device__ __forceinline__ u32x4 rms_apply8(const float (&h)[8], const float rs, const u32x4 wv) {
  const float wf[8] = {lo_f32<T>(wv.x), hi_f32<T>(wv.x), lo_f32<T>(wv.y), hi_f32<T>(wv.y),
                       lo_f32<T>(wv.z), hi_f32<T>(wv.z), lo_f32<T>(wv.w), hi_f32<T>(wv.w)};
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float n16 = lo_f32<T>((uint32_t)pack1<T>(h[j] * rs));
    o[j] = n16 * wf[j];
  }
  u32x4 r;
  r.x = pack2<T>(o[0], o[1]); r.y = pack2<T>(o[2], o[3]);
  r.z = pack2<T>(o[4], o[5]); r.w = pack2<T>(o[6], o[7]);
  return r;
}

// hipGetLastError() is per-thread sticky state shared with every other HIP user in the process
// (torch, RCCL): clear stale errors when an entry point starts, so that hip_check_launch()
// reports only OUR launch.
inline int& hip_last_error_slot() {
  static thread_local int e = 0;
  return e;
}
inline void hip_clear_error() { (void)hipGetLastError(); }
inline int hip_check_launch() {
  const hipError_t e = hipGetLastError();
  hip_last_error_slot() = (int)e;
  return e == hipSuccess ? SLM_OK : SLM_ERR_LAUNCH;
}

inline bool is_pow2(int64_t x) { return x > 0 && (x & (x - 1)) == 0; }
inline int ilog2(int64_t x) {
  int r = 0;
  while ((1ll << (r + 1)) <= x) ++r;
  return r;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace slm
