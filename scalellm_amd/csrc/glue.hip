// glue.hip -- the small memory-bound ops either side of the hot path inside one decoder layer
// (SURVEY 8f rows f1/f2), hand-written for gfx950: 16-byte vector accesses, wave64 reductions.
//
//  slm_rms_norm        <- kernel::rms_norm / rms_norm_residual  (reference
//                         src/kernels/layernorm_kernels.cu:15,125; math src/layers/normalization.h
//                         :17-52: fp32 mean of squares, x * rsqrt(mean + eps) cast to T, * weight)
//  slm_rope_kv_append  <- kernel::apply_rotary_pos_emb (src/kernels/pos_embedding_kernels.cu:35-121)
//                         FUSED with kernel::set_kv_cache (src/kernels/kv_cache_kernels.cu:9-78):
//                         the order rope -> append -> attend is fixed by
//                         src/layers/attention/attention.cpp:36-42, so K is rotated in registers and
//                         written once to its in-place location and once to its cache slot.
//  slm_silu_mul        <- kernel::act_and_mul (silu), src/kernels/activation_kernels.cu:84.
#include "common.h"

namespace slm {

__device__ __forceinline__ float wave_sum64(float v) { return group_sum<64>(v); }

// one workgroup (256 threads) per token; dim % 8 == 0
template <typename T>
__global__ void __launch_bounds__(256) rms_norm_kernel(uint16_t* __restrict__ out,
                                                       const uint16_t* __restrict__ x,
                                                       const uint16_t* __restrict__ weight,
                                                       uint16_t* __restrict__ residual, int64_t dim,
                                                       float eps, const float* __restrict__ part,
                                                       int n_splits, int64_t slab) {
  __shared__ float red[4];
  const int64_t tok = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t nvec = dim / 8;
  constexpr int MAXV = 8;  // up to 8 x 8 x 256 = 16384 columns
  float v[MAXV][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int64_t vi = tid + 256 * i;
    if (vi < nvec) {
      u32x4 a;
      if (part) {  // x = T(sum of the split-K partial slabs): what the reduce kernel would have left
        const f32x4 s0 = splitk_sum4(part + tok * dim + vi * 8, slab, n_splits);
        const f32x4 s1 = splitk_sum4(part + tok * dim + vi * 8 + 4, slab, n_splits);
        a.x = pack2<T>(s0.x, s0.y); a.y = pack2<T>(s0.z, s0.w);
        a.z = pack2<T>(s1.x, s1.y); a.w = pack2<T>(s1.z, s1.w);
      } else {
        a = *reinterpret_cast<const u32x4*>(x + tok * dim + vi * 8);
      }
      float f[8] = {lo_f32<T>(a.x), hi_f32<T>(a.x), lo_f32<T>(a.y), hi_f32<T>(a.y),
                    lo_f32<T>(a.z), hi_f32<T>(a.z), lo_f32<T>(a.w), hi_f32<T>(a.w)};
      if (residual) {  // x = input + residual (fp32), residual = T(x): normalization.h:42-52
        const u32x4 r = *reinterpret_cast<const u32x4*>(residual + tok * dim + vi * 8);
        f[0] += lo_f32<T>(r.x); f[1] += hi_f32<T>(r.x); f[2] += lo_f32<T>(r.y); f[3] += hi_f32<T>(r.y);
        f[4] += lo_f32<T>(r.z); f[5] += hi_f32<T>(r.z); f[6] += lo_f32<T>(r.w); f[7] += hi_f32<T>(r.w);
        u32x4 w;
        w.x = pack2<T>(f[0], f[1]); w.y = pack2<T>(f[2], f[3]);
        w.z = pack2<T>(f[4], f[5]); w.w = pack2<T>(f[6], f[7]);
        *reinterpret_cast<u32x4*>(residual + tok * dim + vi * 8) = w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = f[j];
      ss = rms_sumsq8(f, ss);
    }
  }
  ss = wave_sum64(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float tot = red[0] + red[1] + red[2] + red[3];
  const float rs = rsqrtf(tot / (float)dim + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int64_t vi = tid + 256 * i;
    if (vi < nvec) {
      const u32x4 wv = *reinterpret_cast<const u32x4*>(weight + vi * 8);
      const u32x4 r = rms_apply8<T>(v[i], rs, wv);
      *reinterpret_cast<u32x4*>(out + tok * dim + vi * 8) = r;
    }
  }
}

// one workgroup per token.  Rotation pairs: non-interleaved (i, i + rot/2), interleaved (2i, 2i+1)
// (src/kernels/pos_embedding_kernels.cu:9-30).  cos_sin row = [cos(rot/2) | sin(rot/2)].
template <typename T, typename CS>
__global__ void __launch_bounds__(256) rope_kv_append_kernel(
    uint16_t* __restrict__ q, int64_t q_ts, uint16_t* __restrict__ k, int64_t k_ts,
    const uint16_t* __restrict__ v, int64_t v_ts, const int* __restrict__ positions,
    const CS* __restrict__ cos_sin, int rot_dim, int interleaved, const int* __restrict__ slot_ids,
    uint16_t* __restrict__ key_cache, uint16_t* __restrict__ value_cache, int n_heads,
    int n_kv_heads, int head_dim) {
  const int64_t tok = blockIdx.x;
  const int tid = threadIdx.x;
  const int half = rot_dim / 2;
  const int64_t slot = slot_ids ? (int64_t)slot_ids[tok] : -1;
  const int64_t row = (int64_t)n_kv_heads * head_dim;
  const CS* cs = cos_sin ? cos_sin + (int64_t)positions[tok] * rot_dim : nullptr;
  auto csval = [&](int i) -> float {
    if constexpr (sizeof(CS) == 4) return (float)cs[i];
    else return lo_f32<T>((uint32_t)cs[i]);
  };
  if (cs) {
    // Q: in place
    uint16_t* qt = q + tok * q_ts;
    for (int i = tid; i < n_heads * half; i += 256) {
      const int h = i / half, r = i % half;
      const int i0 = interleaved ? 2 * r : r, i1 = interleaved ? 2 * r + 1 : r + half;
      uint16_t* p = qt + (int64_t)h * head_dim;
      const float a = lo_f32<T>((uint32_t)p[i0]), b = lo_f32<T>((uint32_t)p[i1]);
      const float c = csval(r), s = csval(half + r);
      p[i0] = pack1<T>(a * c - b * s);
      p[i1] = pack1<T>(b * c + a * s);
    }
    // K: in place + cache slot
    uint16_t* kt = k + tok * k_ts;
    for (int i = tid; i < n_kv_heads * half; i += 256) {
      const int h = i / half, r = i % half;
      const int i0 = interleaved ? 2 * r : r, i1 = interleaved ? 2 * r + 1 : r + half;
      uint16_t* p = kt + (int64_t)h * head_dim;
      const float a = lo_f32<T>((uint32_t)p[i0]), b = lo_f32<T>((uint32_t)p[i1]);
      const float c = csval(r), s = csval(half + r);
      const uint16_t o0 = pack1<T>(a * c - b * s), o1 = pack1<T>(b * c + a * s);
      p[i0] = o0;
      p[i1] = o1;
      if (slot >= 0) {
        uint16_t* kc = key_cache + slot * row + (int64_t)h * head_dim;
        kc[i0] = o0;
        kc[i1] = o1;
      }
    }
  }
  if (slot >= 0) {
    // K pass-through dims (rot_dim < head_dim, or no rotary at all) and V: plain row copies
    const int rot_eff = cs ? rot_dim : 0;
    const uint16_t* kt = k + tok * k_ts;
    const int pass = head_dim - rot_eff;
    for (int i = tid; i < n_kv_heads * pass; i += 256) {
      const int h = i / pass, d = rot_eff + i % pass;
      key_cache[slot * row + (int64_t)h * head_dim + d] = kt[(int64_t)h * head_dim + d];
    }
    const uint16_t* vt = v + tok * v_ts;
    if ((row % 8) == 0 && (v_ts % 8) == 0) {
      for (int i = tid; i < row / 8; i += 256)
        *reinterpret_cast<u32x4*>(value_cache + slot * row + i * 8) =
            *reinterpret_cast<const u32x4*>(vt + i * 8);
    } else {
      for (int i = tid; i < row; i += 256) value_cache[slot * row + i] = vt[i];
    }
  }
}

// out[t, i] = silu(x[t, i]) * x[t, d + i]; d % 8 == 0
template <typename T>
__global__ void __launch_bounds__(256) silu_mul_kernel(uint16_t* __restrict__ out,
                                                       const uint16_t* __restrict__ x,
                                                       int64_t n_tokens, int64_t d) {
  const int64_t nvec = d / 8;
  const int64_t total = n_tokens * nvec;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int64_t t = idx / nvec, vi = idx % nvec;
    const u32x4 g = *reinterpret_cast<const u32x4*>(x + t * 2 * d + vi * 8);
    const u32x4 u = *reinterpret_cast<const u32x4*>(x + t * 2 * d + d + vi * 8);
    const u32x4 r = silu_mul8<T>(g, u);
    *reinterpret_cast<u32x4*>(out + t * d + vi * 8) = r;
  }
}

}  // namespace slm

using namespace slm;

extern "C" {

static int rms_norm_launch(void* out, const void* x, const float* part, int32_t n_splits,
                           const void* weight, void* residual, int64_t n_tokens, int64_t dim,
                           float eps, int32_t dtype, void* stream) {
  if (n_tokens == 0) return SLM_OK;
  if (!out || (!x && !part) || !weight || n_tokens < 0) return SLM_ERR_INVALID_ARG;
  if (part && n_splits < 1) return SLM_ERR_INVALID_ARG;
  if (dim <= 0 || dim % 8 || dim > 16384) return SLM_ERR_UNSUPPORTED;
  if (!aligned16(out) || (x && !aligned16(x)) || (part && !aligned16(part)) || !aligned16(weight) ||
      (residual && !aligned16(residual)))
    return SLM_ERR_ALIGNMENT;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const dim3 grid((unsigned)n_tokens), blk(256);
  const int64_t slab = n_tokens * dim;
  if (dtype == SLM_BF16)
    hipLaunchKernelGGL(rms_norm_kernel<bf16_tag>, grid, blk, 0, st, (uint16_t*)out,
                       (const uint16_t*)x, (const uint16_t*)weight, (uint16_t*)residual, dim, eps,
                       part, (int)n_splits, slab);
  else if (dtype == SLM_F16)
    hipLaunchKernelGGL(rms_norm_kernel<f16_tag>, grid, blk, 0, st, (uint16_t*)out,
                       (const uint16_t*)x, (const uint16_t*)weight, (uint16_t*)residual, dim, eps,
                       part, (int)n_splits, slab);
  else
    return SLM_ERR_UNSUPPORTED;
  return hip_check_launch();
}

SLM_API int slm_rms_norm(void* out, const void* x, const void* weight, void* residual,
                         int64_t n_tokens, int64_t dim, float eps, int32_t dtype, void* stream) {
  if (n_tokens != 0 && !x) return SLM_ERR_INVALID_ARG;
  return rms_norm_launch(out, x, nullptr, 0, weight, residual, n_tokens, dim, eps, dtype, stream);
}

SLM_API int slm_rms_norm_splitk(void* out, const float* partials, int32_t n_splits, const void* weight,
                                void* residual, int64_t n_tokens, int64_t dim, float eps,
                                int32_t dtype, void* stream) {
  if (n_tokens != 0 && !partials) return SLM_ERR_INVALID_ARG;
  return rms_norm_launch(out, nullptr, partials, n_splits, weight, residual, n_tokens, dim, eps, dtype,
                         stream);
}

SLM_API int slm_rope_kv_append(void* q, int64_t q_token_stride, void* k, int64_t k_token_stride,
                               const void* v, int64_t v_token_stride, const int32_t* positions,
                               const void* cos_sin, int32_t cos_sin_is_f32, int32_t rot_dim,
                               int32_t interleaved, const int32_t* slot_ids, void* key_cache,
                               void* value_cache, int64_t n_tokens, int32_t n_heads,
                               int32_t n_kv_heads, int32_t head_dim, int32_t dtype, void* stream) {
  if (n_tokens == 0) return SLM_OK;
  if (!k || n_tokens < 0 || n_kv_heads <= 0 || head_dim <= 0) return SLM_ERR_INVALID_ARG;
  if (cos_sin && (!q || !positions || rot_dim <= 0 || rot_dim % 2 || rot_dim > head_dim))
    return SLM_ERR_INVALID_ARG;
  if (slot_ids && (!v || !key_cache || !value_cache)) return SLM_ERR_INVALID_ARG;
  if (dtype != SLM_F16 && dtype != SLM_BF16) return SLM_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const dim3 grid((unsigned)n_tokens), blk(256);
#define SLM_ROPE(TT, CST)                                                                       \
  hipLaunchKernelGGL((rope_kv_append_kernel<TT, CST>), grid, blk, 0, st, (uint16_t*)q,           \
                     q_token_stride, (uint16_t*)k, k_token_stride, (const uint16_t*)v,          \
                     v_token_stride, positions, (const CST*)cos_sin, rot_dim, interleaved,      \
                     slot_ids, (uint16_t*)key_cache, (uint16_t*)value_cache, n_heads, n_kv_heads, \
                     head_dim)
  if (dtype == SLM_BF16) {
    if (cos_sin_is_f32) SLM_ROPE(bf16_tag, float); else SLM_ROPE(bf16_tag, uint16_t);
  } else {
    if (cos_sin_is_f32) SLM_ROPE(f16_tag, float); else SLM_ROPE(f16_tag, uint16_t);
  }
#undef SLM_ROPE
  return hip_check_launch();
}

SLM_API int slm_silu_mul(void* out, const void* x, int64_t n_tokens, int64_t d, int32_t dtype,
                         void* stream) {
  if (n_tokens == 0) return SLM_OK;
  if (!out || !x || n_tokens < 0) return SLM_ERR_INVALID_ARG;
  if (d <= 0 || d % 8) return SLM_ERR_UNSUPPORTED;
  if (!aligned16(out) || !aligned16(x)) return SLM_ERR_ALIGNMENT;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const int64_t total = n_tokens * (d / 8);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const dim3 grid((unsigned)blocks), blk(256);
  if (dtype == SLM_BF16)
    hipLaunchKernelGGL(silu_mul_kernel<bf16_tag>, grid, blk, 0, st, (uint16_t*)out,
                       (const uint16_t*)x, n_tokens, d);
  else if (dtype == SLM_F16)
    hipLaunchKernelGGL(silu_mul_kernel<f16_tag>, grid, blk, 0, st, (uint16_t*)out,
                       (const uint16_t*)x, n_tokens, d);
  else
    return SLM_ERR_UNSUPPORTED;
  return hip_check_launch();
}

}  // extern "C"
