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
//  slm_layer_norm      <- kernel::layer_norm (src/kernels/layernorm_kernels.cu:185-256; CPU path
//                         F::layer_norm, src/layers/normalization.h:54-61): GPT-2 / GPT-NeoX / Bloom / MPT.
//  slm_gelu            <- kernel::gelu_new / gelu_fast and their *_with_mul forms
//                         (src/kernels/activation_kernels.cu:20-40, 111-145; CPU path
//                         src/layers/activation.cpp:24-34, 57-65).
#include "common.h"

namespace slm {

__device__ __forceinline__ float wave_sum64(float v) { return group_sum<64>(v); }

// one workgroup (256 threads) per token; dim % 8 == 0; NV = vectors of 8 columns per thread
// (ceil(dim / 2048) rounded up to 1, 2, 4, 8).  Every load is unconditional (vector index clamped,
// contribution masked): a load inside `if (vi < nvec)` makes hipcc wait for it at the join, one
// memory round trip per unrolled iteration.
template <typename T, int NV>
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
  float v[NV][8];
  u32x4 wv[NV];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {  // the norm weights do not depend on anything: issue them first
    const int64_t vic = min((int64_t)tid + 256 * i, nvec - 1);
    wv[i] = *reinterpret_cast<const u32x4*>(weight + vic * 8);
  }
  u32x4 a[NV], r[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t vic = min((int64_t)tid + 256 * i, nvec - 1);
    if (part) {  // x = T(sum of the split-K partial slabs): what the reduce kernel would have left
      const f32x4 s0 = splitk_sum4(part + tok * dim + vic * 8, slab, n_splits);
      const f32x4 s1 = splitk_sum4(part + tok * dim + vic * 8 + 4, slab, n_splits);
      a[i].x = pack2<T>(s0.x, s0.y); a[i].y = pack2<T>(s0.z, s0.w);
      a[i].z = pack2<T>(s1.x, s1.y); a[i].w = pack2<T>(s1.z, s1.w);
    } else {
      a[i] = *reinterpret_cast<const u32x4*>(x + tok * dim + vic * 8);
    }
    r[i] = residual ? *reinterpret_cast<const u32x4*>(residual + tok * dim + vic * 8) : u32x4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t vi = tid + 256 * i;
    const bool valid = vi < nvec;
    float f[8] = {lo_f32<T>(a[i].x), hi_f32<T>(a[i].x), lo_f32<T>(a[i].y), hi_f32<T>(a[i].y),
                  lo_f32<T>(a[i].z), hi_f32<T>(a[i].z), lo_f32<T>(a[i].w), hi_f32<T>(a[i].w)};
    if (residual) {  // x = input + residual (fp32), residual = T(x): normalization.h:42-52
      f[0] += lo_f32<T>(r[i].x); f[1] += hi_f32<T>(r[i].x); f[2] += lo_f32<T>(r[i].y); f[3] += hi_f32<T>(r[i].y);
      f[4] += lo_f32<T>(r[i].z); f[5] += hi_f32<T>(r[i].z); f[6] += lo_f32<T>(r[i].w); f[7] += hi_f32<T>(r[i].w);
      u32x4 w;
      w.x = pack2<T>(f[0], f[1]); w.y = pack2<T>(f[2], f[3]);
      w.z = pack2<T>(f[4], f[5]); w.w = pack2<T>(f[6], f[7]);
      if (valid) *reinterpret_cast<u32x4*>(residual + tok * dim + vi * 8) = w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[i][j] = f[j];
    if (valid) ss = rms_sumsq8(f, ss);
  }
  ss = wave_sum64(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float tot = red[0] + red[1] + red[2] + red[3];
  const float rs = rsqrtf(tot / (float)dim + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t vi = tid + 256 * i;
    if (vi < nvec) {
      const u32x4 o = rms_apply8<T>(v[i], rs, wv[i]);
      *reinterpret_cast<u32x4*>(out + tok * dim + vi * 8) = o;
    }
  }
}

// (a, b) rotated by (cos, sin): ONE spelling for every RoPE kernel (explicit fma, so the compiler's
// contraction choices cannot differ between them -- the split-K form must reproduce the plain one)
__device__ __forceinline__ void rope_rot(const float a, const float b, const float c, const float s,
                                         float& o0, float& o1) {
  o0 = __builtin_fmaf(a, c, -(b * s));
  o1 = __builtin_fmaf(b, c, a * s);
  // fp32 results in registers before any conversion to T (otherwise fp16 callers may get a
  // single-rounding v_fma_mixlo_f16 in one kernel and fma + convert in another)
  asm("" : "+v"(o0));
  asm("" : "+v"(o1));
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
      float r0, r1;
      rope_rot(a, b, c, s, r0, r1);
      p[i0] = pack1<T>(r0);
      p[i1] = pack1<T>(r1);
    }
    // K: in place + cache slot
    uint16_t* kt = k + tok * k_ts;
    for (int i = tid; i < n_kv_heads * half; i += 256) {
      const int h = i / half, r = i % half;
      const int i0 = interleaved ? 2 * r : r, i1 = interleaved ? 2 * r + 1 : r + half;
      uint16_t* p = kt + (int64_t)h * head_dim;
      const float a = lo_f32<T>((uint32_t)p[i0]), b = lo_f32<T>((uint32_t)p[i1]);
      const float c = csval(r), s = csval(half + r);
      float r0, r1;
      rope_rot(a, b, c, s, r0, r1);
      const uint16_t o0 = pack1<T>(r0), o1 = pack1<T>(r1);
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

// The same operator with q / k / v given as the fp32 split-K partial sums the fused qkv GEMM left
// behind (SLM_W4_DEFER_REDUCE): row t of the GEMM output is [q (n_heads D) | k (n_kv D) | v (n_kv D)]
// and x = T(sum_s part[s][t][col]), summed in the split-K reduce kernel's order and rounded to T at
// the same point, so the result is bit-identical to "reduce, then slm_rope_kv_append" -- minus the
// reduce launch and one round trip of the qkv activations.  q is written rotated to `q` (attention
// reads it there), k / v to `k` / `v` and to their cache slot.  A thread owns 4 + 4 values of one
// head: dims [4u, 4u+4) and [half + 4u, ...) (non-interleaved pairs (i, i + half)) or the 8
// consecutive dims [8u, 8u+8) (interleaved pairs (2i, 2i+1)): two 16-B loads per slab either way.
// PART = false: the same thread -> unit map over q / k / v as they are (slm_rope_kv_append with
// 4-aligned layouts): in place, 8-byte loads, one round trip per thread where the scalar kernel
// above makes one per rotation pair (Llama-3-8B, one token: 11 dependent trips, 5.3 us).
template <typename T, typename CS, bool PART>
__global__ void __launch_bounds__(256) rope_kv_append_splitk_kernel(
    const float* __restrict__ part, int n_splits, int64_t slab /* = n_tokens * N */, int64_t N,
    uint16_t* __restrict__ q, int64_t q_ts, uint16_t* __restrict__ k, int64_t k_ts,
    uint16_t* __restrict__ v, int64_t v_ts, const int* __restrict__ positions,
    const CS* __restrict__ cos_sin, int rot_dim, int interleaved, const int* __restrict__ slot_ids,
    uint16_t* __restrict__ key_cache, uint16_t* __restrict__ value_cache, int n_heads,
    int n_kv_heads, int head_dim) {
  const int64_t tok = blockIdx.x;
  const int tid = threadIdx.x;
  const int half = rot_dim / 2;
  const int64_t slot = slot_ids ? (int64_t)slot_ids[tok] : -1;
  const int64_t row = (int64_t)n_kv_heads * head_dim;
  const CS* cs = cos_sin + (int64_t)positions[tok] * rot_dim;
  auto csval = [&](int i) -> float {
    if constexpr (sizeof(CS) == 4) return (float)cs[i];
    else return lo_f32<T>((uint32_t)cs[i]);
  };
  const float* prow = part + tok * N;
  const int64_t q_cols = (int64_t)n_heads * head_dim;
  // T(sum over slabs) of 4 consecutive columns, as fp32 values that are exact in T
  // (column numbering of the fused GEMM row: [q | k | v])
  auto ld4 = [&](int64_t col, float (&x)[4]) {
    if constexpr (PART) {
      const f32x4 s = splitk_sum4(prow + col, slab, n_splits);
      x[0] = lo_f32<T>((uint32_t)pack1<T>(s.x)); x[1] = lo_f32<T>((uint32_t)pack1<T>(s.y));
      x[2] = lo_f32<T>((uint32_t)pack1<T>(s.z)); x[3] = lo_f32<T>((uint32_t)pack1<T>(s.w));
    } else {
      const uint16_t* src = col < q_cols ? q + tok * q_ts + col
                            : col < q_cols + row ? k + tok * k_ts + (col - q_cols)
                                                 : v + tok * v_ts + (col - q_cols - row);
      const u32x2 w = *reinterpret_cast<const u32x2*>(src);
      x[0] = lo_f32<T>(w.x); x[1] = hi_f32<T>(w.x); x[2] = lo_f32<T>(w.y); x[3] = hi_f32<T>(w.y);
    }
  };
  auto st4 = [&](uint16_t* dst, const float (&x)[4]) {
    u32x2 w;
    w.x = pack2<T>(x[0], x[1]);
    w.y = pack2<T>(x[2], x[3]);
    *reinterpret_cast<u32x2*>(dst) = w;
  };
  const int upr = half / 4;                  // rotary units per head
  const int n_rot_heads = n_heads + n_kv_heads;  // q heads, then k heads
  // gridDim.y workgroups share a token: one 4 + 4-value unit per thread and pass, so the partial
  // loads of a thread are one round trip, not one per loop iteration
  const int tstep = 256 * gridDim.y, t0 = tid + 256 * blockIdx.y;
  for (int i = t0; i < n_rot_heads * upr; i += tstep) {
    const int h = i / upr, u = i % upr;
    const bool is_k = h >= n_heads;
    const int hh = is_k ? h - n_heads : h;
    const int64_t col0 = (int64_t)h * head_dim;  // k heads follow the q heads in the GEMM row
    const int da = interleaved ? 8 * u : 4 * u, db = interleaved ? 8 * u + 4 : half + 4 * u;
    float a[4], b[4], oa[4], ob[4];
    ld4(col0 + da, a);
    ld4(col0 + db, b);
    if (interleaved) {  // pairs (a0,a1) (a2,a3) (b0,b1) (b2,b3), pair index r = 4u + {0,1,2,3}
      const float c0 = csval(4 * u), s0 = csval(half + 4 * u), c1 = csval(4 * u + 1), s1 = csval(half + 4 * u + 1);
      const float c2 = csval(4 * u + 2), s2 = csval(half + 4 * u + 2), c3 = csval(4 * u + 3), s3 = csval(half + 4 * u + 3);
      rope_rot(a[0], a[1], c0, s0, oa[0], oa[1]);
      rope_rot(a[2], a[3], c1, s1, oa[2], oa[3]);
      rope_rot(b[0], b[1], c2, s2, ob[0], ob[1]);
      rope_rot(b[2], b[3], c3, s3, ob[2], ob[3]);
    } else {            // pairs (a[e], b[e]), r = 4u + e
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        rope_rot(a[e], b[e], csval(4 * u + e), csval(half + 4 * u + e), oa[e], ob[e]);
      }
    }
    uint16_t* dst = is_k ? k + tok * k_ts + (int64_t)hh * head_dim : q + tok * q_ts + (int64_t)hh * head_dim;
    st4(dst + da, oa);
    st4(dst + db, ob);
    if (is_k && slot >= 0) {
      uint16_t* kc = key_cache + slot * row + (int64_t)hh * head_dim;
      st4(kc + da, oa);
      st4(kc + db, ob);
    }
  }
  // pass-through dims of q and k (rot_dim < head_dim), then v: plain T(sum) copies, 4 columns each
  const int pass4 = (head_dim - rot_dim) / 4;
  for (int i = t0; i < n_rot_heads * pass4; i += tstep) {
    const int h = i / pass4, d = rot_dim + 4 * (i % pass4);
    const bool is_k = h >= n_heads;
    const int hh = is_k ? h - n_heads : h;
    if (!PART && !(is_k && slot >= 0)) continue;  // in place: only the cache copy is left to do
    float x[4];
    ld4((int64_t)h * head_dim + d, x);
    if constexpr (PART) st4((is_k ? k + tok * k_ts : q + tok * q_ts) + (int64_t)hh * head_dim + d, x);
    if (is_k && slot >= 0) st4(key_cache + slot * row + (int64_t)hh * head_dim + d, x);
  }
  if (!PART && slot < 0) return;
  const int64_t v_col0 = (int64_t)n_rot_heads * head_dim;
  for (int i = t0; i < row / 4; i += tstep) {
    float x[4];
    ld4(v_col0 + 4 * i, x);
    if constexpr (PART) st4(v + tok * v_ts + 4 * i, x);
    if (slot >= 0) st4(value_cache + slot * row + 4 * i, x);
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

// LayerNorm, one workgroup per token: the row stays in registers between the three passes (mean,
// centred variance, normalise) -- the reference kernel re-reads it from memory for each
// (layernorm_kernels.cu:198-226).  out = T((x - mean) * rsqrt(var + eps) * w + b), all in fp32, ONE
// rounding at the end (unlike RMSNorm, which rounds before the weight: layernorm_kernels.cu:222-227).
template <typename T, int NV>
__global__ void __launch_bounds__(256) layer_norm_kernel(uint16_t* __restrict__ out,
                                                         const uint16_t* __restrict__ x,
                                                         const uint16_t* __restrict__ weight,
                                                         const uint16_t* __restrict__ bias, int64_t dim,
                                                         float eps) {
  __shared__ float red[2][4];
  const int64_t tok = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t nvec = dim / 8;
  u32x4 a[NV], wv[NV], bv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {  // every load unconditional (index clamped, contribution masked)
    const int64_t vic = min((int64_t)tid + 256 * i, nvec - 1);
    wv[i] = *reinterpret_cast<const u32x4*>(weight + vic * 8);
    bv[i] = bias ? *reinterpret_cast<const u32x4*>(bias + vic * 8) : u32x4{0u, 0u, 0u, 0u};
    a[i] = *reinterpret_cast<const u32x4*>(x + tok * dim + vic * 8);
  }
  float v[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bool valid = (int64_t)tid + 256 * i < nvec;
    const float f[8] = {lo_f32<T>(a[i].x), hi_f32<T>(a[i].x), lo_f32<T>(a[i].y), hi_f32<T>(a[i].y),
                        lo_f32<T>(a[i].z), hi_f32<T>(a[i].z), lo_f32<T>(a[i].w), hi_f32<T>(a[i].w)};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[i][j] = f[j];
      if (valid) sum += f[j];
    }
  }
  sum = wave_sum64(sum);
  if ((tid & 63) == 0) red[0][tid >> 6] = sum;
  __syncthreads();
  const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)dim;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bool valid = (int64_t)tid + 256 * i < nvec;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[i][j] -= mean;
      if (valid) var = __builtin_fmaf(v[i][j], v[i][j], var);
    }
  }
  var = wave_sum64(var);
  if ((tid & 63) == 0) red[1][tid >> 6] = var;
  __syncthreads();
  const float rs = rsqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)dim + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t vi = tid + 256 * i;
    if (vi < nvec) {
      const float w[8] = {lo_f32<T>(wv[i].x), hi_f32<T>(wv[i].x), lo_f32<T>(wv[i].y), hi_f32<T>(wv[i].y),
                          lo_f32<T>(wv[i].z), hi_f32<T>(wv[i].z), lo_f32<T>(wv[i].w), hi_f32<T>(wv[i].w)};
      const float b[8] = {lo_f32<T>(bv[i].x), hi_f32<T>(bv[i].x), lo_f32<T>(bv[i].y), hi_f32<T>(bv[i].y),
                          lo_f32<T>(bv[i].z), hi_f32<T>(bv[i].z), lo_f32<T>(bv[i].w), hi_f32<T>(bv[i].w)};
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = v[i][j] * rs * w[j] + b[j];
      u32x4 r;
      r.x = pack2<T>(o[0], o[1]); r.y = pack2<T>(o[2], o[3]);
      r.z = pack2<T>(o[4], o[5]); r.w = pack2<T>(o[6], o[7]);
      *reinterpret_cast<u32x4*>(out + tok * dim + vi * 8) = r;
    }
  }
}

// tanh-form GELU (activation_kernels.cu:20-40): x * 0.5 (1 + tanh(u)), u = 0.79788456 (x + 0.044715 x^3)
// ("new", GPT-2) or 0.79788456 x (1 + 0.044715 x^2) ("fast") -- the same polynomial, two roundings apart.
// 0.5 (1 + tanh(u)) = 1 / (1 + 2^(-2 u log2 e)): one exp2 + one rcp, saturating correctly at both ends
// (the reference uses tanh.approx.f32, abs error ~5e-4; this form is good to ~1e-7).
template <bool FAST>
__device__ __forceinline__ float gelu_tanh1(const float x) {
  const float u = FAST ? (0.7978845608028654f * x) * (1.0f + 0.044715f * x * x)
                       : 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + fast_exp2(u * (-2.0f * 1.4426950408889634f)));
}

// out[t, i] = gelu(x[t, i])                    (MUL = false; x row stride d)
// out[t, i] = gelu(x[t, i]) * x[t, d + i]      (MUL = true;  x row stride 2 d: activation.cpp:57-65)
template <typename T, bool FAST, bool MUL>
__global__ void __launch_bounds__(256) gelu_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ x,
                                                   int64_t n_tokens, int64_t d) {
  const int64_t nvec = d / 8;
  const int64_t total = n_tokens * nvec;
  const int64_t xs = MUL ? 2 * d : d;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t t = idx / nvec, vi = idx % nvec;
    const u32x4 g = *reinterpret_cast<const u32x4*>(x + t * xs + vi * 8);
    u32x4 u = {0u, 0u, 0u, 0u};
    if constexpr (MUL) u = *reinterpret_cast<const u32x4*>(x + t * xs + d + vi * 8);
    const float gf[8] = {lo_f32<T>(g.x), hi_f32<T>(g.x), lo_f32<T>(g.y), hi_f32<T>(g.y),
                         lo_f32<T>(g.z), hi_f32<T>(g.z), lo_f32<T>(g.w), hi_f32<T>(g.w)};
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = gelu_tanh1<FAST>(gf[j]);
    if constexpr (MUL) {
      // the reference rounds the activation to T before the multiply (activation_kernels.cu:62-80: the
      // functor returns T): same here
      const float uf[8] = {lo_f32<T>(u.x), hi_f32<T>(u.x), lo_f32<T>(u.y), hi_f32<T>(u.y),
                           lo_f32<T>(u.z), hi_f32<T>(u.z), lo_f32<T>(u.w), hi_f32<T>(u.w)};
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = lo_f32<T>((uint32_t)pack1<T>(o[j])) * uf[j];
    }
    u32x4 r;
    r.x = pack2<T>(o[0], o[1]); r.y = pack2<T>(o[2], o[3]);
    r.z = pack2<T>(o[4], o[5]); r.w = pack2<T>(o[6], o[7]);
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
  const int64_t per_thread = (dim / 8 + 255) / 256;
#define SLM_NORM(TT, NVV)                                                                       \
  hipLaunchKernelGGL((rms_norm_kernel<TT, NVV>), grid, blk, 0, st, (uint16_t*)out,              \
                     (const uint16_t*)x, (const uint16_t*)weight, (uint16_t*)residual, dim, eps, \
                     part, (int)n_splits, slab)
#define SLM_NORM_NV(TT)                                                                         \
  do {                                                                                          \
    if (per_thread <= 1) SLM_NORM(TT, 1); else if (per_thread <= 2) SLM_NORM(TT, 2);            \
    else if (per_thread <= 4) SLM_NORM(TT, 4); else SLM_NORM(TT, 8);                            \
  } while (0)
  if (dtype == SLM_BF16) SLM_NORM_NV(bf16_tag);
  else if (dtype == SLM_F16) SLM_NORM_NV(f16_tag);
  else return SLM_ERR_UNSUPPORTED;
#undef SLM_NORM_NV
#undef SLM_NORM
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
  // 4-aligned layouts (every model on the path): the vector kernel, one round trip per thread
  const bool vec = cos_sin && rot_dim % 8 == 0 && head_dim % 4 == 0 && q_token_stride % 4 == 0 &&
                   k_token_stride % 4 == 0 && (!slot_ids || v_token_stride % 4 == 0) &&
                   !(reinterpret_cast<uintptr_t>(q) & 7) && !(reinterpret_cast<uintptr_t>(k) & 7) &&
                   !(reinterpret_cast<uintptr_t>(v) & 7) && !(reinterpret_cast<uintptr_t>(key_cache) & 7) &&
                   !(reinterpret_cast<uintptr_t>(value_cache) & 7);
  if (vec) {
    const int64_t units = ((int64_t)(n_heads + n_kv_heads) * (rot_dim / 8) + (int64_t)n_kv_heads * head_dim / 4);
    const unsigned gy = units > 768 ? 4u : units > 256 ? 2u : 1u;
    const dim3 vgrid((unsigned)n_tokens, gy), vblk(256);
#define SLM_ROPE_V(TT, CST)                                                                     \
  hipLaunchKernelGGL((rope_kv_append_splitk_kernel<TT, CST, false>), vgrid, vblk, 0, st,        \
                     (const float*)nullptr, 0, (int64_t)0, (int64_t)0, (uint16_t*)q,            \
                     q_token_stride, (uint16_t*)k, k_token_stride, (uint16_t*)const_cast<void*>(v), \
                     v_token_stride, positions, (const CST*)cos_sin, rot_dim, interleaved,      \
                     slot_ids, (uint16_t*)key_cache, (uint16_t*)value_cache, n_heads, n_kv_heads, \
                     head_dim)
    if (dtype == SLM_BF16) {
      if (cos_sin_is_f32) SLM_ROPE_V(bf16_tag, float); else SLM_ROPE_V(bf16_tag, uint16_t);
    } else {
      if (cos_sin_is_f32) SLM_ROPE_V(f16_tag, float); else SLM_ROPE_V(f16_tag, uint16_t);
    }
#undef SLM_ROPE_V
    return hip_check_launch();
  }
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

SLM_API int slm_rope_kv_append_splitk(const float* partials, int32_t n_splits, void* q,
                                      int64_t q_token_stride, void* k, int64_t k_token_stride,
                                      void* v, int64_t v_token_stride, const int32_t* positions,
                                      const void* cos_sin, int32_t cos_sin_is_f32, int32_t rot_dim,
                                      int32_t interleaved, const int32_t* slot_ids, void* key_cache,
                                      void* value_cache, int64_t n_tokens, int32_t n_heads,
                                      int32_t n_kv_heads, int32_t head_dim, int32_t dtype,
                                      void* stream) {
  if (n_tokens == 0) return SLM_OK;
  if (!partials || n_splits < 1 || !q || !k || !v || !positions || !cos_sin || n_tokens < 0 ||
      n_heads <= 0 || n_kv_heads <= 0 || head_dim <= 0)
    return SLM_ERR_INVALID_ARG;
  if (rot_dim <= 0 || rot_dim > head_dim) return SLM_ERR_INVALID_ARG;
  if (slot_ids && (!key_cache || !value_cache)) return SLM_ERR_INVALID_ARG;
  if (dtype != SLM_F16 && dtype != SLM_BF16) return SLM_ERR_UNSUPPORTED;
  // 16-byte partial loads / 8-byte stores: 4-column units everywhere
  if (rot_dim % 8 || head_dim % 4 || q_token_stride % 4 || k_token_stride % 4 || v_token_stride % 4)
    return SLM_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(partials) & 15) || (reinterpret_cast<uintptr_t>(q) & 7) ||
      (reinterpret_cast<uintptr_t>(k) & 7) || (reinterpret_cast<uintptr_t>(v) & 7) ||
      (reinterpret_cast<uintptr_t>(key_cache) & 7) || (reinterpret_cast<uintptr_t>(value_cache) & 7))
    return SLM_ERR_ALIGNMENT;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const int64_t N = (int64_t)(n_heads + 2 * n_kv_heads) * head_dim;
  const int64_t units = ((int64_t)(n_heads + n_kv_heads) * (rot_dim / 8) + (int64_t)n_kv_heads * head_dim / 4);
  const unsigned gy = units > 768 ? 4u : units > 256 ? 2u : 1u;
  const dim3 grid((unsigned)n_tokens, gy), blk(256);
#define SLM_ROPE_SK(TT, CST)                                                                     \
  hipLaunchKernelGGL((rope_kv_append_splitk_kernel<TT, CST, true>), grid, blk, 0, st, partials, n_splits, \
                     n_tokens * N, N, (uint16_t*)q, q_token_stride, (uint16_t*)k, k_token_stride, \
                     (uint16_t*)v, v_token_stride, positions, (const CST*)cos_sin, rot_dim,       \
                     interleaved, slot_ids, (uint16_t*)key_cache, (uint16_t*)value_cache, n_heads, \
                     n_kv_heads, head_dim)
  if (dtype == SLM_BF16) {
    if (cos_sin_is_f32) SLM_ROPE_SK(bf16_tag, float); else SLM_ROPE_SK(bf16_tag, uint16_t);
  } else {
    if (cos_sin_is_f32) SLM_ROPE_SK(f16_tag, float); else SLM_ROPE_SK(f16_tag, uint16_t);
  }
#undef SLM_ROPE_SK
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

SLM_API int slm_layer_norm(void* out, const void* x, const void* weight, const void* bias, int64_t n_tokens,
                           int64_t dim, float eps, int32_t dtype, void* stream) {
  if (n_tokens == 0) return SLM_OK;
  if (!out || !x || !weight || n_tokens < 0) return SLM_ERR_INVALID_ARG;
  if (dim <= 0 || dim % 8 || dim > 16384) return SLM_ERR_UNSUPPORTED;
  if (!aligned16(out) || !aligned16(x) || !aligned16(weight) || (bias && !aligned16(bias))) return SLM_ERR_ALIGNMENT;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const dim3 grid((unsigned)n_tokens), blk(256);
  const int64_t per_thread = (dim / 8 + 255) / 256;
#define SLM_LN(TT, NVV)                                                                             \
  hipLaunchKernelGGL((layer_norm_kernel<TT, NVV>), grid, blk, 0, st, (uint16_t*)out, (const uint16_t*)x, \
                     (const uint16_t*)weight, (const uint16_t*)bias, dim, eps)
#define SLM_LN_NV(TT)                                                                               \
  do {                                                                                              \
    if (per_thread <= 1) SLM_LN(TT, 1); else if (per_thread <= 2) SLM_LN(TT, 2);                    \
    else if (per_thread <= 4) SLM_LN(TT, 4); else SLM_LN(TT, 8);                                    \
  } while (0)
  if (dtype == SLM_BF16) SLM_LN_NV(bf16_tag);
  else if (dtype == SLM_F16) SLM_LN_NV(f16_tag);
  else return SLM_ERR_UNSUPPORTED;
#undef SLM_LN_NV
#undef SLM_LN
  return hip_check_launch();
}

SLM_API int slm_gelu(void* out, const void* x, int64_t n_tokens, int64_t d, int32_t kind, int32_t with_mul,
                     int32_t dtype, void* stream) {
  if (n_tokens == 0) return SLM_OK;
  if (!out || !x || n_tokens < 0) return SLM_ERR_INVALID_ARG;
  if (kind != SLM_GELU_NEW && kind != SLM_GELU_FAST) return SLM_ERR_INVALID_ARG;
  if (d <= 0 || d % 8) return SLM_ERR_UNSUPPORTED;
  if (!aligned16(out) || !aligned16(x)) return SLM_ERR_ALIGNMENT;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const int64_t total = n_tokens * (d / 8);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const dim3 grid((unsigned)blocks), blk(256);
#define SLM_GELU(TT, FF, MM)                                                                        \
  hipLaunchKernelGGL((gelu_kernel<TT, FF, MM>), grid, blk, 0, st, (uint16_t*)out, (const uint16_t*)x, n_tokens, d)
#define SLM_GELU_T(TT)                                                                              \
  do {                                                                                              \
    if (kind == SLM_GELU_FAST) { if (with_mul) SLM_GELU(TT, true, true); else SLM_GELU(TT, true, false); } \
    else { if (with_mul) SLM_GELU(TT, false, true); else SLM_GELU(TT, false, false); }              \
  } while (0)
  if (dtype == SLM_BF16) SLM_GELU_T(bf16_tag);
  else if (dtype == SLM_F16) SLM_GELU_T(f16_tag);
  else return SLM_ERR_UNSUPPORTED;
#undef SLM_GELU_T
#undef SLM_GELU
  return hip_check_launch();
}

}  // extern "C"
