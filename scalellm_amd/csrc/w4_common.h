// w4_common.h -- shared pieces of the int4-weight GEMM kernels (w4.hip, w4_ws.hip): the dequant
// helpers (one code path for the GEMMs and the debug dequant kernel), MFMA wrappers, kernel params.
#pragma once
#include "common.h"

namespace slm {

// ------------------------------------------------------------------------------------------
// unpack helpers (shared by the GEMM and the debug dequant kernel: one code path to test)
// ------------------------------------------------------------------------------------------
template <typename T>
struct W4Dq;

template <>
struct W4Dq<bf16_tag> {
  // bf16 has no packed VALU arithmetic on gfx950: the affine map runs in fp32 on the NIBBLE VALUES
  // (q as a float straight out of v_cvt_f32_ubyteN, no shifts), fma(q, s, -z s) = (q - z) s exactly
  // (<= 5 + 8 significant bits), one rounding in v_cvt_pk_bf16_f32.  Per 8-weight word: 3 mask /
  // shift + 8 cvt + 4 v_pk_fma_f32 + 4 cvt_pk = 19 VALU (the magic-number route through fp32 took 27
  // and produced the same bits).
  //
  // Round 5: 15 VALU per word, same bits.  A nibble in a byte, 0x0q, IS the OCP E4M3 encoding of
  // q * 2^-9 (codes 0..7 are the subnormals m * 2^-9, codes 8..15 the first binade (8 + m) * 2^-9: one
  // linear ramp), so ONE v_cvt_pk_f32_fp8 turns two nibbles into two floats where v_cvt_f32_ubyteN
  // took one instruction each: 3 mask / shift + 4 cvt_pk_f32_fp8 + 4 v_pk_fma_f32 + 4 cvt_pk.  The
  // 2^-9 is folded into the scale (s * 512: a power of two, exact), so fma(q 2^-9, 512 s, -z s) is
  // the same exact (q - z) s as before and the one rounding happens in v_cvt_pk_bf16_f32.
  // (fp8 pairs come out as (e0, e4) / (e1, e5) / (e2, e6) / (e3, e7): v_cvt_pk_bf16_f32 takes its two
  // sources from different registers, so the natural element order costs nothing.)
  f32x2 s2, c2;    // s, -z s          (pair(): the byte route)
  f32x2 s512;      // 512 s            (word(): the fp8 route)
  struct Nib { uint32_t lo, hi; };  // lo: bytes (e0, e4, e1, e5); hi: bytes (e2, e6, e3, e7)
  __device__ __forceinline__ explicit W4Dq(uint32_t sz) {
    const float s = __builtin_bit_cast(float, sz << 16);
    const float zm = __builtin_bit_cast(float, sz & 0xffff0000u);  // 128 + zero, exact
    const float c = -(zm - 128.0f) * s;                            // -zero * s: exact
    s2 = f32x2{s, s};
    c2 = f32x2{c, c};
    const float sb = s * 512.0f;
    s512 = f32x2{sb, sb};
  }
  __device__ __forceinline__ Nib split(uint32_t w) const {
    Nib n;
    n.lo = w & 0x0F0F0F0Fu;
    n.hi = (w >> 4) & 0x0F0F0F0Fu;
    // opaque: otherwise hipcc re-derives every nibble from w with its own shift + and
    asm("" : "+v"(n.lo));
    asm("" : "+v"(n.hi));
    return n;
  }
  // nibble pair i (elements 2i, 2i+1) -> packed bf16 pair
  __device__ __forceinline__ uint32_t pair(const Nib& n, int i) const {
    const uint32_t src = (i & 1) ? n.hi : n.lo;
    f32x2 q;
    if (i < 2) q = f32x2{(float)(src & 0xffu), (float)((src >> 16) & 0xffu)};
    else q = f32x2{(float)((src >> 8) & 0xffu), (float)(src >> 24)};
    const f32x2 r = __builtin_elementwise_fma(q, s2, c2);
    return pack2<bf16_tag>(r[0], r[1]);
  }
  // 8 nibbles -> 4 packed bf16 pairs, element order e = 0..7  (the fp8 route; scales >= 2^119 would
  // overflow in 512 s -- no checkpoint has them, the byte route in pair() has no such bound)
  __device__ __forceinline__ void word(uint32_t w, uint32_t (&out)[4]) const {
    const Nib n = split(w);
    const f32x2 q04 = __builtin_amdgcn_cvt_pk_f32_fp8((int)n.lo, false);  // (e0, e4) * 2^-9
    const f32x2 q15 = __builtin_amdgcn_cvt_pk_f32_fp8((int)n.lo, true);   // (e1, e5)
    const f32x2 q26 = __builtin_amdgcn_cvt_pk_f32_fp8((int)n.hi, false);  // (e2, e6)
    const f32x2 q37 = __builtin_amdgcn_cvt_pk_f32_fp8((int)n.hi, true);   // (e3, e7)
    const f32x2 r04 = __builtin_elementwise_fma(q04, s512, c2);
    const f32x2 r15 = __builtin_elementwise_fma(q15, s512, c2);
    const f32x2 r26 = __builtin_elementwise_fma(q26, s512, c2);
    const f32x2 r37 = __builtin_elementwise_fma(q37, s512, c2);
    out[0] = pack2<bf16_tag>(r04[0], r15[0]);
    out[1] = pack2<bf16_tag>(r26[0], r37[0]);
    out[2] = pack2<bf16_tag>(r04[1], r15[1]);
    out[3] = pack2<bf16_tag>(r26[1], r37[1]);
  }
};

template <>
struct W4Dq<f16_tag> {
  f16x2_t s2, nzm2;
  __device__ __forceinline__ explicit W4Dq(uint32_t sz) {
    const f16x2_t v = __builtin_bit_cast(f16x2_t, sz);
    s2 = f16x2_t{v[0], v[0]};
    nzm2 = f16x2_t{-v[1], -v[1]};  // -(1024 + zero)
  }
  struct Nib { uint32_t w; };
  __device__ __forceinline__ Nib split(uint32_t w) const { return Nib{w}; }
  __device__ __forceinline__ uint32_t pair(const Nib& n, int i) const {
    const uint32_t w = n.w;
    const uint32_t t = ((w >> (4 * i)) & 0x000F000Fu) | 0x64006400u;  // (1024+q_lo, 1024+q_hi)
    const f16x2_t d = __builtin_bit_cast(f16x2_t, t) + nzm2;          // q - z, exact
    return __builtin_bit_cast(uint32_t, d * s2);                      // RN
  }
  __device__ __forceinline__ void word(uint32_t w, uint32_t (&out)[4]) const {
    const Nib n = split(w);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = pair(n, i);
  }
};

// The magic-number route through fp32 (128 + q as a bf16 bit pattern widened to fp32, then
// fma(128 + q, s, -(128 + zero) s): 27 VALU per word, the same bits as W4Dq<bf16_tag>).  Kept for the
// wave-specialised kernel (w4_ws.hip): its producers interleave the dequant with LDS-DMA issue, and
// the short form measured 7 % SLOWER there (gate_up M = 256: 67.5 -> 72.2 us, A/B on one box).
template <typename T>
struct W4DqMagic : W4Dq<T> {
  __device__ __forceinline__ explicit W4DqMagic(uint32_t sz) : W4Dq<T>(sz) {}
  __device__ __forceinline__ uint32_t pair(uint32_t w, int i) const {
    return W4Dq<T>::pair(W4Dq<T>::split(w), i);
  }
};
template <>
struct W4DqMagic<bf16_tag> {
  float s, c;
  __device__ __forceinline__ explicit W4DqMagic(uint32_t sz) {
    s = __builtin_bit_cast(float, sz << 16);
    const float zm = __builtin_bit_cast(float, sz & 0xffff0000u);  // 128 + zero, exact
    c = -zm * s;                                                    // <= 16 significant bits: exact
  }
  // nibble pair i (elements 2i, 2i+1) -> packed bf16 pair
  __device__ __forceinline__ uint32_t pair(uint32_t w, int i) const {
    const uint32_t t = ((w >> (4 * i)) & 0x000F000Fu) | 0x43004300u;  // (128+q_lo, 128+q_hi)
    const float lo = __builtin_bit_cast(float, t << 16);
    const float hi = __builtin_bit_cast(float, t & 0xffff0000u);
    return pack2<bf16_tag>(fmaf(lo, s, c), fmaf(hi, s, c));  // (q - z) * s exact, then RN
  }
  // 8 nibbles -> 4 packed bf16 pairs, element order e = 0..7
  __device__ __forceinline__ void word(uint32_t w, uint32_t (&out)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = pair(w, i);
  }
};


// POST-scaled form (small-M kernels): the MFMA consumes the raw magic-number values (magic + q,
// exact in T) -- unpack is 7 VALU per 8 weights instead of ~27 -- and the affine part is applied to
// the per-group partial sums:  sum_k x_k s (q_k - z) = s * ( sum_k x_k (magic+q_k) - (magic+z) sum_k x_k ).
template <typename T>
struct W4Magic;
template <>
struct W4Magic<bf16_tag> {
  static constexpr uint32_t bits = 0x43004300u;
  static __device__ __forceinline__ void decode(uint32_t sz, float& s, float& zm) {
    s = __builtin_bit_cast(float, sz << 16);
    zm = __builtin_bit_cast(float, sz & 0xffff0000u);
  }
};
template <>
struct W4Magic<f16_tag> {
  static constexpr uint32_t bits = 0x64006400u;
  static __device__ __forceinline__ void decode(uint32_t sz, float& s, float& zm) {
    const f16x2_t v = __builtin_bit_cast(f16x2_t, sz);
    s = (float)v[0];
    zm = (float)v[1];
  }
};

struct GemmKParams {
  const void* a;
  const uint32_t* wq;
  const uint32_t* sz;
  const void* bias;
  void* c;
  float* part;  // [split_k][M][N] fp32 (split_k > 1)
  int64_t M, K, N, lda, ldc;
  int gs_shift;      // log2(group_size) (group_size >= 128 handled via k >> gs_shift too)
  int n_chunks;      // K / 128
  int split_k;
  int chunks_per_split;
  int n_mblocks, n_nblocks;
  int silu;  // SLM_W4_SILU_MUL: column tiles are (gate, up) pairs, c is [M, N/2]
  int ks_tpw;     // w4_ks.hip: consecutive column tiles per workgroup (n_nblocks = tile runs)
  int ks_groups;  // w4_ks.hip: K / group_size (rows of the scale table)
  int ks_dbg;     // w4_ks.hip: probe bits (SLM_W4_KS_DBG), 0 in production
  // w4_xl.hip, stream-K form (round 6): workgroup g (a ticket drawn at start) owns the 128-deep chunks
  // [g * sk_per, (g + 1) * sk_per) of the tile-major work list (tiles x n_chunks); sk_sync = {ticket, flag[g] ...},
  // zero at launch; sk_part = one 256 x 256 fp32 tile image per workgroup (fragment-major)
  int sk_per;
  unsigned* sk_sync;
  float* sk_part;
  // GEMV norm prologue (slm_w4a16_gemv_norm): activations = rms_norm(x + residual_in) * weight,
  // computed in the kernel; norm_weight == NULL: none, `a` is read as usual
  const void* norm_x;          // [M, K] T or NULL
  const float* norm_part;      // [norm_splits, M, K] fp32 or NULL
  int norm_splits;
  float norm_eps;
  const void* norm_res_in;     // [M, K] T or NULL
  void* norm_res_out;          // [M, K] T
  const void* norm_weight;     // [K] T
  void* norm_out;              // optional [M, K] T copy of the normalised activations
};

template <typename T>
struct Mfma;
template <>
struct Mfma<bf16_tag> {
  typedef bf16x8_t frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mfma<f16_tag> {
  typedef f16x8_t frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// SLM_W4_SILU_MUL epilogue of the C^T-accumulator kernels (w4_ws.hip, w4_xl.hip; split_k == 1):
// the wave's two adjacent column tiles t0 (even: gate) and t0 + 1 (up) sit in the same lane at the
// same (i, r), so the pair never leaves its registers.  Lane = token row0 + 32 i; the 16 values
// are the columns (r & 3) + 8 (r >> 2) + 4 (lane >> 5): four consecutive outputs per r >> 2.
template <typename T>
__device__ __forceinline__ void store_ct_silu_pair(const GemmKParams& p, const f32x16 (&acc)[2][4],
                                                   const int64_t t0, const int64_t row0,
                                                   const int lane) {
  if ((t0 + 1) * 32 >= p.N) return;
  const bool wide = ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.c) & 7) == 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t gcol = t0 * 32 + 8 * q + 4 * (lane >> 5);
    const int64_t ocol = (t0 >> 1) * 32 + 8 * q + 4 * (lane >> 5);
    float bg[4] = {0.f, 0.f, 0.f, 0.f}, bu[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      const uint16_t* bp = reinterpret_cast<const uint16_t*>(p.bias) + gcol;
      const u32x2 g = *reinterpret_cast<const u32x2*>(bp);
      const u32x2 u = *reinterpret_cast<const u32x2*>(bp + 32);
      bg[0] = lo_f32<T>(g.x); bg[1] = hi_f32<T>(g.x); bg[2] = lo_f32<T>(g.y); bg[3] = hi_f32<T>(g.y);
      bu[0] = lo_f32<T>(u.x); bu[1] = hi_f32<T>(u.x); bu[2] = lo_f32<T>(u.y); bu[3] = hi_f32<T>(u.y);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = row0 + i * 32;
      if (row >= p.M) continue;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = silu_mul_acc<T>(acc[0][i][4 * q + e] + bg[e], acc[1][i][4 * q + e] + bu[e]);
      uint16_t* dst = reinterpret_cast<uint16_t*>(p.c) + row * p.ldc + ocol;
      u32x2 w;
      w.x = pack2<T>(o[0], o[1]);
      w.y = pack2<T>(o[2], o[3]);
      if (wide) {
        *reinterpret_cast<u32x2*>(dst) = w;
      } else {
        dst[0] = (uint16_t)(w.x & 0xffffu); dst[1] = (uint16_t)(w.x >> 16);
        dst[2] = (uint16_t)(w.y & 0xffffu); dst[3] = (uint16_t)(w.y >> 16);
      }
    }
  }
}

constexpr int W4_KC = 128;  // K granularity of the plan (split-K units, LDS chunk of the small-M kernels)

// dot2 GEMV for M <= 4 (w4_gemv.hip): no MFMA, activations resident in LDS, K split inside the workgroup
// norm: with the RMSNorm prologue (one more fp32 row in LDS)
bool gemv_supported(int64_t M, int64_t K, int64_t group_size, bool norm = false);
void launch_gemv(const GemmKParams& kp, int dtype, int ng, hipStream_t st);
// K splits across workgroups (1 = none; > 1 only when the caller takes fp32 partial slabs)
int gemv_global_splits(int64_t M, int64_t K, int64_t N, bool partials_ok);

// lean weight-streaming kernel for M <= 32 (w4_small.hip): BM = 32, BN = 128, 256 threads
void launch_gemm_small(const GemmKParams& kp, int dtype, int ng, int n_blocks, hipStream_t st);
constexpr size_t W4_SMALL_LDS_BYTES = 2 * 32 * 256;

// K-sliced weight stream for M <= 32 (mt = 1) and 33 <= M <= 64 (mt = 2: two row tiles per unpacked
// weight word) (w4_ks.hip): NW waves x CW chunks of K per workgroup, the activations of a wave's K
// slice live in its registers, partial tiles meet in LDS once per tile
void launch_gemm_ks(const GemmKParams& kp, int dtype, int ng, int cw, int nw, int n_blocks,
                    hipStream_t st, int mt = 1);
bool gemm_ks_config_ok(int ng, int cw, int nw, int mt = 1);
constexpr size_t w4_ks_lds_bytes(int nw) { return (size_t)4 * nw * 4096 + 64; }  // 4 partial slots + counters

// 65 <= M <= 128: all rows in one workgroup, 64-deep chunks, two workgroups per CU next to the decode
// attention stream (w4_m128.hip, round 5); wd = weight ring depth, (64-deep chunks per split) % wd == 0
void launch_gemm_m128(const GemmKParams& kp, int dtype, int group_size, int wd, int kw, int ct, int adma, int n_blocks, hipStream_t st);
constexpr size_t W4_M128_LDS_BYTES = 2 * 128 * 128;

// warp-specialised large-M kernel (w4_ws.hip): BM = 256, BN = 128, 512 threads
void launch_gemm_ws(const GemmKParams& kp, int dtype, int ng, int n_blocks, hipStream_t st);
constexpr size_t W4_WS_LDS_BYTES = 7 * (256 * 64) + 4 * (8 * 1024) + 2 * 4 * (1024 + 512);


// symmetric 256 x 256 kernel for large M x N (w4_xl.hip): 512 threads, 160 KiB LDS
void launch_gemm_xl(const GemmKParams& kp, int dtype, int ng, int n_blocks, hipStream_t st);
void launch_gemm_xl_sk(const GemmKParams& kp, int dtype, int ng, int n_wgs, hipStream_t st);
constexpr int W4_XL_SK_WGS = 256;                                  // one workgroup per CU (160 KiB of LDS each)
constexpr size_t W4_XL_SK_SLOT_BYTES = 256 * 256 * sizeof(float);  // a partial tile
constexpr size_t W4_XL_SK_SYNC_BYTES = 2048;                       // ticket + one flag per workgroup
constexpr size_t W4_XL_LDS_BYTES = 7 * (256 * 64) + 3 * (16 * 1024);

}  // namespace slm
