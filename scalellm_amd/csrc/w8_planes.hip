// w8_planes.hip -- 8-bit weights (Marlin num_bits = 8) as TWO int4 planes of the int4 GEMM.
//
// The reference's quantised linears accept 4 and 8 bits (qlinear_awq_marlin_impl.cpp:25-26,
// qlinear_gptq_marlin_impl.cpp; CPU path construct_weights, qlinear_impl.cpp:21-100, generic in
// `bits`); marlin::gptq_gemm / gptq_repack / awq_repack take num_bits (marlin.h:17-37).  Every
// BASELINE config is int4, so the 8-bit case gets no kernel of its own -- it is REWRITTEN, exactly,
// into the operator the int4 kernels already implement:
//
//   q = 16 hi + lo,  z = 16 zh + zl   (hi, lo, zl in 0..15;  zh = z >> 4 in 0..16)
//   s (q - z) = (16 s) (hi - zh) + s (lo - zl)
//   sum_k x_k s_g (q_k - z_g) = sum_k x_k (16 s_g)(hi_k - zh_g)  +  sum_k x_k s_g (lo_k - zl_g)
//
// i.e. an int4 GEMM over K' = 2K packed rows -- rows [0, K) the high nibbles with scale 16 s (exact
// in T: a power of two) and zero point zh, rows [K, 2K) the low nibbles with scale s and zero
// point zl -- against the activations read twice, which the act-order column gather
// (slm_w4_gemm_args.perm, w4_permute_cols_kernel) already provides: perm2[k'] = perm[k' mod K].
// Same bytes from HBM as a native 8-bit kernel would read (2 x 4 bits per weight); twice the MFMA
// work of an int4 GEMM, which is free in the weight-streaming regime (M <= ~64) and the price at
// large M.  Numerics: the small-M kernels (post-scaled form) accumulate x (magic + nibble) exactly and
// apply scale and zero point in fp32 -- closer to the fp32 oracle than the reference's
// dequant-to-T-then-mma; the large-M kernels round each plane's dequantised value to T separately
// (two roundings of <= half an ulp of the PLANE's magnitude instead of one of the sum).
//
// Checkpoint formats (bit-exact integer work, as the int4 prepack):
//   GPTQ  qweight [K/4, N] int32, byte (k % 4) of word [k/4, n]; qzeros [G, N/4], byte (n % 4),
//         zero = stored + 1 (qlinear_impl.cpp:45: 1..256); no qzeros = symmetric, zero = 128
//   AWQ   qweight [K, N/4] int32, byte order [0, 2, 1, 3] (tests/kernels/quant_utils.py:182-184:
//         byte i of word [k, n/4] holds column 4 (n/4) + order[i]); qzeros [G, N/4] likewise,
//         zero = stored
#include "common.h"

namespace slm {

__device__ __forceinline__ int awq8_pos(int col_in_word) {  // [0,2,1,3] interleave
  return (col_in_word >> 1) + 2 * (col_in_word & 1);
}

__device__ __forceinline__ int64_t w8_paired_src_col(int format, int64_t n_packed, int64_t N) {
  if (!(format & SLM_W4_PAIRED)) return n_packed;
  return (n_packed >> 6) * 32 + (n_packed & 31) + ((n_packed & 32) ? N / 2 : 0);
}

// packed word widx of the [2K/64][N/32][64][4] layout (w4.hip header); packed row kp < K: high
// nibble of checkpoint row perm[kp], kp >= K: low nibble of row perm[kp - K]
__global__ void __launch_bounds__(256) w8_prepack_weight_kernel(
    int format, const uint32_t* __restrict__ qweight, const int* __restrict__ perm, int64_t K,
    int64_t N, uint32_t* __restrict__ wq) {
  const int64_t widx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (widx >= N * (2 * K) / 8) return;
  const int j = (int)(widx & 3);
  const int lane = (int)((widx >> 2) & 63);
  const int64_t tile = widx >> 8;
  const int64_t nt = tile % (N / 32), kt = tile / (N / 32);
  const int64_t n = w8_paired_src_col(format, nt * 32 + (lane & 31), N);
  const int64_t kb = kt * 64 + j * 16 + (lane >> 5) * 8;
  const bool gptq = (format & SLM_W4_FORMAT_MASK) == SLM_W8_GPTQ;
  uint32_t out = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t kp = kb + e;
    const bool lo_plane = kp >= K;
    const int64_t ks = lo_plane ? kp - K : kp;
    const int64_t k = perm ? (int64_t)perm[ks] : ks;
    uint32_t q8 = 0u;
    if (k >= 0)
      q8 = gptq ? (qweight[(k / 4) * N + n] >> (8 * (k % 4))) & 0xFFu
                : (qweight[k * (N / 4) + n / 4] >> (8 * awq8_pos((int)(n % 4)))) & 0xFFu;
    const uint32_t q = lo_plane ? (q8 & 0xFu) : (q8 >> 4);
    const int pos = (e >> 1) + 4 * (e & 1);
    out |= q << (4 * pos);
  }
  wq[widx] = out;
}

// perm2[kp] = perm[kp mod K] (identity when perm == NULL): the activation column of packed row kp
__global__ void __launch_bounds__(256) w8_gather_index_kernel(const int* __restrict__ perm, int64_t K,
                                                              int* __restrict__ perm2) {
  const int64_t kp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (kp >= 2 * K) return;
  const int64_t ks = kp >= K ? kp - K : kp;
  perm2[kp] = perm ? perm[ks] : (int)ks;
}

// sz row gp of the packed table covers packed rows [gp * gpk, (gp + 1) * gpk)
__global__ void __launch_bounds__(256) w8_prepack_sz_kernel(
    int format, const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ scales, int64_t K,
    int64_t N, int64_t gs_src, int64_t gpk, int dtype, uint32_t* __restrict__ sz) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t Gp = 2 * K / gpk;
  if (idx >= Gp * N) return;
  const int64_t gp = idx / N, n = w8_paired_src_col(format, idx % N, N);
  const int64_t k0 = gp * gpk;
  const bool lo_plane = k0 >= K;
  const int64_t g = (lo_plane ? k0 - K : k0) / gs_src;
  uint32_t z8 = 128u;  // no zero-point tensor: symmetric, zero = 2^(bits-1) (Marlin has_zp = false)
  if (qzeros) {
    const uint32_t zw = qzeros[g * (N / 4) + n / 4];
    if ((format & SLM_W4_FORMAT_MASK) == SLM_W8_GPTQ)
      z8 = ((zw >> (8 * (n % 4))) & 0xFFu) + 1u;  // qlinear_impl.cpp:45 (zeros.add_(1)): 1..256
    else
      z8 = (zw >> (8 * awq8_pos((int)(n % 4)))) & 0xFFu;
  }
  const uint32_t z = lo_plane ? (z8 & 0xFu) : (z8 >> 4);  // zh <= 16: magic + 16 is still exact in T
  uint32_t sbits = scales[g * N + n];
  if (!lo_plane) {  // 16 s: exact (power of two) unless it overflows T, which no real scale does
    if (dtype == SLM_BF16) {
      const float s = __builtin_bit_cast(float, sbits << 16) * 16.0f;
      sbits = __builtin_bit_cast(uint32_t, s) >> 16;
    } else {
      const _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)sbits);
      const _Float16 h16 = (_Float16)((float)h * 16.0f);
      sbits = (uint32_t)__builtin_bit_cast(uint16_t, h16);
    }
  }
  const uint32_t zm = (dtype == SLM_BF16 ? 0x4300u : 0x6400u) + z;
  sz[idx] = (sbits & 0xffffu) | (zm << 16);
}

}  // namespace slm

using namespace slm;

static bool w8_format_ok(int32_t format, int64_t N) {
  const int32_t base = format & SLM_W4_FORMAT_MASK;
  if (format & ~(SLM_W4_FORMAT_MASK | SLM_W4_PAIRED)) return false;
  if (base != SLM_W8_GPTQ && base != SLM_W8_AWQ) return false;
  return !(format & SLM_W4_PAIRED) || N % 64 == 0;
}

static bool w8_is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" {

SLM_API int64_t slm_w8_packed_rows(int64_t K) { return K > 0 ? 2 * K : 0; }

// the group size the PACKED table is written at (what slm_w4_gemm_args.group_size must say): the
// checkpoint's, except that a per-channel / odd-sized group (the int4 GEMM wants 32, 64 or a power
// of two >= 128) is written out at 128-row granularity
SLM_API int64_t slm_w8_packed_group_size(int64_t K, int64_t group_size) {
  if (K <= 0 || group_size <= 0 || K % group_size) return 0;
  if (group_size == 32 || group_size == 64) return group_size;
  if (group_size >= 128 && group_size % 128 == 0 && w8_is_pow2(group_size)) return group_size;
  return (group_size % 128 == 0) ? 128 : 0;
}

SLM_API int slm_w8_prepack_weights(int32_t format, const int32_t* qweight, const int32_t* perm, int64_t K,
                                   int64_t N, void* wq_out, int32_t* perm2_out, void* stream) {
  if (!qweight || !wq_out || !perm2_out) return SLM_ERR_INVALID_ARG;
  if (!w8_format_ok(format, N)) return SLM_ERR_UNSUPPORTED;
  if (K <= 0 || N <= 0 || K % 64 || N % 32) return SLM_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const int64_t words = 2 * K * N / 8;
  hipLaunchKernelGGL(w8_prepack_weight_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st,
                     format, reinterpret_cast<const uint32_t*>(qweight), perm, K, N,
                     reinterpret_cast<uint32_t*>(wq_out));
  hipLaunchKernelGGL(w8_gather_index_kernel, dim3((unsigned)((2 * K + 255) / 256)), dim3(256), 0, st, perm, K,
                     perm2_out);
  return hip_check_launch();
}

SLM_API int slm_w8_prepack_sz(int32_t format, const int32_t* qzeros, const void* scales, int64_t K,
                              int64_t N, int64_t group_size, int32_t dtype, void* sz_out, void* stream) {
  if (!scales || !sz_out) return SLM_ERR_INVALID_ARG;
  if (!w8_format_ok(format, N)) return SLM_ERR_UNSUPPORTED;
  if (dtype != SLM_F16 && dtype != SLM_BF16) return SLM_ERR_UNSUPPORTED;
  if (K <= 0 || N <= 0 || N % 32) return SLM_ERR_UNSUPPORTED;
  const int64_t gpk = slm_w8_packed_group_size(K, group_size);
  if (gpk <= 0 || K % gpk) return SLM_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const int64_t Gp = 2 * K / gpk;
  hipLaunchKernelGGL(w8_prepack_sz_kernel, dim3((unsigned)((Gp * N + 255) / 256)), dim3(256), 0, st, format,
                     reinterpret_cast<const uint32_t*>(qzeros), reinterpret_cast<const uint16_t*>(scales), K,
                     N, group_size, gpk, dtype, reinterpret_cast<uint32_t*>(sz_out));
  return hip_check_launch();
}

}  // extern "C"
