// kv_cache.hip -- KV-cache append for gfx950.
//
// Replaces llm::kernel::set_kv_cache (reference src/kernels/kv_cache_kernels.cu:9-78):
//   key_cache[slot_ids[t], h, d] = keys[t, h, d];  value_cache[...] = values[t, h, d]
// A slot row [n_kv_heads, head_dim] is contiguous in the cache (src/memory/kv_cache.cpp:21-27),
// so each token is one contiguous row copy: 16-byte vector copies, several tokens per
// workgroup, keys/values may have different token strides (kv_cache_kernels.cu:54-58).
#include "common.h"

namespace slm {

// VEC = bytes per lane access (16 or 2)
template <int VEC>
__global__ void __launch_bounds__(256) set_kv_cache_kernel(
    const int* __restrict__ slot_ids, const char* __restrict__ keys,
    const char* __restrict__ values, int64_t k_stride_b, int64_t v_stride_b,
    char* __restrict__ key_cache, char* __restrict__ value_cache, int64_t n_tokens,
    int row_bytes, int lanes_per_row, int rows_per_block) {
  const int r = threadIdx.x / lanes_per_row;
  const int c = threadIdx.x % lanes_per_row;
  const int64_t tok = (int64_t)blockIdx.x * rows_per_block + r;
  if (r >= rows_per_block || tok >= n_tokens) return;
  const int64_t slot = slot_ids[tok];
  const char* ks = keys + tok * k_stride_b;
  const char* vs = values + tok * v_stride_b;
  char* kd = key_cache + slot * (int64_t)row_bytes;
  char* vd = value_cache + slot * (int64_t)row_bytes;
  for (int off = c * VEC; off < row_bytes; off += lanes_per_row * VEC) {
    if constexpr (VEC == 16) {
      const u32x4 kx = *reinterpret_cast<const u32x4*>(ks + off);
      const u32x4 vx = *reinterpret_cast<const u32x4*>(vs + off);
      *reinterpret_cast<u32x4*>(kd + off) = kx;
      *reinterpret_cast<u32x4*>(vd + off) = vx;
    } else {
      const uint16_t kx = *reinterpret_cast<const uint16_t*>(ks + off);
      const uint16_t vx = *reinterpret_cast<const uint16_t*>(vs + off);
      *reinterpret_cast<uint16_t*>(kd + off) = kx;
      *reinterpret_cast<uint16_t*>(vd + off) = vx;
    }
  }
}

// decode-step input advance: every thread touches only its own elements, so the
// update is race-free in place
__global__ void __launch_bounds__(256) decode_advance_kernel(
    int32_t* __restrict__ positions, int32_t* __restrict__ kv_cu_lens,
    int32_t* __restrict__ new_cache_slots, const int32_t* __restrict__ block_table,
    const int32_t* __restrict__ block_cu_lens, int32_t n_seqs, int32_t shift, int32_t mask,
    int32_t* __restrict__ overflow_flag) {
  const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > n_seqs) return;
  if (b < n_seqs) {
    const int32_t len = positions[b] + 1;  // tokens already in the cache = position of the new token
    const int32_t base = block_cu_lens[b];
    const int32_t nblk = block_cu_lens[b + 1] - base;
    int32_t blk = len >> shift;
    if (blk >= nblk) {  // the host has not appended the block yet
      if (overflow_flag) atomicOr(overflow_flag, 1);
      blk = nblk > 0 ? nblk - 1 : 0;
    }
    positions[b] = len;
    new_cache_slots[b] = block_table[base + blk] + (len & mask);
  }
  kv_cu_lens[b] += b;  // b = 0 .. n_seqs: every sequence before this offset grew by one token
}

// Step-input build for ANY batch shape (beyond steady decode): prefix sums of the
// per-sequence lengths, then positions and cache slots of every new token.  ONE workgroup: a step
// has at most a few thousand sequences and tokens, the work is two scans and a fill (microseconds),
// and a single workgroup needs no cross-workgroup ordering between the scan and the fill.
constexpr int BSI_THREADS = 1024;
__global__ void __launch_bounds__(BSI_THREADS) build_step_inputs_kernel(
    const int32_t* __restrict__ q_lens, int32_t* __restrict__ kv_cached,
    const int32_t* __restrict__ block_table, const int32_t* __restrict__ block_cu_lens, int32_t n_seqs,
    int32_t shift, int32_t mask, int32_t n_tokens_padded, int32_t commit, int32_t* __restrict__ positions,
    int32_t* __restrict__ q_cu_lens, int32_t* __restrict__ kv_cu_lens, int32_t* __restrict__ new_cache_slots,
    int32_t* __restrict__ overflow_flag) {
  __shared__ int32_t wsum_q[BSI_THREADS / 64], wsum_kv[BSI_THREADS / 64];
  __shared__ int32_t carry[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    carry[0] = 0;
    carry[1] = 0;
    q_cu_lens[0] = 0;
    kv_cu_lens[0] = 0;
  }
  __syncthreads();
  // ---- inclusive scans of q and (cached + q) over the sequences, BSI_THREADS at a time
  for (int32_t base = 0; base < n_seqs; base += BSI_THREADS) {
    const int32_t b = base + tid;
    int32_t q = 0, kv = 0;
    if (b < n_seqs) {
      q = q_lens[b];
      q = q > 0 ? q : 0;
      kv = kv_cached[b] + q;
    }
    int32_t sq = q, skv = kv;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {  // wave-level inclusive scan
      const int32_t oq = __shfl_up(sq, d, 64), okv = __shfl_up(skv, d, 64);
      if (lane >= d) {
        sq += oq;
        skv += okv;
      }
    }
    if (lane == 63) {
      wsum_q[wave] = sq;
      wsum_kv[wave] = skv;
    }
    __syncthreads();
    int32_t pq = carry[0], pkv = carry[1];
    for (int w = 0; w < wave; ++w) {
      pq += wsum_q[w];
      pkv += wsum_kv[w];
    }
    if (b < n_seqs) {
      q_cu_lens[b + 1] = pq + sq;
      kv_cu_lens[b + 1] = pkv + skv;
    }
    __syncthreads();
    if (tid == BSI_THREADS - 1) {
      carry[0] = pq + sq;
      carry[1] = pkv + skv;
    }
    __syncthreads();
  }
  const int32_t n_tok = carry[0];
  // more new tokens than rows: the tail would be dropped while q_cu / kv_cu (and the commit) count
  // it -- flag bit 1 (value 2) and do NOT advance the cache positions over tokens nobody appends
  const bool too_many = n_tok > n_tokens_padded;
  if (too_many && tid == 0 && overflow_flag) atomicOr(overflow_flag, 2);
  // ---- fill: token t belongs to the sequence b with q_cu[b] <= t < q_cu[b + 1] (this workgroup's own
  // writes: visible after the barriers above)
  for (int32_t t = tid; t < n_tokens_padded; t += BSI_THREADS) {
    if (t >= n_tok) {  // graph padding rows (batch.cpp:219-244): position 0, slot 0
      positions[t] = 0;
      new_cache_slots[t] = 0;
      continue;
    }
    int32_t lo = 0, hi = n_seqs;
    while (lo < hi) {
      const int32_t mid = (lo + hi) >> 1;
      if (q_cu_lens[mid + 1] <= t) lo = mid + 1; else hi = mid;
    }
    const int32_t b = lo;
    const int32_t j = kv_cached[b] + (t - q_cu_lens[b]);  // position in the sequence (batch.cpp:155)
    const int32_t bbase = block_cu_lens[b];
    const int32_t nblk = block_cu_lens[b + 1] - bbase;
    int32_t blk = j >> shift;
    if (blk >= nblk) {  // the host has not appended the block yet
      if (overflow_flag) atomicOr(overflow_flag, 1);
      blk = nblk - 1;
    }
    positions[t] = j;
    // (a sequence without any block: slot 0, its table range is empty -- nothing of it is read)
    new_cache_slots[t] = blk >= 0 ? block_table[bbase + blk] + (j & mask) : 0;  // sequence.cpp:303-317
  }
  if (commit && !too_many) {  // Sequence::commit_kv_cache (batch.cpp:197): after every read of kv_cached above
    __syncthreads();
    for (int32_t b = tid; b < n_seqs; b += BSI_THREADS) {
      const int32_t q = q_lens[b];
      if (q > 0) kv_cached[b] += q;
    }
  }
}

}  // namespace slm

using namespace slm;

extern "C" SLM_API int slm_set_kv_cache(const int32_t* slot_ids, const void* keys,
                                        const void* values, int64_t k_token_stride,
                                        int64_t v_token_stride, void* key_cache,
                                        void* value_cache, int64_t n_tokens, int32_t n_kv_heads,
                                        int32_t head_dim, int32_t dtype, void* stream) {
  if (n_tokens == 0) return SLM_OK;
  if (!slot_ids || !keys || !values || !key_cache || !value_cache) return SLM_ERR_INVALID_ARG;
  if (n_tokens < 0 || n_kv_heads <= 0 || head_dim <= 0) return SLM_ERR_INVALID_ARG;
  if (dtype != SLM_F16 && dtype != SLM_BF16) return SLM_ERR_UNSUPPORTED;
  const int64_t row_bytes64 = (int64_t)n_kv_heads * head_dim * 2;
  if (row_bytes64 > (1 << 24)) return SLM_ERR_INVALID_ARG;
  const int row_bytes = (int)row_bytes64;
  const int64_t ksb = k_token_stride * 2, vsb = v_token_stride * 2;
  const bool vec16 = (row_bytes % 16 == 0) && (ksb % 16 == 0) && (vsb % 16 == 0) &&
                     aligned16(keys) && aligned16(values) && aligned16(key_cache) &&
                     aligned16(value_cache);
  hip_clear_error();
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int vec = vec16 ? 16 : 2;
  int lanes = (row_bytes + vec - 1) / vec;
  if (lanes > 256) lanes = 256;
  // round lanes-per-row up to a power of two <= 256 so rows do not straddle oddly
  int lpr = 1;
  while (lpr < lanes) lpr <<= 1;
  const int rows_per_block = 256 / lpr;
  const unsigned grid = (unsigned)((n_tokens + rows_per_block - 1) / rows_per_block);
  if (vec16)
    hipLaunchKernelGGL(set_kv_cache_kernel<16>, dim3(grid), dim3(256), 0, st, slot_ids,
                       (const char*)keys, (const char*)values, ksb, vsb, (char*)key_cache,
                       (char*)value_cache, n_tokens, row_bytes, lpr, rows_per_block);
  else
    hipLaunchKernelGGL(set_kv_cache_kernel<2>, dim3(grid), dim3(256), 0, st, slot_ids,
                       (const char*)keys, (const char*)values, ksb, vsb, (char*)key_cache,
                       (char*)value_cache, n_tokens, row_bytes, lpr, rows_per_block);
  return hip_check_launch();
}

extern "C" SLM_API int slm_decode_advance(int32_t* positions, int32_t* kv_cu_lens,
                                          int32_t* new_cache_slots, const int32_t* block_table,
                                          const int32_t* block_cu_lens, int32_t n_seqs,
                                          int32_t block_size, int32_t* overflow_flag, void* stream) {
  if (n_seqs == 0) return SLM_OK;
  if (!positions || !kv_cu_lens || !new_cache_slots || !block_table || !block_cu_lens || n_seqs < 0)
    return SLM_ERR_INVALID_ARG;
  if (block_size <= 0 || !is_pow2(block_size)) return SLM_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  const dim3 grid((unsigned)((n_seqs + 1 + 255) / 256)), blk(256);
  hipLaunchKernelGGL(decode_advance_kernel, grid, blk, 0, st, positions, kv_cu_lens, new_cache_slots,
                     block_table, block_cu_lens, n_seqs, ilog2(block_size), block_size - 1,
                     overflow_flag);
  return hip_check_launch();
}

extern "C" SLM_API int slm_build_step_inputs(const int32_t* q_lens, int32_t* kv_cached, const int32_t* block_table,
                                             const int32_t* block_cu_lens, int32_t n_seqs, int32_t block_size,
                                             int32_t n_tokens_padded, int32_t commit, int32_t* positions,
                                             int32_t* q_cu_lens, int32_t* kv_cu_lens, int32_t* new_cache_slots,
                                             int32_t* overflow_flag, void* stream) {
  if (n_seqs < 0 || n_tokens_padded < 0) return SLM_ERR_INVALID_ARG;
  if (!q_cu_lens || !kv_cu_lens) return SLM_ERR_INVALID_ARG;
  if (n_seqs > 0 && (!q_lens || !kv_cached || !block_table || !block_cu_lens)) return SLM_ERR_INVALID_ARG;
  if (n_tokens_padded > 0 && (!positions || !new_cache_slots)) return SLM_ERR_INVALID_ARG;
  if (block_size <= 0 || (block_size & (block_size - 1))) return SLM_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hip_clear_error();
  hipLaunchKernelGGL(build_step_inputs_kernel, dim3(1), dim3(BSI_THREADS), 0, st, q_lens, kv_cached, block_table,
                     block_cu_lens, n_seqs, ilog2(block_size), block_size - 1, n_tokens_padded, commit, positions,
                     q_cu_lens, kv_cu_lens, new_cache_slots, overflow_flag);
  return hip_check_launch();
}
