// attn_common.h -- parameter block and helpers shared by the attention kernels.
#pragma once
#include "common.h"

namespace slm {

constexpr int ATTN_TBL_ENT = 2048;  // block-table entries staged per chunk (8 KiB LDS)
constexpr float ATTN_M_INIT = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;

struct AttnKParams {
  void* out;
  const void* q;
  const void* kc;
  const void* vc;
  int64_t o_ts, o_hs, q_ts, q_hs, k_ss, k_hs, v_ss, v_hs;  // strides in elements
  const int* q_cu;
  const int* kv_cu;
  const int* bt;
  const int* bcu;
  const float* alibi;
  float* o_part;   // [n_tokens, n_heads, part_slots, head_dim]
  float* ml_part;  // [n_tokens, n_heads, part_slots, 2]
  int batch, n_tokens, n_heads, n_kv_heads, head_dim;
  int block_shift, block_mask;
  int group;      // q heads per kv head
  int n_chunks;   // group / GC
  int hpw_shift;  // log2(kv heads per wave-load)
  int hgw_shift;  // log2(head groups per workgroup)
  int nhgb;       // head-group blocks = n_kv_heads / (HPW * HGW)
  int n_splits;
  int window;
  float scale_log2;  // (softcap > 0 ? softcap : sm_scale) * log2(e)
  float pre_scale;   // sm_scale / softcap   (softcap > 0 only)
  float softcap;
  // mixed batches: a launch only processes sequences with rows_lo <= q_len * group < rows_hi
  // (device-side lengths); the other classes belong to the other launches of the same call
  int rows_lo, rows_hi;
  // balanced pure-decode partition (attn.hip, plan_attn): workgroup w of a head group streams the
  // KV tokens [w Q, (w + 1) Q) of the CONCATENATED histories of the batch, Q = max(ceil(W / P),
  // bal_qmin) rounded up to bal_align, W = kv_cu[batch], P = n_tokens * n_splits -- equal work per
  // workgroup whatever the spread of the sequence lengths.  A sequence is then covered by
  // pieces of consecutive workgroups: piece j of sequence b lives in partial slot j.
  int bal;         // 0 = classic (every sequence split into n_splits equal parts)
  int bal_qmin;    // bounds the pieces of one sequence by part_slots
  int bal_align;
  int part_slots;  // partial slots per (token, head): n_splits (classic) or the piece bound (balanced)
  int prio;        // > 0: the token kernel raises its wave priority (attn.hip)
};

// the piece size of the balanced partition: ONE formula for the stream kernel and the combine kernel
__device__ __forceinline__ int attn_bal_q(const AttnKParams& p, int W) {
  const int P = p.n_tokens * p.n_splits;
  int q = (W + P - 1) / P;
  q = q > p.bal_qmin ? q : p.bal_qmin;
  return (q + p.bal_align - 1) / p.bal_align * p.bal_align;
}

// tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e)); saturates correctly at +-inf, abs error ~1e-7
// (the reference kernel uses tanh.approx: common/fast_math.h:30-60).
__device__ __forceinline__ float fast_tanh(float x) {
  const float t = fast_exp2(x * (2.0f * LOG2E));
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + t);
}

// MFMA tile kernel (attn_tile.hip): prefill / chunked prefill / speculative verify.
// Returns SLM_OK after launching, or SLM_ERR_UNSUPPORTED when the shape is not covered
// (the caller then uses the token-major kernel).
bool attn_tile_supported(int head_dim);
int launch_attn_tile(const AttnKParams& kp, int dtype, int64_t max_rows, int64_t max_kv_len, hipStream_t st);

}  // namespace slm
