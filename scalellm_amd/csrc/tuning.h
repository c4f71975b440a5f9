// tuning.h -- launch-shape overrides of libslm_hip, in ONE table.
//
// The kernels' launch heuristics (plan_attn, plan_gemm, ...) have override knobs for sweeps and
// for tests that force a particular kernel.  They are NOT read from the environment per call: the
// SLM_* environment variables are parsed exactly once, when the library is first used, into this
// table; afterwards only the explicit C-ABI setter (slm_tuning_set / slm_tuning_clear,
// include/slm_hip.h section 0) changes it.  A lookup is one relaxed atomic load.
#pragma once
#include <stdint.h>

namespace slm {

enum TuneKey : int {
  TUNE_ATTN_NW = 0,       // SLM_ATTN_NW           waves per token-kernel workgroup (1/2/4/8)
  TUNE_ATTN_SPLITS,       // SLM_ATTN_SPLITS       forced split-KV count
  TUNE_ATTN_HGW,          // SLM_ATTN_HGW          head groups per workgroup cap
  TUNE_ATTN_TILE,         // SLM_ATTN_TILE         0 = never use the MFMA tile kernel
  TUNE_ATTN_TILE_SPLITS,  // SLM_ATTN_TILE_SPLITS  forced split count of the tile kernel
  TUNE_ATTN_TILE_PF,      // SLM_ATTN_TILE_PF      tile kernel staging form: 0 = no prefetch, 2 = 32-row single-buffer tiles, 4 = register-staged 64-row tiles, 5 = LDS-DMA without the cross-tile pipeline (default 1: LDS-DMA + pipeline where they exist)
  TUNE_ATTN_U,            // SLM_ATTN_U            K/V register ring depth (2/4)
  TUNE_ATTN_NT,           // SLM_ATTN_NT           non-temporal KV loads on/off
  TUNE_ATTN_TILE_DECODE,  // SLM_ATTN_TILE_DECODE  min GQA group at which q_len = 1 goes to the tile kernel (default 8, 0 = never)
  TUNE_ATTN_BAL,          // SLM_ATTN_BAL          0 = classic per-sequence split-KV for pure decode (no balanced partition)
  TUNE_ATTN_W,            // SLM_ATTN_W            16-byte chunks of a K / V row per lane in the decode stream kernel: 1 / 2 force a form (default: by batch size and KV heads)
  TUNE_ATTN_PRIO,         // SLM_ATTN_PRIO         0 = the decode stream kernel does not raise its wave priority (default 1: s_setprio 3)
  TUNE_W4_GEMV,           // SLM_W4_GEMV           0 off, 1 = M == 1 only, 2 = M <= 4
  TUNE_W4_GEMV_KS,        // SLM_W4_GEMV_KS        forced K slices per GEMV workgroup (1/2/4/8)
  TUNE_W4_SMALL,          // SLM_W4_SMALL          0 = never use the small-M kernel
  TUNE_W4_MT,             // SLM_W4_MT             forced M tile (1/2/4/8/16)
  TUNE_W4_MT_WIDE,        // SLM_W4_MT_WIDE        forced M tile of wide layers (N >= 16384) at 64 < M <= 128
  TUNE_W4_NTW,            // SLM_W4_NTW
  TUNE_W4_PC,             // SLM_W4_PC
  TUNE_W4_SPLITK,         // SLM_W4_SPLITK         forced split-K
  TUNE_W4_POST,           // SLM_W4_POST
  TUNE_W4_KS,             // SLM_W4_KS             0 = never use the K-sliced small-M kernel
  TUNE_W4_KS_CW,          // SLM_W4_KS_CW          forced chunks of K per wave (1/2/4)
  TUNE_W4_KS_NW,          // SLM_W4_KS_NW          forced waves per workgroup (4/8/16)
  TUNE_W4_KS_TPW,         // SLM_W4_KS_TPW         forced column tiles per workgroup
  TUNE_W4_KS_DBG,         // SLM_W4_KS_DBG         probe bits (1 = no activation loads, 2 = no weight loads): WRONG results
  TUNE_W4_KS_MT2,         // SLM_W4_KS_MT2         1 = 33 <= M <= 64 on the two-row-tile K-sliced stream (K <= 4096), 2 = any K, 0 = never; default: on unless the call carries SLM_W4_SHARES_CHIP
  TUNE_W4_M128,           // SLM_W4_M128           w4_m128.hip at 65 <= M <= 128: 1 = always, 0 = never (default: K >= 8192)
  TUNE_W4_M128_WD,        // SLM_W4_M128_WD        weight ring depth of w4_m128.hip in 64-deep chunks (2 / 4)
  TUNE_W4_M128_SPLITS,    // SLM_W4_M128_SPLITS    workgroups w4_m128.hip's split-K aims at (default 512 = two per CU)
  TUNE_W4_M128_KW,        // SLM_W4_M128_KW        waves per column tile of w4_m128.hip (1 / 2; default: by grid size)
  TUNE_W4_SPLIT_TARGET,   // SLM_W4_SPLIT_TARGET   workgroups the general kernel's split-K aims at for M > 64 (default 512)
  TUNE_W4_M128_CT,        // SLM_W4_M128_CT        column tiles per workgroup of w4_m128.hip (4 = 128 columns, 8 = 256 columns)
  TUNE_ATTN_TILE_KV2,     // SLM_ATTN_TILE_KV2     two wave groups share a query tile's KV range in the prefill tile kernel: 1 always, 0 never (default: small plain-prefill grids)
  TUNE_W4_XL_MODEL,       // SLM_W4_XL_MODEL       0 = the 256 x 256 kernel only where its tiles fill whole rounds (no round-count comparison with 256 x 128 tiles)
  TUNE_W4_M128_ADMA,      // SLM_W4_M128_ADMA      activations of the 256-column form by LDS-DMA (1) or through registers (0)
  TUNE_W4_XL_SK,          // SLM_W4_XL_SK          stream-K form of the 256 x 256 kernel: 0 never, 2 wherever it applies (default 1: where the round model says it wins)
  TUNE_COUNT
};

constexpr int32_t TUNE_UNSET = INT32_MIN;

// value of the knob, or `dflt` when it was never set
int tune_get(TuneKey k, int dflt);
bool tune_is_set(TuneKey k);

}  // namespace slm
