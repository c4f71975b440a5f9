// capi.hip -- status strings / version / tuning table of the C ABI (include/slm_hip.h).
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "tuning.h"

namespace slm {

namespace {
const char* const kTuneNames[TUNE_COUNT] = {
    "SLM_ATTN_NW",          "SLM_ATTN_SPLITS",      "SLM_ATTN_HGW",     "SLM_ATTN_TILE",
    "SLM_ATTN_TILE_SPLITS", "SLM_ATTN_TILE_PF",     "SLM_ATTN_U",       "SLM_ATTN_NT",
    "SLM_ATTN_TILE_DECODE", "SLM_ATTN_BAL",         "SLM_ATTN_W",       "SLM_ATTN_PRIO",
    "SLM_W4_GEMV",          "SLM_W4_GEMV_KS",   "SLM_W4_SMALL",
    "SLM_W4_MT",            "SLM_W4_MT_WIDE",       "SLM_W4_NTW",           "SLM_W4_PC",        "SLM_W4_SPLITK",
    "SLM_W4_POST",
    "SLM_W4_KS",            "SLM_W4_KS_CW",         "SLM_W4_KS_NW",     "SLM_W4_KS_TPW",
    "SLM_W4_KS_DBG",        "SLM_W4_KS_MT2",
    "SLM_W4_M128",          "SLM_W4_M128_WD",       "SLM_W4_M128_SPLITS",   "SLM_W4_M128_KW",   "SLM_W4_SPLIT_TARGET",
    "SLM_W4_M128_CT",       "SLM_ATTN_TILE_KV2",    "SLM_W4_XL_MODEL",      "SLM_W4_M128_ADMA",
    "SLM_W4_XL_SK",
};
std::atomic<int32_t> g_tune[TUNE_COUNT];
std::once_flag g_tune_once;

// the ONE place the environment is read: once per process, at the first lookup
void tune_init() {
  for (int k = 0; k < TUNE_COUNT; ++k) {
    const char* v = getenv(kTuneNames[k]);
    g_tune[k].store((v && *v) ? (int32_t)atoi(v) : TUNE_UNSET, std::memory_order_relaxed);
  }
}
int tune_index(const char* name) {
  if (!name) return -1;
  for (int k = 0; k < TUNE_COUNT; ++k)
    if (strcmp(name, kTuneNames[k]) == 0) return k;
  return -1;
}
}  // namespace

int tune_get(TuneKey k, int dflt) {
  std::call_once(g_tune_once, tune_init);
  const int32_t v = g_tune[k].load(std::memory_order_relaxed);
  return v == TUNE_UNSET ? dflt : (int)v;
}
bool tune_is_set(TuneKey k) {
  std::call_once(g_tune_once, tune_init);
  return g_tune[k].load(std::memory_order_relaxed) != TUNE_UNSET;
}

}  // namespace slm

extern "C" {

SLM_API const char* slm_status_string(int status) {
  switch (status) {
    case SLM_OK: return "ok";
    case SLM_ERR_INVALID_ARG: return "invalid argument";
    case SLM_ERR_UNSUPPORTED: return "unsupported dtype / shape";
    case SLM_ERR_WORKSPACE: return "workspace missing or too small";
    case SLM_ERR_LAUNCH: return "kernel launch failed";
    case SLM_ERR_ALIGNMENT: return "pointer or stride not 16-byte aligned";
    default: return "unknown status";
  }
}

SLM_API const char* slm_last_hip_error(void) {
  return hipGetErrorString((hipError_t)slm::hip_last_error_slot());
}

SLM_API const char* slm_version(void) { return "slm_hip 0.2.0 (gfx950)"; }

SLM_API int slm_tuning_set(const char* name, int32_t value) {
  const int k = slm::tune_index(name);
  if (k < 0 || value == slm::TUNE_UNSET) return SLM_ERR_INVALID_ARG;
  std::call_once(slm::g_tune_once, slm::tune_init);
  slm::g_tune[k].store(value, std::memory_order_relaxed);
  return SLM_OK;
}

SLM_API int slm_tuning_clear(const char* name) {
  std::call_once(slm::g_tune_once, slm::tune_init);
  if (!name) {  // NULL clears every knob (environment values included)
    for (int k = 0; k < slm::TUNE_COUNT; ++k) slm::g_tune[k].store(slm::TUNE_UNSET, std::memory_order_relaxed);
    return SLM_OK;
  }
  const int k = slm::tune_index(name);
  if (k < 0) return SLM_ERR_INVALID_ARG;
  slm::g_tune[k].store(slm::TUNE_UNSET, std::memory_order_relaxed);
  return SLM_OK;
}

SLM_API int slm_tuning_get(const char* name, int32_t* value, int32_t* is_set) {
  const int k = slm::tune_index(name);
  if (k < 0) return SLM_ERR_INVALID_ARG;
  std::call_once(slm::g_tune_once, slm::tune_init);
  const int32_t v = slm::g_tune[k].load(std::memory_order_relaxed);
  if (is_set) *is_set = v != slm::TUNE_UNSET;
  if (value) *value = v != slm::TUNE_UNSET ? v : 0;
  return SLM_OK;
}

// ---- section 7: one lane or two (host policy; include/slm_hip.h) --------------------------------
namespace {
struct LaneEntry {
  int32_t world, n_heads, n_kv_heads, head_dim, t_bucket, kv_len;
  int64_t w_bytes;
  float one_us, two_us;
};
std::mutex g_lane_mu;
LaneEntry g_lane_tab[256];
int g_lane_n = 0, g_lane_next = 0;

bool lane_same_geometry(const LaneEntry& e, const slm_lane_query* q) {
  return e.world == q->world_size && e.n_heads == q->n_heads && e.n_kv_heads == q->n_kv_heads &&
         e.head_dim == q->head_dim && e.w_bytes == q->layer_weight_bytes;
}
// the nearest recorded context length for this geometry and batch-size bucket, within a factor 1.5
const LaneEntry* lane_lookup(const slm_lane_query* q) {
  const LaneEntry* best = nullptr;
  double best_d = 1.5;
  const int32_t tb = (q->n_tokens + 31) / 32;
  for (int i = 0; i < g_lane_n; ++i) {
    const LaneEntry& e = g_lane_tab[i];
    if (!lane_same_geometry(e, q) || e.t_bucket != tb || e.kv_len <= 0 || q->kv_max_seq_len <= 0) continue;
    const double r = e.kv_len > q->kv_max_seq_len ? (double)e.kv_len / q->kv_max_seq_len
                                                  : (double)q->kv_max_seq_len / e.kv_len;
    if (r <= best_d) { best_d = r; best = &e; }
  }
  return best;
}
bool lane_hard_conditions(const slm_lane_query* q) {
  if (!q || q->lanes_min == 0) return false;
  if (q->world_size != 1 && !q->tp_lanes_ok) return false;
  return q->q_max_seq_len == 1 && q->n_tokens == q->n_seqs && q->n_tokens >= 64;
}
}  // namespace

SLM_API int32_t slm_decode_lane_split(const slm_lane_query* q) {
  if (!lane_hard_conditions(q)) return 0;
  const int64_t T = q->n_tokens;
  const int32_t half = (int32_t)((T / 2 + 31) / 32 * 32);
  if (q->lanes_min > 0) return T >= q->lanes_min ? half : 0;
  {  // auto: a recorded measurement decides when there is one
    std::lock_guard<std::mutex> lk(g_lane_mu);
    if (const LaneEntry* e = lane_lookup(q)) return e->two_us < 0.985f * e->one_us ? half : 0;
  }
  // ... else the constants (profiles/r04_lanes_sweep_w2.jsonl): 96 <= T <= 256, long
  // sequences (>= 12 MiB of K + V each), the KV stream >= 8 x the layer's weights
  const int64_t eb = q->kv_elem_bytes > 0 ? q->kv_elem_bytes : 2;
  const int64_t kv_seq = 2 * eb * q->n_kv_heads * q->head_dim * (int64_t)q->kv_max_seq_len;
  if (T < 96 || T > 256) return 0;
  if (kv_seq < ((int64_t)12 << 20)) return 0;
  if (kv_seq * T < 8 * q->layer_weight_bytes) return 0;
  return half;
}

SLM_API int slm_decode_lane_policy_record(const slm_lane_query* q, float one_lane_us, float two_lane_us) {
  if (!q || q->n_tokens <= 0 || q->kv_max_seq_len <= 0 || !(one_lane_us > 0.f) || !(two_lane_us > 0.f))
    return SLM_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g_lane_mu);
  LaneEntry e{q->world_size, q->n_heads, q->n_kv_heads, q->head_dim, (q->n_tokens + 31) / 32, q->kv_max_seq_len,
              q->layer_weight_bytes, one_lane_us, two_lane_us};
  for (int i = 0; i < g_lane_n; ++i)   // same point measured again: replace
    if (lane_same_geometry(g_lane_tab[i], q) && g_lane_tab[i].t_bucket == e.t_bucket && g_lane_tab[i].kv_len == e.kv_len) {
      g_lane_tab[i] = e;
      return SLM_OK;
    }
  if (g_lane_n < 256) g_lane_tab[g_lane_n++] = e;
  else { g_lane_tab[g_lane_next] = e; g_lane_next = (g_lane_next + 1) % 256; }
  return SLM_OK;
}

SLM_API int slm_decode_lane_policy_clear(void) {
  std::lock_guard<std::mutex> lk(g_lane_mu);
  g_lane_n = g_lane_next = 0;
  return SLM_OK;
}

SLM_API int32_t slm_decode_lane_policy_measured(const slm_lane_query* q) {
  if (!q) return 0;
  std::lock_guard<std::mutex> lk(g_lane_mu);
  return lane_lookup(q) != nullptr ? 1 : 0;
}

}  // extern "C"
