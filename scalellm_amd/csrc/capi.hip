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
    "SLM_W4_M128",          "SLM_W4_M128_WD",       "SLM_W4_M128_SPLITS",   "SLM_W4_SPLIT_TARGET",
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

}  // extern "C"
