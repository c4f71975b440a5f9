// capi.hip -- status strings / version of the C ABI (include/slm_hip.h).
#include "common.h"

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

SLM_API const char* slm_version(void) { return "slm_hip 0.1.0 (gfx950)"; }

}  // extern "C"
