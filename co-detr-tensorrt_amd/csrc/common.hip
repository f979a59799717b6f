// ABI bookkeeping for libcodetr_hip.so (include/codetr_hip.h).
#include <hip/hip_runtime.h>

#include "codetr_hip.h"

extern "C" {

int codetr_hip_abi_version(void) { return CODETR_HIP_ABI_VERSION; }

const char* codetr_hip_strerror(int code) {
  if (code == 0) return "success";
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  switch (code) {
    case CODETR_E_BADARG: return "null pointer or non-positive dimension";
    case CODETR_E_IM2COL_STEP: return "batch must divide im2col_step";
    case CODETR_E_TOO_LARGE: return "a per-image extent exceeds the kernel's 32-bit in-image offsets";
    case CODETR_E_UNSUPPORTED: return "shape outside what the kernel family implements";
  }
  return "unknown codetr_hip error";
}

}  // extern "C"
