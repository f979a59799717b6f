/* Plain-C consumer of include/codetr_hip.h (no Python, no torch): what a cgo / JNI / TensorRT-plugin style caller
 * links against.  Host-only part: version, error strings, dispatch query, argument validation -- runs without a GPU. */
#include <stdio.h>
#include <string.h>

#include "codetr_hip.h"

int main(void) {
  if (codetr_hip_abi_version() != CODETR_HIP_ABI_VERSION) {
    fprintf(stderr, "library ABI %d, header ABI %d\n", codetr_hip_abi_version(), CODETR_HIP_ABI_VERSION);
    return 1;
  }
  if (strcmp(codetr_hip_strerror(0), "success") != 0) return 2;
  if (strcmp(codetr_msda_variant(2, 8, 32, 5, 4), "tiled_x4") != 0) return 3;
  /* contract errors are detected on the host, before any launch */
  char dummy[16];
  if (codetr_msda_forward_f16(NULL, NULL, (const int64_t *)dummy, (const int64_t *)dummy, dummy, dummy, 1, 1, 1, 8, 1,
                              1, 1, 64, dummy) != CODETR_E_BADARG)
    return 4;
  if (codetr_msda_forward_f32(NULL, dummy, (const int64_t *)dummy, (const int64_t *)dummy, dummy, dummy, 3, 1, 1, 8, 1,
                              1, 1, 2, dummy) != CODETR_E_IM2COL_STEP)
    return 5;
  printf("abi %d ok\n", codetr_hip_abi_version());
  return 0;
}
