// Run-time knobs of DIAGNOSTIC builds only (-DMSDA_ENC_ABLATE and friends, built by the scripts under tools/): the product
// library reads no environment variable, so that a deployment or an exported launch plan cannot change kernels with the
// environment.
#pragma once
#include <stdlib.h>
static inline int diag_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
