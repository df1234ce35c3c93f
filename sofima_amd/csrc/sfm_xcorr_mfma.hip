// int8 MFMA Toeplitz correlation (placeholder until the kernel lands).
#include "sfm_common.h"

namespace sfm {
bool mfma_i8_eligible(const SfmXcorrDesc*) { return false; }
size_t mfma_i8_workspace_bytes(const SfmXcorrDesc*) { return 0; }
int mfma_i8_surface(const SfmXcorrDesc*, void*, float*) {
  return fail(SFM_ERR_INVALID, "MFMA_I8 path not available");
}
}  // namespace sfm
