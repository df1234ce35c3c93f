// Version / error plumbing of the C ABI.
#include "sfm_common.h"

#include <cstring>

namespace sfm {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace sfm

extern "C" {

int sfm_version(void) { return SFM_ABI_VERSION; }

const char* sfm_last_error(void) { return sfm::error_buffer(); }

int sfm_device_count(int* count) {
  if (!count) return sfm::fail(SFM_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return SFM_OK;
}

}  // extern "C"
