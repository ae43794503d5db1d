// Error plumbing + version for the C ABI (include/rag_amd.h).
#include "common.h"

namespace ragmi {

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

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RAGMI_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return RAGMI_OK;
}

}  // namespace ragmi

extern "C" int ragmi_version(void) { return 100; }
extern "C" const char* ragmi_last_error(void) { return ragmi::error_buffer(); }
