// Error plumbing and version of libsolpex_hip.so.
#include "sx_common.h"
#include <stdarg.h>

namespace sx {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
  return SX_ERR_HIP;
}
}  // namespace sx

extern "C" int sx_version(void) { return SX_VERSION; }
extern "C" const char* sx_last_error(void) { return sx::g_err; }
