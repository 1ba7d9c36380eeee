// Error plumbing and version of libsolpex_hip.so.
#include "sx_common.h"
#include <stdarg.h>
#include <stdlib.h>

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

int ensure_max_lds(const void* kern, std::atomic<unsigned long long>& done, const char* what) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return SX_OK;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return hip_fail(e, what);
  done.fetch_or(bit, std::memory_order_release);
  return SX_OK;
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
}  // namespace sx

extern "C" int sx_version(void) { return SX_VERSION; }
extern "C" const char* sx_last_error(void) { return sx::g_err; }
