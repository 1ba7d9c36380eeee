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

// fp32 arithmetic of the convolution kernels: 1 = split-bf16 (three bf16 MFMAs per product block), 0 = exact fp32 MFMA
static std::atomic<int> g_fp32_split{-1};
static thread_local int t_fp32_exact_scope = 0;       // > 0: inside a network-level call that asked for exact arithmetic
Fp32ExactScope::Fp32ExactScope(bool on) : on_(on) { if (on_) ++t_fp32_exact_scope; }
Fp32ExactScope::~Fp32ExactScope() { if (on_) --t_fp32_exact_scope; }
int fp32_split() {
  if (t_fp32_exact_scope > 0) return 0;
  return fp32_split_process();
}
int fp32_split_process() {
  int v = g_fp32_split.load(std::memory_order_relaxed);
  if (v < 0) {
    v = env_int("SX_FP32_EXACT", 0) ? 0 : 1;
    g_fp32_split.store(v, std::memory_order_relaxed);
  }
  return v;
}
}  // namespace sx

extern "C" int sx_set_fp32_exact(int on) {
  const int was = sx::fp32_split() ? 0 : 1;
  if (on >= 0) sx::g_fp32_split.store(on ? 0 : 1, std::memory_order_relaxed);      // on < 0: query only
  return was;
}
extern "C" int sx_version(void) { return SX_VERSION; }
extern "C" const char* sx_last_error(void) { return sx::g_err; }
