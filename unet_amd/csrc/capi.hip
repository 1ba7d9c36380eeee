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

// fp32 arithmetic of the convolution kernels, process-wide: bf16 planes of the split form (2: three bf16 MFMAs per product block,
// 3: six) or 0 = exact fp32 MFMA
static std::atomic<int> g_fp32_planes{-1};
static thread_local int t_fp32_wide = 0;       // inside a network-level call on a wide network: 1 = exact kernels, 3 = three planes
Fp32WideScope::Fp32WideScope(int mode) : prev_(t_fp32_wide) { if (mode) t_fp32_wide = mode; }
Fp32WideScope::~Fp32WideScope() { t_fp32_wide = prev_; }
static int fp32_planes_process() {
  int v = g_fp32_planes.load(std::memory_order_relaxed);
  if (v < 0) {
    v = env_int("SX_FP32_EXACT", 0) ? 0 : (env_int("SX_FP32_PLANES", 2) == 3 ? 3 : 2);
    g_fp32_planes.store(v, std::memory_order_relaxed);
  }
  return v;
}
int fp32_split_process() { return fp32_planes_process() != 0; }
int fp32_planes() {
  const int v = fp32_planes_process();
  if (v == 0 || t_fp32_wide == 1) return 0;
  return (v == 3 || t_fp32_wide == 3) ? 3 : 2;
}
int fp32_split() { return fp32_planes() != 0; }
}  // namespace sx

extern "C" int sx_set_fp32_planes(int planes) {
  const int was = sx::fp32_planes_process();
  if (planes == 0 || planes == 2 || planes == 3) sx::g_fp32_planes.store(planes, std::memory_order_relaxed);      // anything else: query only
  return was;
}
extern "C" int sx_set_fp32_exact(int on) {
  const int was = sx::fp32_planes_process() == 0 ? 1 : 0;
  if (on >= 0) sx::g_fp32_planes.store(on ? 0 : 2, std::memory_order_relaxed);      // on < 0: query only
  return was;
}
extern "C" int sx_version(void) { return SX_VERSION; }
extern "C" const char* sx_last_error(void) { return sx::g_err; }
