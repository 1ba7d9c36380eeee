// Common device/host helpers for the SOLPEx conditional U-Net HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>

#include "../../include/solpex_hip.h"

namespace sx {

// ------------------------------------------------------------------ error plumbing
void set_error(const char* fmt, ...);
int  hip_fail(hipError_t e, const char* what);

#define SX_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      ::sx::set_error(__VA_ARGS__);           \
      return SX_ERR_INVALID;                  \
    }                                         \
  } while (0)

#define SX_LAUNCH_CHECK(what)                                   \
  do {                                                          \
    hipError_t e__ = hipGetLastError();                         \
    if (e__ != hipSuccess) return ::sx::hip_fail(e__, what);    \
  } while (0)

#define SX_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != SX_OK) return rc__; \
  } while (0)

// ------------------------------------------------------------------ element types
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float  f32x4;
typedef __attribute__((ext_vector_type(2))) float  f32x2;
typedef __attribute__((ext_vector_type(4))) short  s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kDtype = SX_F32;
  static constexpr int EPL = 4;          // elements per 16-byte piece
  static constexpr int CK  = 16;         // channels per 64-byte K chunk
  typedef f32x4 Vec;                     // one 16-byte piece
};
template <> struct Elem<bf16> {
  static constexpr int kDtype = SX_BF16;
  static constexpr int EPL = 8;
  static constexpr int CK  = 32;
  typedef bf16x8 Vec;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16  from_f32<bf16>(float v) { return (bf16)v; }   // v_cvt_pk_bf16_f32 (RNE, NaN-safe)

// 4 consecutive elements <-> 4 floats
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16* p, f32x4 v) {
  bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *reinterpret_cast<bf16x4*>(p) = o;
}

// 8 consecutive elements <-> 8 floats (the elementwise kernels' unit of work)
struct f32x8 { float v[8]; };
__device__ __forceinline__ f32x8 load8(const float* p) {
  f32x8 r;
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  r.v[0] = a[0]; r.v[1] = a[1]; r.v[2] = a[2]; r.v[3] = a[3];
  r.v[4] = b[0]; r.v[5] = b[1]; r.v[6] = b[2]; r.v[7] = b[3];
  return r;
}
__device__ __forceinline__ f32x8 load8(const bf16* p) {
  f32x8 r;
  bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (float)a[i];
  return r;
}
__device__ __forceinline__ void store8(float* p, const f32x8& r) {
  f32x4 a = {r.v[0], r.v[1], r.v[2], r.v[3]}, b = {r.v[4], r.v[5], r.v[6], r.v[7]};
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16* p, const f32x8& r) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16)r.v[i];
  *reinterpret_cast<bf16x8*>(p) = a;
}

// ------------------------------------------------------------------ math
// v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE division here costs ~10 extra VALU instructions per element, and the
// element-wise GroupNorm / SiLU passes sit close to their VALU floor (measured: see DESIGN.md, "element-wise passes")
__device__ __forceinline__ float sigmoidf_(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896341f)); }
__device__ __forceinline__ float siluf_(float z) { return z * sigmoidf_(z); }
__device__ __forceinline__ float dsiluf_(float z) {
  float s = sigmoidf_(z);
  return s * (1.0f + z * (1.0f - s));
}

// ------------------------------------------------------------------ wave / block reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 16 lanes that share (lane >> 4)
__device__ __forceinline__ float sum16(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum of NV values; result valid in thread 0.  `red` needs NV * (blockDim/64) floats.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wid * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += red[w * NV + i];
      v[i] = s;
    }
  }
  __syncthreads();
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB) once per (kernel, device): `done` is a per-kernel bit mask of
// device ordinals (a function-local static at the call site); racing first calls are harmless (the call is idempotent)
int ensure_max_lds(const void* kern, std::atomic<unsigned long long>& done, const char* what);
// integer tuning knob from the environment (callers keep the result in a function-local static: read once per process)
int env_int(const char* name, int dflt);

// 1: fp32 convolutions and weight gradients run as three bf16 MFMAs on split operands (default; conv_x3.hip); 0: exact
// fp32 MFMA (SX_FP32_EXACT=1 or sx_set_fp32_exact).  The packed fp32 weights differ between the two: repack after a switch.
int fp32_split();
// the process-wide mode alone, whatever scope is active: the WEIGHT GRADIENTS follow it (measured at BASELINE configs[4], full
// grid, tools/fp32_mix.py: the split arithmetic's error comes from the forward / data-gradient convolutions alone --
// convolutions split 3.95e-4 with exact or split weight gradients, convolutions exact 3.31e-5 with either)
int fp32_split_process();
// bf16 planes of the split arithmetic in force: 0 = exact fp32 MFMA, 2 = hi + lo (three MFMAs per block), 3 = hi + mid + lo
// (six MFMAs, fp32-grade products).  fp32_split() == (fp32_planes() != 0).
int fp32_planes();
// While alive on this thread the convolutions of wide networks leave the two-plane kernels: mode 3 = three planes, 1 = exact,
// 0 = no change.  The whole-network entry points (unet_plan.hip) use it for architectures where two planes were measured
// outside the 1e-4 contract.
struct Fp32WideScope {
  explicit Fp32WideScope(int mode);
  ~Fp32WideScope();
  Fp32WideScope(const Fp32WideScope&) = delete;
  Fp32WideScope& operator=(const Fp32WideScope&) = delete;
 private:
  int prev_;
};

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
static inline size_t dtype_size(int dtype) { return dtype == SX_BF16 ? 2 : 4; }

}  // namespace sx
