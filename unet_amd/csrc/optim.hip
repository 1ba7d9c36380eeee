// Optimizer step of the training loop (train.py:82,226-233): global gradient norm,
// clip_grad_norm_ (scale = min(1, max_norm / (norm + 1e-6))) and decoupled-weight-decay Adam
// (torch.optim.AdamW, amsgrad off), fused over ONE flat fp32 parameter arena that is also the
// gradient all-reduce bucket.  28 B of HBM traffic per parameter (read p,g,m,v; write p,m,v).
#include "sx_common.h"

namespace sx {

// Block partials of the gradient norm live in CALLER-OWNED memory (`parts`, SX_SUMSQ_PARTS floats: the library keeps no mutable
// device state, so steps that overlap on one GPU -- two engines, two streams -- cannot clip with each other's norm).  The norm
// must come out bit-identical on every data-parallel rank -- it sets the clip factor, and ranks whose clip factors differ in
// the last bit drift apart -- so the partials are summed in a fixed order instead of racing on an atomic.

// fixed-order sum of the block partials; every thread of the calling block gets the total (same bits in every block)
__device__ __forceinline__ float sumsq_total(const float* __restrict__ parts, int nparts) {
  float v[1] = {(int)threadIdx.x < nparts ? parts[threadIdx.x] : 0.f};
  __shared__ float red[4];
  __shared__ float tot;
  block_sum<1>(v, red);
  if (threadIdx.x == 0) tot = v[0];
  __syncthreads();
  return tot;
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ parts, int nparts, float* __restrict__ out) {
  const float t = sumsq_total(parts, nparts);
  if (threadIdx.x == 0) out[0] += t;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ parts) {
  float v[1] = {0.f};
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(g + i * 4);
    v[0] += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float x = g[n4 * 4 + threadIdx.x]; v[0] += x * x; }
  __shared__ float red[4];
  block_sum<1>(v, red);
  if (threadIdx.x == 0) parts[blockIdx.x] = v[0];
}

struct AdamArgs {
  float* p; const float* g; float* m; float* v; int64_t n;
  float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;
  float max_norm; const float* gnorm_sq; float grad_mul;
  float* gnorm_out;
  float* clear_after;
  const float* parts; int fused_parts;      // > 0: the norm comes from these block partials of sx_grad_sumsq_partial (no gnorm_sq)
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float gmul) {
  g *= gmul;
  p *= 1.0f - a.lr * a.wd;
  m = a.beta1 * m + (1.0f - a.beta1) * g;
  v = a.beta2 * v + (1.0f - a.beta2) * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p -= (a.lr / a.bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  float gmul = a.grad_mul;
  if (a.gnorm_sq || a.fused_parts > 0) {
    // fused: every block adds the partials itself, in the same fixed order -> the same clip factor everywhere, and no
    // one-block launch in between
    const float ss = a.fused_parts > 0 ? sumsq_total(a.parts, a.fused_parts) : a.gnorm_sq[0];
    const float norm = sqrtf(ss) * fabsf(a.grad_mul);
    if (a.max_norm > 0.f) gmul *= fminf(1.0f, a.max_norm / (norm + 1e-6f));
    if (a.gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) a.gnorm_out[0] = norm;
  }
  if (a.clear_after && blockIdx.x == 0 && threadIdx.x == 0) a.clear_after[0] = 0.f;
  const int64_t n4 = a.n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 p = *reinterpret_cast<f32x4*>(a.p + i * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.g + i * 4);
    f32x4 m = *reinterpret_cast<f32x4*>(a.m + i * 4);
    f32x4 v = *reinterpret_cast<f32x4*>(a.v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { float pe = p[e], me = m[e], ve = v[e]; adam_one(pe, g[e], me, ve, a, gmul); p[e] = pe; m[e] = me; v[e] = ve; }
    *reinterpret_cast<f32x4*>(a.p + i * 4) = p;
    *reinterpret_cast<f32x4*>(a.m + i * 4) = m;
    *reinterpret_cast<f32x4*>(a.v + i * 4) = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    adam_one(a.p[i], a.g[i], a.m[i], a.v[i], a, gmul);
  }
}

}  // namespace sx

using namespace sx;

// Deterministic: fixed grid, fixed-order sums.
static int sumsq_blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(n, 1024), SX_SUMSQ_PARTS)); }

// Block partials of sum g^2 into parts[0 .. count) (caller-owned, SX_SUMSQ_PARTS floats); returns count (>= 1) or a negative
// status.  sx_adamw_step_fused(parts, count) adds them up itself.
extern "C" int sx_grad_sumsq_partial(const float* g, int64_t n, float* parts, void* stream) {
  SX_REQUIRE(g && parts && n > 0 && ((uintptr_t)g % 16) == 0, "sx_grad_sumsq_partial: bad arguments");
  const int blocks = sumsq_blocks(n);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, parts);
  SX_LAUNCH_CHECK("grad_sumsq");
  return blocks;
}

// out[0] += sum g^2 (caller zeroes out); parts: SX_SUMSQ_PARTS floats of caller-owned scratch.
extern "C" int sx_grad_sumsq(const float* g, int64_t n, float* out, float* parts, void* stream) {
  SX_REQUIRE(g && out && parts && n >= 0 && ((uintptr_t)g % 16) == 0, "sx_grad_sumsq: bad arguments");
  if (n == 0) return SX_OK;
  const int blocks = sumsq_blocks(n);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, parts);
  SX_LAUNCH_CHECK("grad_sumsq");
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)parts, blocks, out);
  SX_LAUNCH_CHECK("grad_sumsq_final");
  return SX_OK;
}

static int adamw_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float max_norm, const float* gnorm_sq, const float* parts, int nparts,
                        float grad_mul, float* gnorm_out, float* clear_after, void* stream, const char* who) {
  SX_REQUIRE(p && g && m && v && n >= 0 && step >= 1, "%s: bad arguments", who);
  SX_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
             "%s: arenas must be 16-byte aligned", who);
  if (n == 0) return SX_OK;
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.max_norm = max_norm; a.gnorm_sq = gnorm_sq; a.grad_mul = grad_mul; a.gnorm_out = gnorm_out; a.clear_after = clear_after;
  a.parts = parts; a.fused_parts = nparts;
  SX_REQUIRE(!clear_after || clear_after != gnorm_sq, "%s: clear_after must not alias gnorm_sq", who);
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(n, 1024), 2048));
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  SX_LAUNCH_CHECK("adamw_step");
  return SX_OK;
}

// One AdamW step over a flat arena.  step is the 1-based step count.  If gnorm_sq != NULL the
// gradients are first scaled by min(1, max_norm / (sqrt(gnorm_sq) * |grad_mul| + 1e-6)) (max_norm <= 0:
// no clipping); gnorm_sq == NULL: no clipping, whatever max_norm says.  grad_mul is a plain multiplier (e.g. 1/world_size,
// or 1/loss_scale).  gnorm_out (nullable) receives the pre-clip norm.  clear_after (nullable, must not alias gnorm_sq) is
// set to 0 by the kernel: callers that alternate between two sum-of-squares slots get the next step's slot zeroed for free.
extern "C" int sx_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float max_norm,
                             const float* gnorm_sq, float grad_mul, float* gnorm_out, float* clear_after, void* stream) {
  return adamw_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, max_norm, gnorm_sq, nullptr, 0, grad_mul, gnorm_out,
                      clear_after, stream, "sx_adamw_step");
}

// The same step with clip_grad_norm_ taken from the block partials of sx_grad_sumsq_partial(g, n, parts) (same stream, just
// before): every block adds them up itself, in the same fixed order -- clip + AdamW in two launches instead of three.
// nparts must be what that call returned for this n.
extern "C" int sx_adamw_step_fused(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, int step, float max_norm,
                                   const float* parts, int nparts, float grad_mul, float* gnorm_out, void* stream) {
  SX_REQUIRE(parts && n > 0 && nparts == sumsq_blocks(n), "sx_adamw_step_fused: %d partials do not belong to %lld values (expected %d)",
             nparts, (long long)n, n > 0 ? sumsq_blocks(n) : 0);
  return adamw_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, max_norm, nullptr, parts, nparts, grad_mul, gnorm_out,
                      nullptr, stream, "sx_adamw_step_fused");
}
