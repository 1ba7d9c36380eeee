// Batched inverse problems through the frozen U-Net (scripts/eval_inverse_cycle_conditional_unet.py:105-115, 332-411):
// N independent "recover params from fields" optimisations run as ONE batch, so every quantity the reference computes
// per case (B = 1) is per SAMPLE here: the weighted masked MSE and its denominator, the L2 regulariser, the Adam state,
// the cosine learning-rate step, the clamp and the early-stop bookkeeping.
#include "sx_common.h"

namespace sx {

// fit[b] = sum_{c,px} (pred - target)^2 m w_c / max(sum_px m * sum_c w_c, 1e-8)          (weighted_masked_mse, :105-115)
// mae[b] = sum_{c in sel,px} |pred - target| m / max(sum_px m * n_sel, 1e-8)             (masked_mae, :125-129)
// d_pred[b,px,c] = grad_scale * 2 (pred - target) m w_c / den_b  (optional; padding channels are zeroed)
// One block per sample: pass 1 sums the mask, pass 2 does the rest.
template <typename T>
__global__ __launch_bounds__(256) void inverse_fit_kernel(const T* __restrict__ pred, int ld, const float* __restrict__ target,
                                                          const float* __restrict__ mask, const float* __restrict__ wc,
                                                          const float* __restrict__ sel, int C, int HW,
                                                          float* __restrict__ fit, float* __restrict__ mae,
                                                          T* __restrict__ dpred, int dld, float gscale) {
  __shared__ float red[16];
  __shared__ float w_s[64], s_s[64];
  const int b = blockIdx.x;
  if (threadIdx.x < 64) {
    w_s[threadIdx.x] = threadIdx.x < C ? wc[threadIdx.x] : 0.f;
    s_s[threadIdx.x] = threadIdx.x < C ? (sel ? sel[threadIdx.x] : 1.f) : 0.f;
  }
  const float* mb = mask + (int64_t)b * HW;
  float v[1] = {0.f};
  for (int px = threadIdx.x; px < HW; px += blockDim.x) v[0] += mb[px];
  block_sum<1>(v, red);                      // ends with a barrier; result in thread 0
  __shared__ float msum_s;
  if (threadIdx.x == 0) msum_s = v[0];
  __syncthreads();
  float wsum = 0.f, nsel = 0.f;
  for (int c = 0; c < C; ++c) { wsum += w_s[c]; nsel += s_s[c]; }
  const float den = fmaxf(msum_s * wsum, 1e-8f);
  const float den_mae = fmaxf(msum_s * nsel, 1e-8f);
  const float gs = gscale * 2.0f / den;
  float acc[2] = {0.f, 0.f};
  for (int px = threadIdx.x; px < HW; px += blockDim.x) {
    const float m = mb[px];
    const T* pp = pred + ((int64_t)b * HW + px) * ld;
    T* dp = dpred ? dpred + ((int64_t)b * HW + px) * dld : nullptr;
    for (int c = 0; c < C; ++c) {
      const float d = to_f32(pp[c]) - target[((int64_t)b * C + c) * HW + px];
      acc[0] += d * d * m * w_s[c];
      acc[1] += fabsf(d) * m * s_s[c];
      if (dp) dp[c] = from_f32<T>(gs * d * m * w_s[c]);
    }
    if (dp) for (int c = C; c < dld; ++c) dp[c] = from_f32<T>(0.f);
  }
  block_sum<2>(acc, red);
  if (threadIdx.x == 0) {
    fit[b] = acc[0] / den;
    if (mae) mae[b] = acc[1] / den_mae;
  }
}

struct InvAdamArgs {
  float* p; const float* g; float* m; float* v;    // [N][P]; g = d fit / d p from the FiLM backward
  const float* fit;                                 // [N]
  float* best; int* wait; int* active; float* loss_out;
  int N, P;
  float lr, beta1, beta2, eps, bc1, bc2_sqrt, reg, clip, es_tol;
  int es_patience;
};

// One optimisation step of every ACTIVE sample, in the reference's order (:375-411): loss = fit + reg*mean(p^2) at the
// current p -> Adam step -> clamp -> early-stop test on that loss.  rel_improv is evaluated exactly as written there:
// while best is still +inf it is inf/inf = NaN, which never beats the tolerance -- so, like the reference, a run with
// patience > 0 stops after `patience` steps.
__global__ void inverse_adam_kernel(InvAdamArgs a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.N || !a.active[n]) return;
  float* p = a.p + (int64_t)n * a.P;
  float sq = 0.f;
  for (int j = 0; j < a.P; ++j) sq += p[j] * p[j];
  const float loss = a.fit[n] + a.reg * (sq / (float)a.P);
  if (a.loss_out) a.loss_out[n] = loss;
  for (int j = 0; j < a.P; ++j) {
    const int64_t i = (int64_t)n * a.P + j;
    const float g = a.g[i] + a.reg * 2.0f * p[j] / (float)a.P;
    const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * g * g;
    a.m[i] = m; a.v[i] = v;
    float pj = p[j] - (a.lr / a.bc1) * (m / (sqrtf(v) / a.bc2_sqrt + a.eps));
    if (a.clip > 0.f) pj = fminf(fmaxf(pj, -a.clip), a.clip);
    p[j] = pj;
  }
  if (a.es_patience > 0) {
    const float best = a.best[n];
    const float rel = (best - loss) / fmaxf(fabsf(best), 1e-12f);
    if (rel > a.es_tol) { a.best[n] = loss; a.wait[n] = 0; }
    else if (++a.wait[n] >= a.es_patience) a.active[n] = 0;
  }
}

}  // namespace sx

using namespace sx;

extern "C" int sx_inverse_fit(int dtype, int B, int C, int H, int W, const void* pred, int pred_ld, const float* target,
                              const float* mask, const float* channel_weights, const float* selected, float* fit,
                              float* mae, void* d_pred, int dpred_ld, float grad_scale, void* stream) {
  SX_REQUIRE(pred && target && mask && channel_weights && fit, "sx_inverse_fit: null pointer");
  SX_REQUIRE(dtype == SX_F32 || dtype == SX_BF16, "sx_inverse_fit: bad dtype %d", dtype);
  SX_REQUIRE(B > 0 && C > 0 && C <= 64 && H > 0 && W > 0 && pred_ld >= C && (!d_pred || dpred_ld >= C), "sx_inverse_fit: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16)
    hipLaunchKernelGGL(inverse_fit_kernel<bf16>, dim3(B), dim3(256), 0, st, (const bf16*)pred, pred_ld, target, mask, channel_weights,
                       selected, C, H * W, fit, mae, (bf16*)d_pred, dpred_ld, grad_scale);
  else
    hipLaunchKernelGGL(inverse_fit_kernel<float>, dim3(B), dim3(256), 0, st, (const float*)pred, pred_ld, target, mask, channel_weights,
                       selected, C, H * W, fit, mae, (float*)d_pred, dpred_ld, grad_scale);
  SX_LAUNCH_CHECK("inverse_fit");
  return SX_OK;
}

extern "C" int sx_inverse_adam_step(int N, int P, float* p, const float* grad, float* m, float* v, const float* fit, float lr,
                                    float beta1, float beta2, float eps, int step, float reg, float clip,
                                    int es_patience, float es_tol, float* best_loss, int* wait, int* active, float* loss_out,
                                    void* stream) {
  SX_REQUIRE(p && grad && m && v && fit && active && N > 0 && P > 0 && step >= 1, "sx_inverse_adam_step: bad arguments");
  SX_REQUIRE(es_patience <= 0 || (best_loss && wait), "sx_inverse_adam_step: early stopping needs best_loss and wait");
  InvAdamArgs a;
  a.p = p; a.g = grad; a.m = m; a.v = v; a.fit = fit; a.best = best_loss; a.wait = wait; a.active = active; a.loss_out = loss_out;
  a.N = N; a.P = P; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.reg = reg; a.clip = clip; a.es_tol = es_tol; a.es_patience = es_patience;
  hipLaunchKernelGGL(inverse_adam_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, a);
  SX_LAUNCH_CHECK("inverse_adam_step");
  return SX_OK;
}
