// Output normalizers and masked error metrics on the device.
//   normalizer maps   data.py:96-110 (log), :139-143 (linear), :157-163,187-199 (symlog),
//                     :212-223 (per-channel dispatch)           -> sx_norm_apply
//   mae_norm          losses.py:226-227                          -> sx_masked_abs_err
//   batch_error_sums_ev losses.py:229-253                        -> sx_denorm_err_sums
// One fused pass each: the reference slices per channel, maps, concatenates and reduces separately.
#include "sx_common.h"

namespace sx {

struct NormTable {           // per channel
  int kind[SX_MAX_NORM_CH];
  float mu[SX_MAX_NORM_CH], sigma[SX_MAX_NORM_CH], eps[SX_MAX_NORM_CH], scale[SX_MAX_NORM_CH];
};

__device__ __forceinline__ float finite_or_zero(float v) { return (v == v && fabsf(v) != INFINITY) ? v : 0.f; }

// z -> physical units  (inverse), mask multiply as in the reference (linear kind ignores the mask)
__device__ __forceinline__ float norm_inverse_one(const NormTable& t, int c, float z, float m, bool has_mask) {
  const int kind = t.kind[c];
  if (kind == SX_NORM_LINEAR) return z * t.sigma[c] + t.mu[c];
  float y;
  if (kind == SX_NORM_LOG) y = expf(z * (t.sigma[c] + 1e-8f) + t.mu[c]) - t.eps[c];
  else {
    const float u = z * t.sigma[c] + t.mu[c];
    const float sg = u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f);
    y = sg * t.scale[c] * expm1f(fabsf(u));
  }
  y = finite_or_zero(y);
  return has_mask ? y * m : y;
}

__device__ __forceinline__ float norm_transform_one(const NormTable& t, int c, float y, float m) {
  const int kind = t.kind[c];
  if (kind == SX_NORM_LINEAR) return (y - t.mu[c]) / t.sigma[c];
  y = (y == y) ? y : 0.f;                       // nan_to_num: nan -> 0
  float z;
  if (kind == SX_NORM_LOG) {
    if (fabsf(y) == INFINITY) y = 0.f;          // data.py:97 maps +-inf -> 0
    z = (logf(fmaxf(y, 0.f) + t.eps[c]) - t.mu[c]) / (t.sigma[c] + 1e-8f);
  } else {
    if (y == INFINITY) y = 3.4028234663852886e38f;       // torch.nan_to_num default: +-inf -> +-FLT_MAX
    if (y == -INFINITY) y = -3.4028234663852886e38f;
    const float sg = y > 0.f ? 1.f : (y < 0.f ? -1.f : 0.f);
    z = (sg * log1pf(fabsf(y) / t.scale[c]) - t.mu[c]) / t.sigma[c];
  }
  return finite_or_zero(z) * m;
}

template <typename T>
__device__ __forceinline__ float ld_strided(const T* p, int64_t sb, int64_t sc, int64_t sh, int64_t sw, int b, int c, int y, int x) {
  return to_f32(p[b * sb + c * sc + y * sh + x * sw]);
}

// out (B,C,H,W) fp32 contiguous
template <typename T>
__global__ void norm_apply_kernel(NormTable t, int inverse, const T* __restrict__ in, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                                  const float* __restrict__ mask, float* __restrict__ out, int B, int C, int H, int W) {
  const int64_t n = (int64_t)B * C * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H); p /= H;
    const int c = (int)(p % C);
    const int b = (int)(p / C);
    const float v = ld_strided(in, sb, sc, sh, sw, b, c, y, x);
    const float m = mask ? mask[((int64_t)b * H + y) * W + x] : 1.f;
    out[i] = inverse ? norm_inverse_one(t, c, v, m, mask != nullptr) : norm_transform_one(t, c, v, m);
  }
}

// out[0] += sum |d|, out[1] += sum d^2, out[2] += sum mask (expanded over channels); d = (inv(p) - inv(t)) * m
// with denorm == 0: d = (p - t) * m (mae_norm numerator) and out[2] += sum mask over PIXELS only.
template <typename T>
__global__ __launch_bounds__(256) void err_sums_kernel(NormTable t, int denorm, const T* __restrict__ pred, int64_t sb, int64_t sc, int64_t sh,
                                                       int64_t sw, const float* __restrict__ target, const float* __restrict__ mask,
                                                       float* __restrict__ out, int B, int C, int H, int W) {
  float v[3] = {0.f, 0.f, 0.f};
  const int64_t n = (int64_t)B * C * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H); p /= H;
    const int c = (int)(p % C);
    const int b = (int)(p / C);
    const float m = mask[((int64_t)b * H + y) * W + x];
    float pv = ld_strided(pred, sb, sc, sh, sw, b, c, y, x), tv = target[i];
    if (denorm) {
      pv = finite_or_zero(norm_inverse_one(t, c, pv, m, true));
      tv = finite_or_zero(norm_inverse_one(t, c, tv, m, true));
    }
    const float d = (pv - tv) * m;
    v[0] += fabsf(d);
    v[1] += d * d;
    if (denorm || c == 0) v[2] += m;
  }
  __shared__ float red[12];
  block_sum<3>(v, red);
  if (threadIdx.x == 0) { atomicAdd(out + 0, v[0]); atomicAdd(out + 1, v[1]); atomicAdd(out + 2, v[2]); }
}

// Physical-space masked L1 term of the training loss (train.py:200-205) and its gradient:
//   loss_ev = sum_{b,c,px} |inv(p)*m - inv(y)*m| * m / max(sum_{b,px} m, 1e-8)          (inv = the normalizer's inverse)
//   out[0] += lam * loss_ev ;  d_pred[b,px,c] += lam * grad_scale * sign(.) * m * d inv(p)/dp / den
// One thread per pixel walks the channels: 16-byte NHWC accesses of pred / d_pred, planar target reads coalesced per channel.
// d inv/dp: log  exp(z) * (sigma + 1e-8) * m, z = p*(sigma+1e-8)+mu ; symlog  scale * exp(|u|) * sigma * m, u = p*sigma+mu ;
// linear  sigma ; zero where the inverse is not finite (the reference's torch.where passes no gradient there).
template <typename T>
__global__ __launch_bounds__(256) void ev_l1_kernel(NormTable t, const T* __restrict__ pred, int pred_ld, const float* __restrict__ target,
                                                    const float* __restrict__ mask, const float* __restrict__ den, float lam,
                                                    float grad_mul, float* __restrict__ out, T* __restrict__ d_pred, int dpred_ld,
                                                    int B, int C, int HW) {
  const float inv_den = 1.0f / fmaxf(den[0], 1e-8f);
  float v[1] = {0.f};
  const int64_t npx = (int64_t)B * HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npx; i += (int64_t)gridDim.x * blockDim.x) {
    const float m = mask[i];
    if (m == 0.f) continue;
    const int b = (int)(i / HW);
    const int64_t px = i - (int64_t)b * HW;
    for (int c = 0; c < C; ++c) {
      const float p = to_f32(pred[i * pred_ld + c]);
      const float y = target[((int64_t)b * C + c) * HW + px];
      const float pe = norm_inverse_one(t, c, p, m, true), ye = norm_inverse_one(t, c, y, m, true);
      const float d = pe - ye;
      v[0] += fabsf(d) * m;
      if (d_pred) {
        float dv;
        const int kind = t.kind[c];
        if (kind == SX_NORM_LINEAR) dv = t.sigma[c];
        else if (kind == SX_NORM_LOG) {
          const float e = expf(p * (t.sigma[c] + 1e-8f) + t.mu[c]);
          dv = finite_or_zero(e - t.eps[c]) == e - t.eps[c] ? e * (t.sigma[c] + 1e-8f) * m : 0.f;
        } else {
          const float u = p * t.sigma[c] + t.mu[c];
          const float yy = t.scale[c] * expm1f(fabsf(u));
          dv = (finite_or_zero(yy) == yy && u != 0.f) ? t.scale[c] * expf(fabsf(u)) * t.sigma[c] * m : (u == 0.f ? t.scale[c] * t.sigma[c] * m : 0.f);
        }
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float g = lam * grad_mul * inv_den * sg * m * dv;
        T* dst = d_pred + i * dpred_ld + c;
        *dst = from_f32<T>(to_f32(*dst) + g);
      }
    }
  }
  __shared__ float red[4];
  block_sum<1>(v, red);
  if (threadIdx.x == 0 && out) atomicAdd(out, lam * v[0] * inv_den);
}

static int fill_table(NormTable& t, int C, const int* kinds, const float* mu, const float* sigma, const float* eps, const float* scale, const char* who) {
  if (C < 1 || C > SX_MAX_NORM_CH) { set_error("%s: %d channels (max %d)", who, C, SX_MAX_NORM_CH); return SX_ERR_UNSUPPORTED; }
  memset(&t, 0, sizeof(t));
  for (int c = 0; c < C; ++c) {
    if (kinds) {
      if (kinds[c] < 0 || kinds[c] > 2) { set_error("%s: unknown normalizer kind %d", who, kinds[c]); return SX_ERR_INVALID; }
      t.kind[c] = kinds[c]; t.mu[c] = mu[c]; t.sigma[c] = sigma[c]; t.eps[c] = eps[c]; t.scale[c] = scale[c];
    }
  }
  return SX_OK;
}

static inline int gsz2(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 2048)); }

}  // namespace sx

using namespace sx;

// kinds/mu/sigma/eps/scale: HOST arrays of length C.
extern "C" int sx_norm_apply(int dtype, int inverse, int B, int C, int H, int W, const void* in, int64_t sb, int64_t sc,
                             int64_t sh, int64_t sw, const float* mask, const int* kinds, const float* mu, const float* sigma,
                             const float* eps, const float* scale, float* out, void* stream) {
  SX_REQUIRE(in && out && kinds && mu && sigma && eps && scale, "sx_norm_apply: null pointer");
  NormTable t;
  SX_TRY(fill_table(t, C, kinds, mu, sigma, eps, scale, "sx_norm_apply"));
  const int64_t n = (int64_t)B * C * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16) hipLaunchKernelGGL(norm_apply_kernel<bf16>, dim3(gsz2(n)), dim3(256), 0, st, t, inverse, (const bf16*)in, sb, sc, sh, sw, mask, out, B, C, H, W);
  else hipLaunchKernelGGL(norm_apply_kernel<float>, dim3(gsz2(n)), dim3(256), 0, st, t, inverse, (const float*)in, sb, sc, sh, sw, mask, out, B, C, H, W);
  SX_LAUNCH_CHECK("norm_apply");
  return SX_OK;
}

// denorm != 0: batch_error_sums_ev (kinds... required); denorm == 0: mae_norm sums.  out[3] is accumulated (caller zeroes).
extern "C" int sx_masked_err_sums(int dtype, int denorm, int B, int C, int H, int W, const void* pred, int64_t sb, int64_t sc,
                                  int64_t sh, int64_t sw, const float* target, const float* mask, const int* kinds, const float* mu,
                                  const float* sigma, const float* eps, const float* scale, float* out, void* stream) {
  SX_REQUIRE(pred && target && mask && out, "sx_masked_err_sums: null pointer");
  SX_REQUIRE(!denorm || (kinds && mu && sigma && eps && scale), "sx_masked_err_sums: normalizer table missing");
  NormTable t;
  SX_TRY(fill_table(t, denorm ? C : 1, denorm ? kinds : nullptr, mu, sigma, eps, scale, "sx_masked_err_sums"));
  const int64_t n = (int64_t)B * C * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16) hipLaunchKernelGGL(err_sums_kernel<bf16>, dim3(gsz2(n)), dim3(256), 0, st, t, denorm, (const bf16*)pred, sb, sc, sh, sw, target, mask, out, B, C, H, W);
  else hipLaunchKernelGGL(err_sums_kernel<float>, dim3(gsz2(n)), dim3(256), 0, st, t, denorm, (const float*)pred, sb, sc, sh, sw, target, mask, out, B, C, H, W);
  SX_LAUNCH_CHECK("masked_err_sums");
  return SX_OK;
}

// lam_ev term of train_unet's loss (train.py:200-205).  den: DEVICE pointer to sum(mask) over the (global) batch; out[0] (nullable)
// += lam * loss_ev; d_pred (nullable, NHWC `dtype`) += lam * grad_scale * d loss_ev / d pred (call after sx_loss_forward_backward).
extern "C" int sx_loss_ev_l1(int dtype, int B, int C, int H, int W, const void* pred, int pred_ld, const float* target,
                             const float* mask, const int* kinds, const float* mu, const float* sigma, const float* eps,
                             const float* scale, const float* den, float lam, float* out, void* d_pred, int dpred_ld,
                             float grad_scale, void* stream) {
  SX_REQUIRE(pred && target && mask && den && kinds && mu && sigma && eps && scale, "sx_loss_ev_l1: null pointer");
  SX_REQUIRE(dtype == SX_F32 || dtype == SX_BF16, "sx_loss_ev_l1: bad dtype %d", dtype);
  NormTable t;
  SX_TRY(fill_table(t, C, kinds, mu, sigma, eps, scale, "sx_loss_ev_l1"));
  const int64_t npx = (int64_t)B * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16) hipLaunchKernelGGL(ev_l1_kernel<bf16>, dim3(gsz2(npx)), dim3(256), 0, st, t, (const bf16*)pred, pred_ld, target, mask, den, lam, grad_scale, out, (bf16*)d_pred, dpred_ld, B, C, H * W);
  else hipLaunchKernelGGL(ev_l1_kernel<float>, dim3(gsz2(npx)), dim3(256), 0, st, t, (const float*)pred, pred_ld, target, mask, den, lam, grad_scale, out, (float*)d_pred, dpred_ld, B, C, H * W);
  SX_LAUNCH_CHECK("loss_ev_l1");
  return SX_OK;
}
