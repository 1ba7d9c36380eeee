// Latent MLPs (solpex/latent.py:14-48: ParamToZ / ZToParam = Linear -> SiLU [-> LayerNorm] [-> Dropout] ... -> Linear) and
// their SmoothL1 training step (latent.py:51-310), fp32 as the reference.
//
// The layers are 5..256 wide at batch 64..256: every GEMM is a few MFLOP, so a step is bound by launch latency, not by
// MFMA or HBM.  The design goal is therefore FEW launches: one strided fp32 MFMA GEMM kernel serves all three products
// of a Linear layer (y = x W^T, dx = dy W, dW = dy^T x) through its stride arguments, and its epilogue carries what
// would otherwise be separate element-wise passes (bias, SiLU with the pre-activation saved, multiplication by SiLU'
// of the previous layer's pre-activation, the bias gradient as row sums of the A operand).
#include "sx_common.h"
#include <algorithm>

namespace sx {

struct GemmArgs {
  const float* A; int64_t a_sm, a_sk;      // A(m, k) = A[m * a_sm + k * a_sk]
  const float* B; int64_t b_sn, b_sk;      // B(n, k) = B[n * b_sn + k * b_sk]
  float* C; int64_t ldc;                   // C[m * ldc + n]
  int M, N, K;
  const float* bias;                       // [N] or null
  float* pre; int64_t ld_pre;              // act != 0: pre-activation saved here
  const float* dact; int64_t ld_dact;      // non-null: result multiplied by SiLU'(dact[m][n])
  const float* mul; int64_t ld_mul;        // non-null: result multiplied by mul[m][n] first (dropout mask)
  float* rowsum;                           // non-null: rowsum[m] (+)= sum_k A(m, k)   (blocks with blockIdx.x == 0)
  int act, accumulate;
};

// C = A B^T with v_mfma_f32_16x16x4_f32 (exact fp32).  Block = 4 waves = 32 x 32 outputs.  The reduction index may be
// visited in any order as long as A and B agree: a lane (n, g) feeds k = k0 + 4g + j in MFMA j, which makes its four
// loads of a K-contiguous operand one 16-byte run.
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * 32 + (wid >> 1) * 16, n0 = blockIdx.x * 32 + (wid & 1) * 16;
  const int am = m0 + n, bn = n0 + n;
  const bool am_ok = am < a.M, bn_ok = bn < a.N;
  const float* __restrict__ Ap = a.A + (int64_t)am * a.a_sm;
  const float* __restrict__ Bp = a.B + (int64_t)bn * a.b_sn;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float rs = 0.f;
  for (int k0 = 0; k0 < a.K; k0 += 32) {
    float av[2][4], bv[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + h * 16 + 4 * g + j;
        const bool kok = k < a.K;
        av[h][j] = (am_ok && kok) ? Ap[(int64_t)k * a.a_sk] : 0.f;
        bv[h][j] = (bn_ok && kok) ? Bp[(int64_t)k * a.b_sk] : 0.f;
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][j], bv[h][j], acc, 0, 0, 0);
        rs += av[h][j];
      }
  }
  if (a.rowsum && blockIdx.x == 0 && (wid & 1) == 0) {
    // lanes (n, g = 0..3) hold disjoint k subsets of row m0 + n
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    if (g == 0 && am_ok) a.rowsum[am] = a.accumulate ? a.rowsum[am] + rs : rs;
  }
  // lane (n, g) holds C[m0 + 4g + r][n0 + n]
  const int cn = n0 + n;
  if (cn >= a.N) return;
  const float bb = a.bias ? a.bias[cn] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int cm = m0 + 4 * g + r;
    if (cm >= a.M) continue;
    float v = acc[r] + bb;
    if (a.mul) v *= a.mul[(int64_t)cm * a.ld_mul + cn];
    if (a.dact) v *= dsiluf_(a.dact[(int64_t)cm * a.ld_dact + cn]);
    if (a.act) {
      a.pre[(int64_t)cm * a.ld_pre + cn] = v;
      v = siluf_(v);
    }
    float* dst = a.C + (int64_t)cm * a.ldc + cn;
    *dst = a.accumulate ? *dst + v : v;
  }
}

// nn.LayerNorm over the last dimension (latent.py:21-22), one wave per row.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int rows, int N,
                                                            float* __restrict__ y, float* __restrict__ stats) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * N;
  float s = 0.f;
  for (int i = lane; i < N; i += 64) s += xr[i];
  const float mean = wave_sum(s) / N;
  float q = 0.f;
  for (int i = lane; i < N; i += 64) { const float d = xr[i] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / N + eps);
  for (int i = lane; i < N; i += 64) y[(int64_t)row * N + i] = (xr[i] - mean) * rstd * gamma[i] + beta[i];
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; dgamma / dbeta accumulate through fp32 atomics.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ stats, const float* __restrict__ dy,
                                                            int rows, int N, float* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, const float* __restrict__ dact) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  const float* xr = x + (int64_t)row * N;
  const float* dr = dy + (int64_t)row * N;
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < N; i += 64) {
    const float gi = dr[i] * gamma[i], xh = (xr[i] - mean) * rstd;
    s1 += gi; s2 += gi * xh;
  }
  s1 = wave_sum(s1) / N; s2 = wave_sum(s2) / N;
  for (int i = lane; i < N; i += 64) {
    const float xh = (xr[i] - mean) * rstd;
    float v = rstd * (dr[i] * gamma[i] - s1 - xh * s2);
    if (dact) v *= dsiluf_(dact[(int64_t)row * N + i]);      // x = SiLU(pre): hand back the gradient of pre
    dx[(int64_t)row * N + i] = v;
    if (dgamma) { atomicAdd(dgamma + i, dr[i] * xh); atomicAdd(dbeta + i, dr[i]); }
  }
}

// nn.SmoothL1Loss(beta) with mean reduction (latent.py:74,148,229): out += weight * mean(huber(pred - tgt));
// dpred (+)= grad_weight / n * clamp((pred - tgt) / beta, -1, 1).
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, int64_t n,
                                                        float beta, float* __restrict__ out, float* __restrict__ dpred,
                                                        float grad_weight, int accumulate) {
  __shared__ float red[4];
  float s[1] = {0.f};
  const float inv_n = 1.0f / (float)n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = pred[i] - tgt[i], ad = fabsf(d);
    s[0] += ad < beta ? 0.5f * d * d / beta : ad - 0.5f * beta;
    if (dpred) {
      const float gq = (ad < beta ? d / beta : (d > 0.f ? 1.f : -1.f)) * grad_weight * inv_n;
      dpred[i] = accumulate ? dpred[i] + gq : gq;
    }
  }
  block_sum<1>(s, red);
  if (threadIdx.x == 0 && out) atomicAdd(out, s[0] * inv_n);
}

__global__ void mul_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ y, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = x[i] * m[i];
}

}  // namespace sx

using namespace sx;

extern "C" int sx_gemm_f32(int M, int N, int K, const float* A, int64_t a_sm, int64_t a_sk, const float* B, int64_t b_sn,
                           int64_t b_sk, float* Cm, int64_t ldc, const float* bias, int act, float* pre, int64_t ld_pre,
                           const float* mul, int64_t ld_mul, const float* dact, int64_t ld_dact, float* rowsum,
                           int accumulate, void* stream) {
  SX_REQUIRE(A && B && Cm, "sx_gemm_f32: null pointer");
  SX_REQUIRE(M > 0 && N > 0 && K > 0 && ldc >= N, "sx_gemm_f32: bad sizes M=%d N=%d K=%d ldc=%lld", M, N, K, (long long)ldc);
  SX_REQUIRE(!act || (pre && ld_pre >= N), "sx_gemm_f32: the SiLU epilogue needs a pre-activation buffer");
  SX_REQUIRE((!mul || ld_mul >= N) && (!dact || ld_dact >= N), "sx_gemm_f32: bad epilogue operand strides");
  SX_REQUIRE(cdiv(M, 32) <= 65535, "sx_gemm_f32: M too large");
  GemmArgs a;
  a.A = A; a.a_sm = a_sm; a.a_sk = a_sk; a.B = B; a.b_sn = b_sn; a.b_sk = b_sk; a.C = Cm; a.ldc = ldc;
  a.M = M; a.N = N; a.K = K; a.bias = bias; a.pre = pre; a.ld_pre = ld_pre; a.dact = dact; a.ld_dact = ld_dact;
  a.mul = mul; a.ld_mul = ld_mul; a.rowsum = rowsum; a.act = act; a.accumulate = accumulate;
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(cdiv(N, 32), cdiv(M, 32)), dim3(256), 0, (hipStream_t)stream, a);
  SX_LAUNCH_CHECK("gemm_f32");
  return SX_OK;
}

extern "C" int sx_layernorm_forward(int rows, int N, const float* x, const float* gamma, const float* beta, float eps, float* y,
                                    float* stats, void* stream) {
  SX_REQUIRE(x && gamma && beta && y && stats && rows > 0 && N > 0, "sx_layernorm_forward: bad arguments");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, rows, N, y, stats);
  SX_LAUNCH_CHECK("layernorm_fwd");
  return SX_OK;
}

extern "C" int sx_layernorm_backward(int rows, int N, const float* x, const float* gamma, const float* stats, const float* dy,
                                     float* dx, float* dgamma, float* dbeta, const float* dact, void* stream) {
  SX_REQUIRE(x && gamma && stats && dy && dx && rows > 0 && N > 0 && (!dgamma == !dbeta), "sx_layernorm_backward: bad arguments");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, stats, dy, rows, N, dx, dgamma, dbeta, dact);
  SX_LAUNCH_CHECK("layernorm_bwd");
  return SX_OK;
}

extern "C" int sx_smooth_l1(int64_t n, const float* pred, const float* target, float beta, float* out, float* dpred,
                            float grad_weight, int accumulate, void* stream) {
  SX_REQUIRE(pred && target && n > 0 && beta > 0.f, "sx_smooth_l1: bad arguments");
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(n, 1024), 256));
  hipLaunchKernelGGL(smooth_l1_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, n, beta, out, dpred, grad_weight, accumulate);
  SX_LAUNCH_CHECK("smooth_l1");
  return SX_OK;
}

extern "C" int sx_mul(int64_t n, const float* x, const float* m, float* y, void* stream) {
  SX_REQUIRE(x && m && y && n > 0, "sx_mul: bad arguments");
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(n, 1024), 1024));
  hipLaunchKernelGGL(mul_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, m, y, n);
  SX_LAUNCH_CHECK("mul");
  return SX_OK;
}
