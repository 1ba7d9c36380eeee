// FiLM generators (models.py:38-58): [gamma|beta] = W2 * SiLU(W1 p + b1) + b2, returned as
// (1 + gamma, beta) per (sample, channel).  All generators of the network run in ONE launch
// (grid = samples x generators).  Tiny GEMVs: <0.01 % of the FLOPs, so plain FMA code.
#include "sx_common.h"
#include "film.h"

namespace sx {

__global__ __launch_bounds__(256) void film_fwd_kernel(FilmArgs a) {
  __shared__ float hid[FILM_MAX_HID];
  __shared__ float pv[64];
  const int b = blockIdx.x;
  const FilmGenDesc& d = a.g[blockIdx.y];
  for (int p = threadIdx.x; p < a.P; p += blockDim.x) pv[p] = a.params[(int64_t)b * a.P + p];
  __syncthreads();
  for (int j = threadIdx.x; j < a.Hf; j += blockDim.x) {
    float s = d.b1[j];
    for (int p = 0; p < a.P; ++p) s += d.w1[j * a.P + p] * pv[p];
    if (d.pre) d.pre[(int64_t)b * a.Hf + j] = s;
    hid[j] = siluf_(s);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * d.C; k += blockDim.x) {
    float s = d.b2[k];
    const float* w = d.w2 + (int64_t)k * a.Hf;
    for (int j = 0; j < a.Hf; ++j) s += w[j] * hid[j];
    if (k < d.C) d.gf[(int64_t)b * d.C + k] = 1.0f + s;
    else d.bf[(int64_t)b * d.C + (k - d.C)] = s;
  }
}

// phase A (grid = B x ngen): dpre[b][j] = (sum_k dout[b][k] W2[k][j]) * silu'(pre[b][j]); dparams[b][p] += sum_j dpre W1[j][p]
__global__ __launch_bounds__(256) void film_bwd_a_kernel(FilmArgs a) {
  __shared__ float dout[2 * 1024];
  __shared__ float dp[FILM_MAX_HID];
  const int b = blockIdx.x;
  const FilmGenDesc& d = a.g[blockIdx.y];
  for (int k = threadIdx.x; k < 2 * d.C; k += blockDim.x) {
    const int c = k < d.C ? k : k - d.C;
    const float* src = k < d.C ? d.dgf : d.dbf;
    dout[k] = src[((int64_t)b * d.C + c) * d.dg_stride];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < a.Hf; j += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < 2 * d.C; ++k) s += dout[k] * d.w2[(int64_t)k * a.Hf + j];
    const float v = s * dsiluf_(d.pre[(int64_t)b * a.Hf + j]);
    d.dpre[(int64_t)b * a.Hf + j] = v;
    dp[j] = v;
  }
  __syncthreads();
  if (a.dparams)
    for (int p = threadIdx.x; p < a.P; p += blockDim.x) {
      float s = 0.f;
      for (int j = 0; j < a.Hf; ++j) s += dp[j] * d.w1[j * a.P + p];
      atomicAdd(a.dparams + (int64_t)b * a.P + p, s);
    }
}

// phase B (grid = row blocks x ngen): weight / bias gradients, reduced over the batch in a fixed order.
__global__ __launch_bounds__(256) void film_bwd_b_kernel(FilmArgs a) {
  const FilmGenDesc& d = a.g[blockIdx.y];
  const int n_w2 = 2 * d.C * a.Hf, n_b2 = 2 * d.C, n_w1 = a.Hf * a.P, n_b1 = a.Hf;
  const int total = n_w2 + n_b2 + n_w1 + n_b1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float s = 0.f;
    float* dst;
    if (i < n_w2) {
      const int k = i / a.Hf, j = i - k * a.Hf;
      const int c = k < d.C ? k : k - d.C;
      const float* src = k < d.C ? d.dgf : d.dbf;
      for (int b = 0; b < a.B; ++b)
        s += src[((int64_t)b * d.C + c) * d.dg_stride] * siluf_(d.pre[(int64_t)b * a.Hf + j]);
      dst = d.dw2 + i;
    } else if (i < n_w2 + n_b2) {
      const int k = i - n_w2;
      const int c = k < d.C ? k : k - d.C;
      const float* src = k < d.C ? d.dgf : d.dbf;
      for (int b = 0; b < a.B; ++b) s += src[((int64_t)b * d.C + c) * d.dg_stride];
      dst = d.db2 + k;
    } else if (i < n_w2 + n_b2 + n_w1) {
      const int q = i - n_w2 - n_b2;
      const int j = q / a.P, p = q - j * a.P;
      for (int b = 0; b < a.B; ++b) s += d.dpre[(int64_t)b * a.Hf + j] * a.params[(int64_t)b * a.P + p];
      dst = d.dw1 + q;
    } else {
      const int j = i - n_w2 - n_b2 - n_w1;
      for (int b = 0; b < a.B; ++b) s += d.dpre[(int64_t)b * a.Hf + j];
      dst = d.db1 + j;
    }
    *dst = a.accumulate ? *dst + s : s;
  }
}

int film_forward(const FilmArgs& a, hipStream_t st) {
  if (a.ngen <= 0) return SX_OK;
  if (a.ngen > FILM_MAX_GEN || a.Hf > FILM_MAX_HID || a.P > 64) {
    set_error("FiLM: unsupported sizes (generators=%d hidden=%d P=%d)", a.ngen, a.Hf, a.P);
    return SX_ERR_UNSUPPORTED;
  }
  for (int i = 0; i < a.ngen; ++i)
    if (a.g[i].C > 1024) { set_error("FiLM: C=%d > 1024", a.g[i].C); return SX_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL(film_fwd_kernel, dim3(a.B, a.ngen), dim3(256), 0, st, a);
  SX_LAUNCH_CHECK("film_fwd");
  return SX_OK;
}

int film_backward(const FilmArgs& a, bool weight_grads, hipStream_t st) {
  if (a.ngen <= 0) return SX_OK;
  hipLaunchKernelGGL(film_bwd_a_kernel, dim3(a.B, a.ngen), dim3(256), 0, st, a);
  SX_LAUNCH_CHECK("film_bwd_a");
  if (weight_grads) {
    hipLaunchKernelGGL(film_bwd_b_kernel, dim3(64, a.ngen), dim3(256), 0, st, a);
    SX_LAUNCH_CHECK("film_bwd_b");
  }
  return SX_OK;
}

}  // namespace sx
