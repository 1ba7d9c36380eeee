// FiLM generators (models.py:38-58): [gamma|beta] = W2 * SiLU(W1 p + b1) + b2, returned as
// (1 + gamma, beta) per (sample, channel).  All generators of the network run in ONE launch
// (grid = samples x generators).  Tiny GEMVs: <0.01 % of the FLOPs, so plain FMA code.
#include "sx_common.h"
#include "film.h"
#include <algorithm>

namespace sx {

__global__ __launch_bounds__(256) void film_fwd_kernel(FilmArgs a) {
  __shared__ float hid[FILM_MAX_HID];
  __shared__ float pv[64];
  const int b = blockIdx.x;
  const FilmGenDesc& d = a.g[blockIdx.y];
  if (a.zero) {
    const int nthr = gridDim.x * gridDim.y * blockDim.x;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < a.zero_n; i += nthr) a.zero[i] = 0.f;
  }
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

// phase A (grid = samples/8 x ngen): dpre[b][j] = (sum_k dout[b][k] W2[k][j]) * silu'(pre[b][j]);
// dparams[b][p] += sum_j dpre[b][j] W1[j][p].  A block handles 4 samples so every W2 element it loads serves 4 FMAs;
// threads = (hidden unit j) x (k mod FKP), combined through LDS.
static constexpr int FB = 4;
static constexpr int FKP = 8;        // threads that share one hidden unit (k is split FKP ways): 128 x FKP threads
__global__ __launch_bounds__(1024) void film_bwd_a_kernel(FilmArgs a) {
  __shared__ float dout[FB][2 * 1024];
  __shared__ float part[FKP][FB][128];
  __shared__ float dp[FB][128];
  const FilmGenDesc& d = a.g[blockIdx.y];
  const int b0 = blockIdx.x * FB, nb = min(FB, a.B - b0);
  const int K = 2 * d.C;
  for (int i = threadIdx.x; i < FB * K; i += blockDim.x) {
    const int bb = i / K, k = i - bb * K;
    const int c = k < d.C ? k : k - d.C;
    const float* src = k < d.C ? d.dgf : d.dbf;
    dout[bb][k] = bb < nb ? src[((int64_t)(b0 + bb) * d.C + c) * d.dg_stride] : 0.f;
  }
  __syncthreads();
  const int jl = threadIdx.x & 127, kp = threadIdx.x >> 7;
  for (int j0 = 0; j0 < a.Hf; j0 += 128) {
    const int j = j0 + jl;
    float s[FB];
#pragma unroll
    for (int bb = 0; bb < FB; ++bb) s[bb] = 0.f;
    if (j < a.Hf)
      for (int k = kp; k < K; k += 16 * FKP) {    // 16 independent loads in flight per thread before the first FMA
        float w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = (k + FKP * u < K) ? d.w2[(int64_t)(k + FKP * u) * a.Hf + j] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int bb = 0; bb < FB; ++bb) s[bb] += dout[bb][min(k + FKP * u, K - 1)] * w[u];
      }
#pragma unroll
    for (int bb = 0; bb < FB; ++bb) part[kp][bb][jl] = s[bb];
    __syncthreads();
    for (int i = threadIdx.x; i < FB * 128; i += blockDim.x) {
      const int bb = i >> 7, jj = i & 127;
      float v = 0.f;
      if (bb < nb && j0 + jj < a.Hf) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < FKP; ++q) t += part[q][bb][jj];
        v = t * dsiluf_(d.pre[(int64_t)(b0 + bb) * a.Hf + j0 + jj]);
        d.dpre[(int64_t)(b0 + bb) * a.Hf + j0 + jj] = v;
      }
      dp[bb][jj] = v;
    }
    __syncthreads();
    if (a.dparams)
      for (int i = threadIdx.x; i < nb * a.P; i += blockDim.x) {
        const int bb = i / a.P, p = i - bb * a.P;
        float v = 0.f;
        for (int jj = 0; jj < min(128, a.Hf - j0); ++jj) v += dp[bb][jj] * d.w1[(j0 + jj) * a.P + p];
        atomicAdd(a.dparams + (int64_t)(b0 + bb) * a.P + p, v);
      }
    __syncthreads();
  }
}

// phase B (grid = row chunks x ngen): weight / bias gradients, summed over the batch in a fixed order.
// Rows 0 .. 2C-1 are rows of dW2 (+ db2), rows 2C .. 2C+Hf-1 rows of dW1 (+ db1).  A block takes FR rows and
// stages the batch-side operand once in LDS: hid[b][j] = silu(pre[b][j]) for the dW2 rows, params[b][p] for dW1.
static constexpr int FR = 16, FBT = 32;      // rows per block, batch tile
__global__ __launch_bounds__(256) void film_bwd_b_kernel(FilmArgs a) {
  __shared__ float hid[FBT][128 + 1];
  __shared__ float lhs[FBT][FR];
  const FilmGenDesc& d = a.g[blockIdx.y];
  const int K2 = 2 * d.C;
  const int r0 = blockIdx.x * FR;
  if (r0 >= K2 + a.Hf) return;
  const bool w2rows = r0 < K2;                 // K2 and Hf chunks never share a block (K2 % FR == 0)
  const int ncols = w2rows ? a.Hf : a.P;       // columns of the gradient rows this block owns
  const int nrows = min(FR, (w2rows ? K2 : K2 + a.Hf) - r0);
  for (int c0 = 0; c0 < ncols; c0 += 128) {
    const int cw = min(128, ncols - c0);
    // thread -> (column, row group): 128 columns x 2 row halves
    const int cl = threadIdx.x & 127, rh = threadIdx.x >> 7;
    float acc[FR / 2], bias[FR / 2];
#pragma unroll
    for (int i = 0; i < FR / 2; ++i) { acc[i] = 0.f; bias[i] = 0.f; }
    for (int bt = 0; bt < a.B; bt += FBT) {
      const int nb = min(FBT, a.B - bt);
      __syncthreads();
      for (int i = threadIdx.x; i < FBT * 128; i += blockDim.x) {
        const int bb = i >> 7, cc = i & 127;
        float v = 0.f;
        if (bb < nb && cc < cw)
          v = w2rows ? siluf_(d.pre[(int64_t)(bt + bb) * a.Hf + c0 + cc]) : a.params[(int64_t)(bt + bb) * a.P + c0 + cc];
        hid[bb][cc] = v;
      }
      for (int i = threadIdx.x; i < FBT * FR; i += blockDim.x) {
        const int bb = i / FR, rr = i - bb * FR;
        float v = 0.f;
        if (bb < nb && rr < nrows) {
          const int row = r0 + rr;
          if (w2rows) {
            const int c = row < d.C ? row : row - d.C;
            v = (row < d.C ? d.dgf : d.dbf)[((int64_t)(bt + bb) * d.C + c) * d.dg_stride];
          } else {
            v = d.dpre[(int64_t)(bt + bb) * a.Hf + (row - K2)];
          }
        }
        lhs[bb][rr] = v;
      }
      __syncthreads();
      for (int bb = 0; bb < nb; ++bb) {
        const float h = hid[bb][cl];
#pragma unroll
        for (int i = 0; i < FR / 2; ++i) {
          const float l = lhs[bb][rh * (FR / 2) + i];
          acc[i] += l * h;
          bias[i] += l;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < FR / 2; ++i) {
      const int rr = rh * (FR / 2) + i;
      if (rr >= nrows) continue;
      const int row = r0 + rr;
      if (cl < cw) {
        float* dst = w2rows ? d.dw2 + (int64_t)row * a.Hf + c0 + cl : d.dw1 + (int64_t)(row - K2) * a.P + c0 + cl;
        *dst = a.accumulate ? *dst + acc[i] : acc[i];
      }
      if (c0 == 0 && cl == 0) {
        float* dst = w2rows ? d.db2 + row : d.db1 + (row - K2);
        *dst = a.accumulate ? *dst + bias[i] : bias[i];
      }
    }
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
  hipLaunchKernelGGL(film_bwd_a_kernel, dim3(cdiv(a.B, FB), a.ngen), dim3(128 * FKP), 0, st, a);
  SX_LAUNCH_CHECK("film_bwd_a");
  if (weight_grads) {
    int cmax = 0;
    for (int i = 0; i < a.ngen; ++i) cmax = std::max(cmax, a.g[i].C);
    // 2C is a multiple of FR (C % 8 == 0), so a block never mixes dW2 rows with dW1 rows
    hipLaunchKernelGGL(film_bwd_b_kernel, dim3(cdiv(2 * cmax + a.Hf, FR), a.ngen), dim3(256), 0, st, a);
    SX_LAUNCH_CHECK("film_bwd_b");
  }
  return SX_OK;
}

}  // namespace sx
