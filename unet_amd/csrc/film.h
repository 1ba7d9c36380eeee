// FiLM generator launch descriptors shared by film.hip and unet_plan.hip.
#pragma once
#include "sx_common.h"
#include "norm_act.h"

namespace sx {

static constexpr int FILM_MAX_GEN = 16;
static constexpr int FILM_MAX_HID = 1024;

struct FilmGenDesc {
  const float* w1; const float* b1; const float* w2; const float* b2;   // (Hf,P) (Hf) (2C,Hf) (2C)
  float* dw1; float* db1; float* dw2; float* db2;                       // gradients (bwd)
  float* gf; float* bf;          // [B][C] outputs
  float* pre;                    // [B][Hf] saved pre-activation
  const float* dgf; const float* dbf; int dg_stride;   // incoming grads: element (b,c) at (b*C + c)*dg_stride
  float* dpre;                   // [B][Hf] scratch (bwd)
  int C;
};
struct FilmArgs {
  FilmGenDesc g[FILM_MAX_GEN];
  int ngen, B, P, Hf;
  const float* params;   // [B][P]
  float* dparams;        // [B][P] (bwd, accumulated with atomics; caller zeroes)
  int accumulate;
  float* zero; int zero_n;   // forward only: fp32 words the launch also clears (the network's GroupNorm statistics; saves a memset dispatch)
};

// One block of the forward pass: generator (w1, b1, w2, b2) for sample b.  hid: FILM_MAX_HID floats of LDS, pv: 64.
// Shared by film_fwd_kernel and the merged encode-head launch (conv_igemm.hip).
__device__ __forceinline__ void film_fwd_block(const float* w1, const float* b1, const float* w2, const float* b2, float* gf, float* bf,
                                               float* pre, int C, int b, int P, int Hf, const float* params, float* hid, float* pv) {
  for (int p = threadIdx.x; p < P; p += blockDim.x) pv[p] = params[(int64_t)b * P + p];
  __syncthreads();
  for (int j = threadIdx.x; j < Hf; j += blockDim.x) {
    float s = b1[j];
    for (int p = 0; p < P; ++p) s += w1[j * P + p] * pv[p];
    if (pre) pre[(int64_t)b * Hf + j] = s;
    hid[j] = siluf_(s);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * C; k += blockDim.x) {
    float s = b2[k];
    const float* w = w2 + (int64_t)k * Hf;
    for (int j = 0; j < Hf; ++j) s += w[j] * hid[j];
    if (k < C) gf[(int64_t)b * C + k] = 1.0f + s;
    else bf[(int64_t)b * C + (k - C)] = s;
  }
}

int film_forward(const FilmArgs& a, hipStream_t st);
// gn (nullable): GroupNorm affine / bias gradient reductions of the same backward half.  In a training step (weight gradients
// wanted, no gradient w.r.t. the FiLM inputs) ONE launch runs the dW2 rows, the dW1 rows -- each block rebuilding the dpre
// values of its 16 hidden units itself instead of waiting for a phase-A launch -- and these reductions: three dispatches
// (film_bwd_a 15 us, film_bwd_b 20 us, gn_param_grad 12 us) become one.  Returns with *gn_done = 1 when they rode along.
// rd (nullable): pending slab reductions of the same half (their split-K launch already enqueued): they join the launch too
// when the table fits, *rd_done = 1.
struct WgReduceDesc;
int film_backward(const FilmArgs& a, bool weight_grads, hipStream_t st, const GnGradDesc* gn = nullptr, int n_gn = 0,
                  int* gn_done = nullptr, const WgReduceDesc* rd = nullptr, int n_rd = 0, int* rd_done = nullptr);

}  // namespace sx
