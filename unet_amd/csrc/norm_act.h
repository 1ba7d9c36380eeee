// Batched GroupNorm affine-parameter gradients (norm_act.hip), used by unet_plan.hip.
#pragma once
#include "sx_common.h"

namespace sx {

// sums: [nb][C][stride] fp32; element 0 -> dgamma, element 1 -> dbeta (dbeta may be null: bias gradients reuse the
// launch with nb = 1, stride = 1).  nb = 0 means the launch's batch size.
struct GnGradDesc { const float* sums; float* dgamma; float* dbeta; int C, blk0, nb, stride; };
static constexpr int GN_GRAD_MAX = 32;
struct GnGradArgs { GnGradDesc d[GN_GRAD_MAX]; int n, B, accumulate; };

// One block (256 threads: 64 channels x 4 batch lanes, fixed-order LDS combine) of the reduction below; `block` indexes the
// launch's block ranges (GnGradDesc::blk0).  Shared with the FiLM tail launch of a training step (film.hip), which runs these
// blocks beside its own instead of paying a dispatch for them.
// (the table is read from the kernel-argument segment: see film.hip)
typedef const GnGradArgs __attribute__((address_space(4))) * GnGradArgsK;
// (AK: any kernel-argument table with members d[], n, B, accumulate -- the training tail carries a shorter one)
template <typename AK>
__device__ __forceinline__ void gn_param_grad_block(AK Ak, int block) {
  __shared__ float gn_red[2][4][64];
  int di = 0;
  const int n = Ak->n;
  while (di + 1 < n && block >= Ak->d[di + 1].blk0) ++di;
  GnGradDesc D;
  D.sums = Ak->d[di].sums; D.dgamma = Ak->d[di].dgamma; D.dbeta = Ak->d[di].dbeta;
  D.C = Ak->d[di].C; D.blk0 = Ak->d[di].blk0; D.nb = Ak->d[di].nb; D.stride = Ak->d[di].stride;
  struct { int B, accumulate; } A = {Ak->B, Ak->accumulate};
  const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const int c = (block - D.blk0) * 64 + cl;
  float g = 0.f, bt = 0.f;
  const int nb = D.nb > 0 ? D.nb : A.B;
  if (c < D.C)
    for (int b = bl; b < nb; b += 4) {
      const float* v = D.sums + ((int64_t)b * D.C + c) * D.stride;
      g += v[0];
      if (D.dbeta) bt += v[1];
    }
  gn_red[0][bl][cl] = g;
  gn_red[1][bl][cl] = bt;
  __syncthreads();
  if (bl == 0 && c < D.C) {
    g = gn_red[0][0][cl] + gn_red[0][1][cl] + gn_red[0][2][cl] + gn_red[0][3][cl];
    bt = gn_red[1][0][cl] + gn_red[1][1][cl] + gn_red[1][2][cl] + gn_red[1][3][cl];
    D.dgamma[c] = A.accumulate ? D.dgamma[c] + g : g;
    if (D.dbeta) D.dbeta[c] = A.accumulate ? D.dbeta[c] + bt : bt;
  }
}
// block ranges of a descriptor table; returns the number of blocks
template <typename ARGS>
inline int gn_param_grad_layout(ARGS& A) {
  int blk = 0;
  for (int i = 0; i < A.n; ++i) { A.d[i].blk0 = blk; blk += (A.d[i].C + 63) / 64; }
  return blk;
}

// dgamma[c] (+)= sum_b sums[b][c][0], dbeta[c] (+)= sum_b sums[b][c][1] for every descriptor, one launch per 32
int gn_param_grad_multi(const GnGradDesc* descs, int n, int B, int accumulate, hipStream_t st);

// true when sx_gn_silu_backward runs both phases of this shape in one launch (act_bwd_group_kernel): a producer of d_out then
// need not form the sums
bool gn_bwd_group_fused_ok(int B, int HW, int C, int groups);

// Phase 1 of GroupNorm + SiLU backward alone (per-(sample, channel) sums; mode 0 or 2 of sx_gn_silu_backward): what a
// producer of d_out runs when it could not fuse the sums into its own epilogue.
int gn_bwd_reduce_only(int dtype, int mode, int B, int HW, int C, int groups, float eps, const void* y, int y_ld, const float* stats,
                       const float* gamma, const float* beta, const void* d_out, int do_ld, const void* s, int s_ld,
                       const float* film_g, const float* drop, float* sums, hipStream_t st);

}  // namespace sx
