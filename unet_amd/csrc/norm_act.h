// Batched GroupNorm affine-parameter gradients (norm_act.hip), used by unet_plan.hip.
#pragma once
#include "sx_common.h"

namespace sx {

// sums: [nb][C][stride] fp32; element 0 -> dgamma, element 1 -> dbeta (dbeta may be null: bias gradients reuse the
// launch with nb = 1, stride = 1).  nb = 0 means the launch's batch size.
struct GnGradDesc { const float* sums; float* dgamma; float* dbeta; int C, blk0, nb, stride; };
static constexpr int GN_GRAD_MAX = 32;
struct GnGradArgs { GnGradDesc d[GN_GRAD_MAX]; int n, B, accumulate; };

// dgamma[c] (+)= sum_b sums[b][c][0], dbeta[c] (+)= sum_b sums[b][c][1] for every descriptor, one launch per 32
int gn_param_grad_multi(const GnGradDesc* descs, int n, int B, int accumulate, hipStream_t st);

// Phase 1 of GroupNorm + SiLU backward alone (per-(sample, channel) sums; mode 0 or 2 of sx_gn_silu_backward): what a
// producer of d_out runs when it could not fuse the sums into its own epilogue.
int gn_bwd_reduce_only(int dtype, int mode, int B, int HW, int C, int groups, float eps, const void* y, int y_ld, const float* stats,
                       const float* gamma, const float* beta, const void* d_out, int do_ld, const void* s, int s_ld,
                       const float* film_g, const float* drop, float* sums, hipStream_t st);

}  // namespace sx
