// Batched GroupNorm affine-parameter gradients (norm_act.hip), used by unet_plan.hip.
#pragma once
#include "sx_common.h"

namespace sx {

struct GnGradDesc { const float* sums; float* dgamma; float* dbeta; int C, blk0; };   // sums: [B][C][4] fp32
static constexpr int GN_GRAD_MAX = 32;
struct GnGradArgs { GnGradDesc d[GN_GRAD_MAX]; int n, B, accumulate; };

// dgamma[c] (+)= sum_b sums[b][c][0], dbeta[c] (+)= sum_b sums[b][c][1] for every descriptor, one launch per 32
int gn_param_grad_multi(const GnGradDesc* descs, int n, int B, int accumulate, hipStream_t st);

}  // namespace sx
