// Shared declarations of the weight-gradient kernels (conv_wgrad.hip: generic tier + slab reduction,
// wgrad_ws.hip: persistent LDS-DMA bf16 tier).
#pragma once
#include "sx_common.h"

namespace sx {

// One pending slab reduction: out[(a*Cb + b)*ntaps + tap] (+)= sum_s slab[s][tap][a][b]
struct WgReduceDesc {
  const float* slab;
  float* out;
  int nsplit, ntaps, Ca, Cb, CaPad, CbPad;   // Ca / Cb: channels actually written (<= padded slab extents)
  int accumulate;
  int tiled;                                 // slab layout: 0 = [tap][a][b]; 1 = 16x16 tiles, [tap][a/16][b/16][a%16][b%16] (wgrad_ws)
  int a16_lo, a16_n, tap_lo, tap_n;          // tiled slabs: the row blocks / taps this descriptor reduces (a16_n = 0: all of them)
  int blk0, nblk;                            // block range inside a fused launch (filled by wgrad_reduce_multi)
};
static constexpr int WG_MAX_REDUCE = 40;     // descriptors per fused launch (kernel-argument budget)

// Fused reduction of up to WG_MAX_REDUCE pending slabs (more are split over several launches).
int wgrad_reduce_multi(const WgReduceDesc* descs, int n, hipStream_t st);

// Launches the split-K kernel of one layer into `slab` and fills `desc` (slab, nsplit, ntaps, CaPad, CbPad).
// The caller completes desc (out, Ca, Cb, accumulate) and reduces now or later.
// `defer` (nullable): a problem the persistent bf16 tier covers is not launched but described in *defer, and the call
// returns WG_DEFERRED; the caller launches a batch of them with wgrad_ws_launch (before reducing).
struct WgWsLaunch;
static constexpr int WG_DEFERRED = 2;
int wgrad_launch(int dtype, int mode, int B, int H, int W, int Ca, int Cb, const void* P, int p_ld, const void* Q,
                 int q_ld, int Hq, int Wq, void* slab, int64_t slab_bytes, WgReduceDesc* desc, hipStream_t st, WgWsLaunch* defer);

struct WgWsArgs {
  const bf16* P; const bf16* Q; float* slab;
  int B, H, W, p_ld, q_ld;
  int Hq, Wq;                   // Q image dims (= H, W except for the transposed conv: >= 2H, 2W)
  int Ca, Cb, CaPad, CbPad;
  int a_tiles, b_tiles, tg_count, R, S;
  int tiles_x, tiles_y, ntiles, tiles_per_split;
  int dbg;
  unsigned long long* stamps;   // diagnostic build only
};

// A prepared problem of the persistent bf16 tier: kernel arguments, instantiation code, grid and dynamic LDS.
struct WgWsLaunch { WgWsArgs a; int variant, blocks, smem; };
int wgrad_ws_prepare(int mode, int B, int H, int W, int Ca, int Cb, const void* P, int p_ld, const void* Q, int q_ld, int Hq, int Wq,
                     float* slab, WgReduceDesc* desc, WgWsLaunch* out, int dual = 0, int batched = 1);
// batched: the problem will share a launch with others (fewer, longer splits: it need not fill the chip alone); the slab
// never exceeds wgrad_ws_slab_bytes(), which sizes for the stand-alone plan
int wgrad_ws_launch(const WgWsLaunch* l, int n, hipStream_t st);

// Persistent bf16 tier.  slab bytes it needs (0 = shape not covered) / launch (0 = launched, 1 = not covered, <0 error).
int64_t wgrad_ws_slab_bytes(int mode, int B, int H, int W, int Ca, int Cb);
int wgrad_ws_try(int mode, int B, int H, int W, int Ca, int Cb, const void* P, int p_ld, const void* Q, int q_ld, int Hq, int Wq,
                 float* slab, WgReduceDesc* desc, hipStream_t st, int dual = 0);
// dual: P = [dY1 | dR] of a DoubleConv with 2 * C == 64 channels (one a tile) and >= 64 input channels; rows C .. 2C-1 of the
// slab are valid for the centre tap only (the 1x1 shortcut's weight gradient).  The caller reduces the two row ranges with
// two descriptors (a16_lo / a16_n / tap_lo / tap_n).

}  // namespace sx
