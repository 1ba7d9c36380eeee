// Shared declarations of the weight-gradient kernels (conv_wgrad.hip: generic tier + slab reduction,
// wgrad_ws.hip: persistent LDS-DMA bf16 tier).
#pragma once
#include "sx_common.h"
#include <algorithm>

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

// One 256-thread block of the fused slab reduction; blk_local = its index inside the descriptor's D.nblk blocks.
// A WAVE owns a unit (256 contiguous bytes x all splits): lane = (split lane sl = lane / 16, e = lane % 16); a thread adds
// the float4 of splits sl, sl + 4, ... (independent loads, unrolled), the four split lanes are combined by two lane
// exchanges, lane (e, sl) stores element sl of the sum.  No LDS, no block barrier.  (Rounds 1-2: a BLOCK per unit, 16
// split lanes, one load per thread and two barriers per unit -- 6000 blocks of one unit each, 2.4 TB/s.)
__device__ __forceinline__ void wgrad_reduce_block(const WgReduceDesc& D, int blk_local) {
  const int64_t slab_sz = (int64_t)D.ntaps * D.CaPad * D.CbPad;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = lane & 15, sl = lane >> 4;
  const int A16 = D.CaPad >> 4, B16 = D.CbPad >> 4;
  const int a16_n = D.a16_n ? D.a16_n : A16, tap_n = D.a16_n ? D.tap_n : D.ntaps;
  const int bchunks = (D.Cb + 63) >> 6;
  const int64_t nunits = D.tiled ? (int64_t)tap_n * a16_n * B16 * 4 : (int64_t)D.ntaps * D.Ca * bchunks;
  for (int64_t u = ((int64_t)blk_local) * 4 + wave; u < nunits; u += (int64_t)D.nblk * 4) {
    const float* src;
    int aa, bb, tap, tn;
    bool ok;
    if (D.tiled) {
      // unit = four consecutive rows of one 16x16 tile; thread e: row 4q + e/4, cols (e%4)*4..+3
      const int q = (int)(u & 3);
      const int64_t tl = u >> 2;                                // (tap * a16_n + a16) * B16 + b16, local to this descriptor
      const int b16 = (int)(tl % B16);
      const int64_t ta = tl / B16;
      const int a16 = (int)(ta % a16_n);
      tap = (int)(ta / a16_n);
      const int64_t tile = ((int64_t)(D.tap_lo + tap) * A16 + D.a16_lo + a16) * B16 + b16;
      src = D.slab + (tile << 8) + q * 64 + e * 4;
      aa = a16 * 16 + q * 4 + (e >> 2); bb = b16 * 16 + (e & 3) * 4 + sl; tn = tap_n;
      ok = true;
    } else {
      // unit = 64 consecutive b of one (tap, a) row
      const int bc = (int)(u % bchunks);
      const int64_t row = u / bchunks;
      aa = (int)(row % D.Ca); tap = (int)(row / D.Ca);
      const int b = bc * 64 + e * 4;
      src = D.slab + ((int64_t)tap * D.CaPad + aa) * D.CbPad + b;
      bb = b + sl; tn = D.ntaps;
      ok = b < D.CbPad;
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
#pragma unroll 4
      for (int k = sl; k < D.nsplit; k += 4) s += *reinterpret_cast<const f32x4*>(src + k * slab_sz);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = s[r];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      s[r] = v;
    }
    if (aa < D.Ca && bb < D.Cb) {
      const float t = sl == 0 ? s[0] : (sl == 1 ? s[1] : (sl == 2 ? s[2] : s[3]));
      float* dst = D.out + ((int64_t)aa * D.Cb + bb) * tn + tap;
      *dst = D.accumulate ? (*dst + t) : t;
    }
  }
}
// blocks a descriptor gets in a fused launch (four units per block, <= 1024 blocks)
inline int wgrad_reduce_blocks(const WgReduceDesc& d) {
  const long long units = d.tiled ? (long long)(d.a16_n ? d.tap_n : d.ntaps) * (d.a16_n ? d.a16_n : (d.CaPad >> 4)) * (d.CbPad >> 4) * 4
                                  : (long long)d.ntaps * d.Ca * ((d.Cb + 63) / 64);
  return (int)std::min<long long>((units + 3) / 4, 1024);
}

// Fused reduction of up to WG_MAX_REDUCE pending slabs (more are split over several launches).
int wgrad_reduce_multi(const WgReduceDesc* descs, int n, hipStream_t st);

// Launches the split-K kernel of one layer into `slab` and fills `desc` (slab, nsplit, ntaps, CaPad, CbPad).
// The caller completes desc (out, Ca, Cb, accumulate) and reduces now or later.
// `defer` (nullable): a problem the persistent bf16 tier covers is not launched but described in *defer, and the call
// returns WG_DEFERRED; the caller launches a batch of them with wgrad_ws_launch (before reducing).
struct WgWsLaunch;
struct WgX3Launch;
static constexpr int WG_DEFERRED = 2, WG_DEFERRED_X3 = 3;
// `defer_x3` (nullable): the same for the fp32 tier's split-bf16 kernel (WG_DEFERRED_X3, wgrad_x3_launch)
int wgrad_launch(int dtype, int mode, int B, int H, int W, int Ca, int Cb, const void* P, int p_ld, const void* Q,
                 int q_ld, int Hq, int Wq, void* slab, int64_t slab_bytes, WgReduceDesc* desc, hipStream_t st, WgWsLaunch* defer,
                 WgX3Launch* defer_x3 = nullptr);

// arguments of the generic split-K kernels (conv_wgrad.hip)
struct WgradArgs {
  const void* P; const void* Q; float* slab;
  int B, H, W;          // P pixel grid
  int Hq, Wq;           // Q pixel dims
  int p_ld, q_ld;
  int Ca, Cb, CaPad, CbPad;
  int ntaps, tapw, q_scale, pad;
  int tw_log2, TW, TH, LW, LH, TP;   // TP = pixels per tile (multiple of 32)
  int tiles_x, tiles_y, ntiles, tiles_per_split;
  int nbf;              // b fragments (16 channels) in this block's b tile
};
// A prepared fp32-tier problem: blocks = a_tiles * b_tiles * nsplit of 512 threads, `cost` = estimated time of one block
struct WgX3Launch { WgradArgs a; int variant, a_tiles, b_tiles, nsplit, smem; double cost; };
int wgrad_x3_launch(WgX3Launch* l, int n, hipStream_t st);      // (sorts l: longest blocks first)

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
