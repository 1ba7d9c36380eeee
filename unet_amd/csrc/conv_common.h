// Shared declarations of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_ws.hip).
#pragma once
#include "sx_common.h"

namespace sx {

static constexpr int ROWB = 96;   // LDS bytes per staged row (64 payload + 32 pad)

// bf16 packed-weight layout.  Units of 16 rows x 32 k (1 KB) in the order [row / 16][k / 32][tap]; inside a unit the 16-byte
// group (row, 8 consecutive k) sits at lane g * 16 + n, g = (k % 32) / 8, n = row % 16 -- the lane that feeds it to the
// 16x16x32 MFMA as the A operand.  A wave of conv_ws_kernel therefore fetches each (16 rows, 32 k, tap) fragment as ONE
// contiguous 1 KB load; the [tap][row][k] form cost 16 half-line requests per fragment and the weight prologue ran at
// 13 B/cycle/CU.  fp32 packs stay [tap][row][k].
__host__ __device__ inline int64_t wpack_off_bf16(int ntaps, int kchunks, int tap, int row, int k) {
  return ((((int64_t)(row >> 4) * kchunks + (k >> 5)) * ntaps + tap) << 9) + (((((k & 31) >> 3) << 4) + (row & 15)) << 3) + (k & 7);
}

// fp32 tier in split form (conv_x3.hip): per (tap, row, 32-k chunk) `planes` runs of 32 bf16 values -- hi = bf16(w),
// mid = bf16(w - hi) [, lo = bf16(w - hi - mid)] -- 128 bytes (= the plain fp32 pack [tap][row][k]) with two planes, 192 with
// three.  Index of the hi value in bf16 elements; plane q sits 32 * q elements further.
__host__ __device__ inline int64_t wpack_off_x3(int RTot, int kchunks, int tap, int row, int k, int planes = 2) {
  return (((int64_t)tap * RTot + row) * kchunks + (k >> 5)) * (32 * planes) + (k & 31);
}

struct ConvArgs {
  const void* x; const void* w; void* y; const float* bias; float* gn_stats;
  int B, H, W;
  int Hx, Wx, x_ld;
  int Hy, Wy, y_ld;
  int K, KPad;        // GEMM-K channels per tap (real / padded)
  int KRow;           // row stride of the packed weights in elements (0 = KPad; set when a launch covers a K slice)
  int R, RPad, RTot;  // output channels (real / padded per tap group / rows per tap in `w`)
  int ntaps, tapw;
  int in_scale, pad;
  int shuffle, accumulate;
  int epi_counted;    // conv_ws: full tiles store through counted instructions (set by the launcher; SX_WS_EPI_COUNTED=0: A/B knob)
  int tw_log2, TW, TH, LW, LH;
  int tiles_x, tiles_y;
  int cpg;
  // fused variants of the weights-stationary kernel (conv_ws.hip)
  const void* w2;     // packed 1x1 weights: shortcut forward (second output) / shortcut dgrad (extra centre-tap K chunks)
  void* y2; int y2_ld;
  unsigned long long* stamps;
  int kc0;            // first 32-wide K chunk of the packed weights this launch covers (K slices, see KRow)    // diagnostic build (-DSX_WS_DBG=1) only: per-block s_memtime stamps
  float* ch_sum; int ch_sum_n;   // optional: ch_sum[c] += sum over all output pixels of y[..., c], c < ch_sum_n (atomics)
  // WS_BS: y is the gradient w.r.t. the output of GroupNorm + SiLU (bs_mode 1) or of a whole DoubleConv block output
  // (bs_mode 2: FiLM, dropout, residual in front); the epilogue accumulates the per-(sample, channel) sums GroupNorm's
  // backward needs (norm_act.hip: act_bwd_reduce_kernel) while the values are in registers:
  //   sums[b][c] += { sum dz * n_hat, sum dz, sum y * R, sum y },  dz = y * gf * drop * silu'(z)      (last two: mode 2)
  int bs_mode;
  const void* bs_y; int bs_y_ld;        // raw convolution output the GroupNorm normalised, same pixel grid as y
  const void* bs_s; int bs_s_ld;        // mode 2: the block's shortcut branch
  const float* bs_stats;                // [B][G][2] sum / sum of squares (cpg = channels per group, bs_inv_n = 1 / (cpg * H * W))
  const float* bs_gamma; const float* bs_beta;
  const float* bs_gf; const float* bs_drop;   // [B][C] FiLM scale, dropout factor (nullable)
  float* bs_sums;                       // [B][C][4], zeroed by the caller
  float bs_eps, bs_inv_n;
};

template <typename T> struct Mma;
template <> struct Mma<bf16> {
  __device__ static __forceinline__ f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // 16 k-values per chunk: lane group g owns k = 4g..4g+3, MFMA step s consumes element s of
  // every group (a consistent permutation of k on both operands).
  __device__ static __forceinline__ f32x4 run(f32x4 a, f32x4 b, f32x4 c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
    return c;
  }
};


// GroupNorm partial sums of a block's channels -> group sums -> one atomic per group and statistic.  sm_stat holds
// [WN][BMC][2] per-channel (sum, sum of squares); called by every thread of the block after the barrier that publishes it.
// Power-of-two group sizes: a butterfly over the lanes of a group.  (One thread walking its group's channels through LDS
// was a chain of 2 * cpg dependent reads: ~6k cycles per flush at 32 channels per group, paid per TILE at the 16x8 level --
// round-3 ablation stamps, profiles/r03_conv_kq_stamps.txt.)  Other sizes keep the walk.
template <int BMC, int WN>
__device__ __forceinline__ void gn_stats_flush(const float* sm_stat, int tid, int cb, int R, int cpg, float* stats_b) {
  const int cend = min(R, cb + BMC);
  if ((BMC & (BMC - 1)) == 0 && (cpg & (cpg - 1)) == 0) {
    if ((tid >> 6) * 64 < BMC) {
      const int c = cb + tid;
      float t1 = 0.f, t2 = 0.f;
      if (c < cend) {
#pragma unroll
        for (int w = 0; w < WN; ++w) {
          t1 += sm_stat[(w * BMC + tid) * 2 + 0];
          t2 += sm_stat[(w * BMC + tid) * 2 + 1];
        }
      }
      const int seg = min(min(cpg, BMC), 64);      // divides cpg and BMC (powers of two): a segment never straddles groups
      for (int o = 1; o < seg; o <<= 1) {
        t1 += __shfl_xor(t1, o, 64);
        t2 += __shfl_xor(t2, o, 64);
      }
      if (c < cend && (tid & (seg - 1)) == 0) {
        atomicAdd(stats_b + (c / cpg) * 2 + 0, t1);
        atomicAdd(stats_b + (c / cpg) * 2 + 1, t2);
      }
    }
  } else if (tid < BMC) {
    const int c = cb + tid;
    if (c < cend && (tid == 0 || (c % cpg) == 0)) {
      const int grp = c / cpg;
      float t1 = 0.f, t2 = 0.f;
      for (int cc = c; cc < cend && cc / cpg == grp; ++cc)
#pragma unroll
        for (int w = 0; w < WN; ++w) {
          t1 += sm_stat[(w * BMC + (cc - cb)) * 2 + 0];
          t2 += sm_stat[(w * BMC + (cc - cb)) * 2 + 1];
        }
      atomicAdd(stats_b + grp * 2 + 0, t1);
      atomicAdd(stats_b + grp * 2 + 1, t2);
    }
  }
}

// weights-stationary bf16 fast path (conv_ws.hip); returns SX_ERR_UNSUPPORTED-free "not handled" as 1
// WS_GATHER: the data gradient of ConvTranspose2d(2,2) as a 1x1 convolution over K = 4 taps x Cin gathered from the
// fine grid (output pixel (y, x), tap (ty, tx) reads input pixel (2y + ty, 2x + tx)).
// var: 0 plain; WS_DUAL: y2 = conv1x1(x, w2) beside the 3x3 (same input tile); WS_CENTER: the input has 2K channels and
// channels [K, 2K) only feed the centre tap with w2 (data gradient of conv1 + shortcut in one pass)
enum { WS_PLAIN = 0, WS_DUAL = 1, WS_CENTER = 2, WS_BS = 3, WS_GATHER = 4 };
int conv_ws_try(const ConvArgs& a, int mode, hipStream_t st, int var = WS_PLAIN);
int channel_sum_slots(int dtype, int B, int H, int W, int Hs, int Ws, int C, int C_out, const void* t, int ld, float* out, int slots, void* stream);
int conv3_shortcut_dgrad_slots(int dtype, int B, int H, int W, int C, int Cin, const void* dyr, int dyr_ld, const void* packed3_t,
                               const void* packed1_t, void* dx, int dx_ld, float* ch_sum, int ch_sum_n, int ch_slots, void* stream);
// sx_conv_forward_gn_bwd_sums with the option to leave the sums to the one-launch GroupNorm backward of small levels
// (conv_igemm.hip; `deferred` null = always form them)
int conv_forward_gn_bwd_sums(int dtype, int mode, int B, int H, int W, int Cin, int Cout, const void* x, int x_ld,
                             const void* packed, void* y, int y_ld, int gn_mode, const void* gn_y, int gn_y_ld,
                             const void* gn_s, int gn_s_ld, const float* stats, int groups, float eps,
                             const float* gamma, const float* beta, const float* film_g, const float* drop,
                             float* sums, int* deferred, void* stream);
// fp32 tensors, split-bf16 arithmetic (conv_x3.hip); var = WS_PLAIN or WS_CENTER; 1 = variant / shape not covered
int conv_x3_dispatch(const ConvArgs& a, hipStream_t st, int var = WS_PLAIN);

// Weight packing for a whole table of layers in one launch (conv_igemm.hip).
struct PackDesc {
  const float* w; void* out;
  int mode, tflip, Cout, Cin, RTot, RPad, KPad;
  int blk0, nblk;
  int64_t total;
};
static constexpr int PACK_MAX = 56;
// pad: optional zero-padded fp32 copy out[i] = i < n ? src[i] : 0 for i < npad <= 256 (the head's bias), done by block 0
struct PackPad { const float* src; float* dst; int n, npad; };
struct PackArgs { PackDesc d[PACK_MAX]; int n; PackPad pad; int split; };
int pack_desc_fill(int dtype, int mode, int transpose_flip, const float* w, int Cout, int Cin, void* packed, PackDesc* d);
int pack_weights_multi(int dtype, const PackDesc* descs, int n, hipStream_t st, const PackPad* pad = nullptr);
// The weight packs, the input pack and the FiLM generators of a training step's forward pass as block ranges of ONE launch
// (conv_igemm.hip: encode_head_kernel).  0 = launched, 1 = tables too large for the kernel-argument segment (run them separately).
struct FilmArgs;
struct EncodeHeadExtra {
  const FilmArgs& film;                                        // forward fields only (ngen may be 0)
  const float* x; int64_t sb, sc, sh, sw; void* xout;          // input pack
  int B, Cx, H, W, xld;
};
int encode_head_launch(int dtype, const PackDesc* descs, int n, const PackPad* pad, const EncodeHeadExtra& E, hipStream_t st);

}  // namespace sx
