// Implicit-GEMM convolution on gfx950 MFMA with an LDS-staged input halo.
//
// One kernel covers the four geometries of the U-Net (include/solpex_hip.h, sx_conv_mode):
//   3x3 pad 1, 1x1, ConvTranspose2d(2,2) fprop (a 1x1 GEMM with a pixel-shuffle store) and its data
//   gradient (a 4-tap gather).  Data gradients of the 3x3 / 1x1 convs are the same kernel run on
//   the transposed+flipped packed weights.
//
// GEMM orientation: A = packed weights (rows = output channels), B = input pixels (cols), so the
// MFMA result D[channel 4g+r][pixel l&15] leaves every lane with 4 CONSECUTIVE channels of one
// pixel -> one 8-byte (bf16) / 16-byte (fp32) NHWC store per tile, and per-channel GroupNorm
// partial sums are a 16-lane reduction.
//
// K is walked in 64-byte chunks (32 bf16 / 16 fp32 channels).  Per chunk a block stages
//   - the input halo tile  [(TH-1)*s+taph][(TW-1)*s+tapw] pixels x 64 B   and
//   - the weight slice     [ntaps][BMC rows] x 64 B
// into LDS (rows padded to 96 B: conflict-free for the ds_read_b128 fragment reads), then every
// wave runs ntaps x MI x NI MFMAs on fragments read straight from those two images.
#include "conv_common.h"
#include "norm_act.h"
#include "film.h"
#include <vector>
#include <stdlib.h>

namespace sx {

template <typename T, int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  typedef typename Elem<T>::Vec Vec;
  constexpr int EPL = Elem<T>::EPL, CK = Elem<T>::CK;
  constexpr int BMC = WM * MI * 16;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(WN * NI * 16 == 128, "128-pixel tile");
  extern __shared__ __align__(16) char smem[];
  char* sm_pix = smem;
  char* sm_w = smem + a.LH * a.LW * ROWB;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int n = lane & 15, g = lane >> 4;

  int bid = blockIdx.x;
  const int tx_i = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty_i = bid % a.tiles_y; bid /= a.tiles_y;
  const int b = bid;
  const int ty0 = ty_i * a.TH, tx0 = tx_i * a.TW;
  const int cb = blockIdx.y * BMC;

  int pix_off[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = (wn * NI + i) * 16 + n;
    const int py = m >> a.tw_log2, px = m & (a.TW - 1);
    pix_off[i] = ((py * a.in_scale) * a.LW + px * a.in_scale) * ROWB + g * 16;
  }
  int w_off[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) w_off[j] = ((wm * MI + j) * 16 + n) * ROWB + g * 16;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int j = 0; j < MI; ++j)
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
  const int npix_pieces = a.LH * a.LW * 4;
  const int nw_pieces = a.ntaps * BMC * 4;
  const int iy0 = ty0 * a.in_scale - a.pad, ix0 = tx0 * a.in_scale - a.pad;

  // fp32: the next K chunk's pieces are fetched into registers under this chunk's MFMAs (8 + 9 pieces per thread at most:
  // the transposed conv's 512-pixel gather image, nine taps x 64 rows of weights); coordinates are decoded once
  constexpr bool PREFETCH = sizeof(T) == 4;
  constexpr int NPX = PREFETCH ? 8 : 1, NPW = PREFETCH ? 9 : 1;
  Vec pre_x[NPX], pre_w[NPW];
  int64_t px_off[NPX], pw_off[NPW];      // element offsets without the chunk term; < 0 = piece reads as zero
  if constexpr (PREFETCH) {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int p = tid + i * 256, r = p >> 2, q = p & 3;
      const int ly = r / a.LW, lx = r - ly * a.LW;
      const int iy = iy0 + ly, ix = ix0 + lx;
      const bool ok = p < npix_pieces && iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx;
      px_off[i] = ok ? ((int64_t)(b * a.Hx + iy) * a.Wx + ix) * a.x_ld + q * EPL : -1;
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = tid + i * 256, r = p >> 2, q = p & 3;
      const int tap = r / BMC, row = r - tap * BMC;
      const bool ok = p < nw_pieces && cb + row < a.RTot;
      pw_off[i] = ok ? ((int64_t)tap * a.RTot + cb + row) * a.KPad + q * EPL : -1;
    }
  }
  auto fetch = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      Vec v = {};
      if (px_off[i] >= 0 && kc + ((tid + i * 256) & 3) * EPL < a.K) v = *reinterpret_cast<const Vec*>(xg + px_off[i] + kc);
      pre_x[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      Vec v = {};
      if (pw_off[i] >= 0) v = *reinterpret_cast<const Vec*>(wg + pw_off[i] + kc);
      pre_w[i] = v;
    }
  };
  if constexpr (PREFETCH) fetch(0);
  for (int kc = 0; kc < a.KPad; kc += CK) {
    __syncthreads();
    if constexpr (PREFETCH) {
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const int p = tid + i * 256;
        if (p < npix_pieces) *reinterpret_cast<Vec*>(sm_pix + (p >> 2) * ROWB + (p & 3) * 16) = pre_x[i];
      }
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        const int p = tid + i * 256;
        if (p < nw_pieces) *reinterpret_cast<Vec*>(sm_w + (p >> 2) * ROWB + (p & 3) * 16) = pre_w[i];
      }
      __syncthreads();
      if (kc + CK < a.KPad) fetch(kc + CK);
    } else {
    for (int p = tid; p < npix_pieces; p += 256) {
      const int r = p >> 2, q = p & 3;
      const int ly = r / a.LW, lx = r - ly * a.LW;
      const int iy = iy0 + ly, ix = ix0 + lx;
      const int c = kc + q * EPL;
      Vec v = {};
      if (iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx && c < a.K)
        v = *reinterpret_cast<const Vec*>(xg + ((int64_t)(b * a.Hx + iy) * a.Wx + ix) * a.x_ld + c);
      *reinterpret_cast<Vec*>(sm_pix + r * ROWB + q * 16) = v;
    }
    for (int p = tid; p < nw_pieces; p += 256) {
      const int r = p >> 2, q = p & 3;
      const int tap = r / BMC, row = r - tap * BMC;
      Vec v = {};
      if (cb + row < a.RTot) {
        if constexpr (sizeof(T) == 2) v = *reinterpret_cast<const Vec*>(wg + wpack_off_bf16(a.ntaps, a.KPad >> 5, tap, cb + row, kc + q * EPL));
        else v = *reinterpret_cast<const Vec*>(wg + ((int64_t)tap * a.RTot + cb + row) * a.KPad + kc + q * EPL);
      }
      *reinterpret_cast<Vec*>(sm_w + r * ROWB + q * 16) = v;
    }
    __syncthreads();
    }
    for (int tap = 0; tap < a.ntaps; ++tap) {
      const int tyy = tap / a.tapw, txx = tap - tyy * a.tapw;
      const int toff = (tyy * a.LW + txx) * ROWB;
      const char* wt = sm_w + tap * BMC * ROWB;
      Vec A[MI], Bf[NI];
#pragma unroll
      for (int j = 0; j < MI; ++j) A[j] = *reinterpret_cast<const Vec*>(wt + w_off[j]);
#pragma unroll
      for (int i = 0; i < NI; ++i) Bf[i] = *reinterpret_cast<const Vec*>(sm_pix + pix_off[i] + toff);
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[j][i] = Mma<T>::run(A[j], Bf[i], acc[j][i]);
    }
  }

  // ------------------------------------------------------------------ epilogue
  T* __restrict__ yg = reinterpret_cast<T*>(a.y);
  const bool do_stats = a.gn_stats != nullptr;
  float* sm_stat = reinterpret_cast<float*>(smem);   // [WN][BMC][2], reused after the K loop
  if (do_stats) __syncthreads();

#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int cl = (wm * MI + j) * 16 + 4 * g;   // channel within the block tile
    const int c0 = cb + cl;
    int co = c0, tap = 0;
    if (a.shuffle) { tap = c0 / a.RPad; co = c0 - tap * a.RPad; }
    const bool cvalid = (c0 < a.RTot) && (co < a.R);
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && cvalid) bias = *reinterpret_cast<const f32x4*>(a.bias + co);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int m = (wn * NI + i) * 16 + n;
      const int py = m >> a.tw_log2, px = m & (a.TW - 1);
      int oy = ty0 + py, ox = tx0 + px;
      const bool pvalid = (oy < a.H) && (ox < a.W);
      if (a.shuffle) { oy = 2 * oy + (tap >> 1); ox = 2 * ox + (tap & 1); }
      f32x4 v = acc[j][i] + bias;
      if (pvalid && cvalid) {
        T* dst = yg + ((int64_t)(b * a.Hy + oy) * a.Wy + ox) * a.y_ld + co;
        if (a.accumulate) v += load4(dst);
        store4(dst, v);
        if (do_stats) { s1 += v; s2 += v * v; }
      }
    }
    if (do_stats) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t1 = sum16(s1[r]), t2 = sum16(s2[r]);
        if (n == 0) {
          sm_stat[(wn * BMC + cl + r) * 2 + 0] = t1;
          sm_stat[(wn * BMC + cl + r) * 2 + 1] = t2;
        }
      }
    }
  }
  if (do_stats) {
    __syncthreads();
    gn_stats_flush<BMC, WN>(sm_stat, tid, cb, a.R, a.cpg, a.gn_stats + (int64_t)b * (a.R / a.cpg) * 2);
  }
}

template <typename T, int MI, int NI, int WM, int WN>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
  constexpr int BMC = WM * MI * 16;
  size_t smem = (size_t)(a.LH * a.LW + a.ntaps * BMC) * ROWB;
  size_t stat_bytes = (size_t)WN * BMC * 2 * sizeof(float);
  if (smem < stat_bytes) smem = stat_bytes;
  if (smem > 160 * 1024) { set_error("conv tile needs %zu B of LDS", smem); return SX_ERR_UNSUPPORTED; }
  if (sizeof(T) == 4 && (a.LH * a.LW * 4 > 8 * 256 || a.ntaps * BMC * 4 > 9 * 256)) {      // the kernel's register-prefetch slots
    set_error("conv: fp32 tile of %d pixels x %d taps exceeds the prefetch slots", a.LH * a.LW, a.ntaps);
    return SX_ERR_UNSUPPORTED;
  }
  auto kern = conv_igemm_kernel<T, MI, NI, WM, WN>;
  static std::atomic<unsigned long long> done{0};
  SX_TRY(ensure_max_lds(reinterpret_cast<const void*>(kern), done, "hipFuncSetAttribute(conv)"));
  dim3 grid(a.tiles_x * a.tiles_y * a.B, cdiv(a.RTot, BMC));
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a);
  SX_LAUNCH_CHECK("conv_igemm");
  return SX_OK;
}

template <typename T>
static int conv_dispatch(const ConvArgs& a, hipStream_t st) {
  if (a.RTot <= 16) return launch_conv<T, 1, 2, 1, 4>(a, st);
  if (a.RTot <= 32) return launch_conv<T, 2, 2, 1, 4>(a, st);
  return launch_conv<T, 2, 4, 2, 2>(a, st);
}

// ------------------------------------------------------------------ weight packing
// out[tap][row][k] for every layer of a descriptor table (one launch packs the whole network)
// one block (index `blk` inside the descriptor's D.nblk blocks) of the element-per-thread pack
template <typename T>
__device__ __forceinline__ void pack_weight_block(const PackDesc& D, int blk, int split_planes) {
  struct { int split; } A = {split_planes};
  const float* __restrict__ w = D.w;
  T* __restrict__ out = reinterpret_cast<T*>(D.out);
  const int mode = D.mode, tflip = D.tflip, Cout = D.Cout, Cin = D.Cin, RTot = D.RTot, RPad = D.RPad, KPad = D.KPad;
  for (int64_t idx = blk * (int64_t)blockDim.x + threadIdx.x; idx < D.total;
       idx += (int64_t)D.nblk * blockDim.x) {
    const int k = (int)(idx % KPad);
    const int64_t t2 = idx / KPad;
    const int row = (int)(t2 % RTot);
    const int tap = (int)(t2 / RTot);
    float v = 0.f;
    if (mode == SX_CONV3 || mode == SX_CONV1) {
      const int nt = (mode == SX_CONV3) ? 9 : 1;
      if (!tflip) {            // rows = Cout, k = Cin : w[co][ci][tap]
        if (row < Cout && k < Cin) v = w[((int64_t)row * Cin + k) * nt + tap];
      } else {                 // rows = Cin, k = Cout : w[co=k][ci=row][nt-1-tap]
        if (row < Cin && k < Cout) v = w[((int64_t)k * Cin + row) * nt + (nt - 1 - tap)];
      }
    } else if (mode == SX_CONVT_FWD) {   // w[ci][co][ty][tx]; rows = tp*RPad + co, k = ci
      const int tp = row / RPad, co = row - tp * RPad;
      if (co < Cout && k < Cin) v = w[((int64_t)k * Cout + co) * 4 + tp];
    } else {                             // SX_CONVT_BWD: rows = ci, k = co, taps = 4
      if (row < Cin && k < Cout) v = w[((int64_t)row * Cout + k) * 4 + tap];
    }
    if constexpr (sizeof(T) == 2) out[wpack_off_bf16((int)(D.total / ((int64_t)RTot * KPad)), KPad >> 5, tap, row, k)] = from_f32<T>(v);
    else if (A.split) {          // A.split = number of bf16 planes (2 or 3)
      bf16* o2 = reinterpret_cast<bf16*>(D.out) + wpack_off_x3(RTot, KPad >> 5, tap, row, k, A.split);
      float r = v;
      for (int q = 0; q < A.split; ++q) {
        const bf16 h = (bf16)r;
        o2[32 * q] = h;
        r -= (float)h;
      }
    } else out[idx] = from_f32<T>(v);
  }
}

template <typename T>
__global__ void pack_weight_kernel(PackArgs A) {
  if (blockIdx.x == 0 && A.pad.dst && (int)threadIdx.x < A.pad.npad)
    A.pad.dst[threadIdx.x] = (int)threadIdx.x < A.pad.n ? A.pad.src[threadIdx.x] : 0.f;
  int di = 0;
  while (di + 1 < A.n && (int)blockIdx.x >= A.d[di + 1].blk0) ++di;
  pack_weight_block<T>(A.d[di], (int)blockIdx.x - A.d[di].blk0, A.split);
}

// Forward pack [tap][co][ci] and transposed + tap-flipped pack [nt-1-tap][ci][co] of one 3x3 / 1x1 weight from ONE read:
// a block stages a 32 (co) x 32 (ci) x taps tile in LDS (coalesced runs of 32*taps floats per co) and writes both
// orientations in 64-byte runs.  The element-per-thread kernel above gathered with a stride of taps floats (forward) or
// Cin*taps floats (transposed) and took 26 us per step for 9 MB of output.
struct PackPair {
  const float* w; void* out_f; void* out_t;
  int Cout, Cin, nt;
  int RTot_f, KPad_f, RTot_t, KPad_t;
  int blk0, tiles_ci;
};
static constexpr int PAIR_MAX = 32;
struct PackPairArgs { PackPair d[PAIR_MAX]; int n; int split; };

// 8 consecutive k of one row as `planes` runs of 8 bf16 (wpack_off_x3)
__device__ __forceinline__ void store8_split(bf16* p, const f32x8& r, int planes) {
  f32x8 rest = r;
  for (int q = 0; q < planes; ++q) {
    bf16x8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) { h[i] = (bf16)rest.v[i]; rest.v[i] -= (float)h[i]; }
    *reinterpret_cast<bf16x8*>(p + 32 * q) = h;
  }
}

template <typename T, int NT>
__device__ __forceinline__ void pack_pair_tile(const PackPair& D, int co0, int ci0, float (*tile)[32 * 9 + 1], int split) {
  constexpr int RUN = 32 * NT, TOTAL = 32 * RUN, U = NT == 9 ? 36 : 4;
  // U independent loads in flight per thread (a block is 36 trips of one load otherwise, on ~1 block per CU): all 36 of a
  // 3x3 tile at once (17.2 us per step with U = 4, 14.5 with 12, 13.3 with 36)
  for (int e0 = threadIdx.x; e0 < TOTAL; e0 += 256 * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * 256;
      const int col = e / RUN, r = e - col * RUN;
      const int co = co0 + col, ci = ci0 + r / NT;
      v[u] = (e < TOTAL && co < D.Cout && ci < D.Cin) ? D.w[((int64_t)co * D.Cin + ci0) * NT + r] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * 256;
      if (e < TOTAL) tile[e / RUN][e % RUN] = v[u];
    }
  }
  __syncthreads();
  T* __restrict__ of = reinterpret_cast<T*>(D.out_f);
  T* __restrict__ ot = reinterpret_cast<T*>(D.out_t);
  // 8 consecutive k per thread = one 16-byte (bf16) store; KPad is a multiple of 16 / 32, so a group is all in or all out
  for (int e = threadIdx.x; e < NT * 128; e += 256) {
    const int tap = e >> 7, hi = (e >> 2) & 31, lo8 = (e & 3) * 8;
    f32x8 v;
    {   // forward: row co0 + hi, k = ci0 + lo8 .. +7
      const int co = co0 + hi, ci = ci0 + lo8;
      if (co < D.RTot_f && ci < D.KPad_f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v.v[i] = tile[hi][(lo8 + i) * NT + tap];
        if constexpr (sizeof(T) == 2) store8(of + wpack_off_bf16(NT, D.KPad_f >> 5, tap, co, ci), v);
        else if (split) store8_split(reinterpret_cast<bf16*>(D.out_f) + wpack_off_x3(D.RTot_f, D.KPad_f >> 5, tap, co, ci, split), v, split);
        else store8(of + ((int64_t)tap * D.RTot_f + co) * D.KPad_f + ci, v);
      }
    }
    {   // transposed: row ci0 + hi, k = co0 + lo8 .. +7, taps reversed
      const int ci = ci0 + hi, co = co0 + lo8;
      if (ci < D.RTot_t && co < D.KPad_t) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v.v[i] = tile[lo8 + i][hi * NT + tap];
        if constexpr (sizeof(T) == 2) store8(ot + wpack_off_bf16(NT, D.KPad_t >> 5, NT - 1 - tap, ci, co), v);
        else if (split) store8_split(reinterpret_cast<bf16*>(D.out_t) + wpack_off_x3(D.RTot_t, D.KPad_t >> 5, NT - 1 - tap, ci, co, split), v, split);
        else store8(ot + ((int64_t)(NT - 1 - tap) * D.RTot_t + ci) * D.KPad_t + co, v);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_pair_kernel(PackPairArgs A) {
  __shared__ float tile[32][32 * 9 + 1];
  int di = 0;
  while (di + 1 < A.n && (int)blockIdx.x >= A.d[di + 1].blk0) ++di;
  const PackPair& D = A.d[di];
  const int tb = (int)blockIdx.x - D.blk0;
  const int co0 = (tb / D.tiles_ci) * 32, ci0 = (tb % D.tiles_ci) * 32;
  if (D.nt == 9) pack_pair_tile<T, 9>(D, co0, ci0, tile, A.split);
  else pack_pair_tile<T, 1>(D, co0, ci0, tile, A.split);
}

// ------------------------------------------------------------------ the head of a training step's forward pass in ONE launch
// Weight packs (pair tiles + the element-per-thread rest + the head's bias pad), the input pack and every FiLM generator are
// mutually independent and each too small to fill the chip (277 + 832 + 2048 + 448 blocks of 5-14 us): as four launches they
// cost their sum, as block ranges of one grid about the longest of them.  Tables sized to stay inside the 4 KB kernel-argument
// segment; networks that exceed them take the separate launches.
static constexpr int HEAD_PAIRS = 24, HEAD_REST = 8, HEAD_GEN = 8;
struct HeadFilm { const float* w1; const float* b1; const float* w2; const float* b2; float* gf; float* bf; float* pre; int C; };
struct HeadArgs {
  PackPair pair[HEAD_PAIRS];
  PackDesc rest[HEAD_REST];
  HeadFilm gen[HEAD_GEN];
  PackPad pad;
  const float* x; int64_t sb, sc, sh, sw; void* xout;          // input pack
  const float* fparams; float* zero;                           // FiLM: [B][P] inputs; words the launch also clears
  int n_pair, n_rest, ngen, split;
  int B, Cx, H, W, xld, P, Hf, zero_n;
  int blk_pair, blk_film, blk_rest, blk_in;                    // block counts of the four roles, in this order
};
static_assert(sizeof(HeadArgs) <= 4096, "kernel-argument segment");

template <typename T>
__global__ __launch_bounds__(256) void encode_head_kernel(HeadArgs A) {
  __shared__ float tile[32][32 * 9 + 1];
  int blk = blockIdx.x;
  if (blk < A.blk_pair) {
    int di = 0;
    while (di + 1 < A.n_pair && blk >= A.pair[di + 1].blk0) ++di;
    const PackPair& D = A.pair[di];
    const int tb = blk - D.blk0;
    const int co0 = (tb / D.tiles_ci) * 32, ci0 = (tb % D.tiles_ci) * 32;
    if (D.nt == 9) pack_pair_tile<T, 9>(D, co0, ci0, tile, A.split);
    else pack_pair_tile<T, 1>(D, co0, ci0, tile, A.split);
    return;
  }
  blk -= A.blk_pair;
  if (blk < A.blk_film) {
    float* hid = &tile[0][0];                  // FILM_MAX_HID floats
    float* pv = hid + FILM_MAX_HID;
    if (A.zero) {
      const int nthr = A.blk_film * blockDim.x;
      for (int i = blk * blockDim.x + threadIdx.x; i < A.zero_n; i += nthr) A.zero[i] = 0.f;
    }
    const int gi = blk / A.B, b = blk - gi * A.B;
    const HeadFilm& G = A.gen[gi];
    film_fwd_block(G.w1, G.b1, G.w2, G.b2, G.gf, G.bf, G.pre, G.C, b, A.P, A.Hf, A.fparams, hid, pv);
    return;
  }
  blk -= A.blk_film;
  if (blk < A.blk_rest) {
    if (blk == 0 && A.pad.dst && (int)threadIdx.x < A.pad.npad)
      A.pad.dst[threadIdx.x] = (int)threadIdx.x < A.pad.n ? A.pad.src[threadIdx.x] : 0.f;
    int di = 0;
    while (di + 1 < A.n_rest && blk >= A.rest[di + 1].blk0) ++di;
    pack_weight_block<T>(A.rest[di], blk - A.rest[di].blk0, A.split);
    return;
  }
  blk -= A.blk_rest;
  {   // input pack: (B, C, H, W) fp32 with element strides -> NHWC T, xld channels, zero padded (misc.hip: pack_input_kernel)
    const int groups = A.xld >> 3;
    const int64_t total = (int64_t)A.B * A.H * A.W * groups;
    T* out = reinterpret_cast<T*>(A.xout);
    for (int64_t i = blk * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)A.blk_in * blockDim.x) {
      const int c0 = (int)(i % groups) << 3;
      int64_t q = i / groups;
      const int w = (int)(q % A.W); q /= A.W;
      const int h = (int)(q % A.H);
      const int b = (int)(q / A.H);
      const float* src = A.x + b * A.sb + h * A.sh + w * A.sw;
      f32x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v.v[e] = c0 + e < A.Cx ? src[(c0 + e) * A.sc] : 0.f;
      store8(out + (i << 3), v);
    }
  }
}

struct PackGeom { int ntaps, R, K, RPad, RTot, KPad; };
static PackGeom pack_geom(int mode, int dtype, int tflip, int Cout, int Cin) {
  const int CK = 32;       // both tiers pad K to 32 (fp32: the split form walks 32-channel chunks, the exact form 16)
  PackGeom p;
  if (mode == SX_CONV3 || mode == SX_CONV1) {
    p.ntaps = mode == SX_CONV3 ? 9 : 1;
    p.R = tflip ? Cin : Cout;
    p.K = tflip ? Cout : Cin;
    p.RPad = round_up(p.R, 16);
    p.RTot = p.RPad;
  } else if (mode == SX_CONVT_FWD) {
    p.ntaps = 1; p.R = Cout; p.K = Cin; p.RPad = round_up(Cout, 16); p.RTot = 4 * p.RPad;
  } else {
    p.ntaps = 4; p.R = Cin; p.K = Cout; p.RPad = round_up(Cin, 16); p.RTot = p.RPad;
  }
  p.KPad = round_up(p.K, CK);
  return p;
}

}  // namespace sx

using namespace sx;

extern "C" int64_t sx_packed_weight_elems(int mode, int dtype, int c_rows, int c_k) {
  // c_rows / c_k are the GEMM's output / reduction channel counts (after any transpose).
  const int CK = 32;
  const int ntaps = mode == SX_CONV3 ? 9 : (mode == SX_CONVT_BWD ? 4 : 1);
  const int rtot = round_up(c_rows, 16) * (mode == SX_CONVT_FWD ? 4 : 1);
  // fp32: room for the three-plane split form (192 bytes per 32 k, conv_x3.hip), whatever arithmetic mode is in force
  const int64_t n = (int64_t)ntaps * rtot * round_up(c_k, CK);
  return dtype == SX_F32 ? n + n / 2 : n;
}

namespace sx {
int pack_desc_fill(int dtype, int mode, int transpose_flip, const float* w, int Cout, int Cin, void* packed, PackDesc* d) {
  SX_REQUIRE(w && packed, "sx_pack_conv_weight: null pointer");
  SX_REQUIRE(mode >= SX_CONV3 && mode <= SX_CONVT_BWD, "sx_pack_conv_weight: bad mode %d", mode);
  SX_REQUIRE(dtype == SX_F32 || dtype == SX_BF16, "sx_pack_conv_weight: bad dtype %d", dtype);
  PackGeom p = pack_geom(mode, dtype, transpose_flip, Cout, Cin);
  d->w = w; d->out = packed; d->mode = mode; d->tflip = transpose_flip; d->Cout = Cout; d->Cin = Cin;
  d->RTot = p.RTot; d->RPad = p.RPad; d->KPad = p.KPad;
  d->total = (int64_t)p.ntaps * p.RTot * p.KPad;
  d->blk0 = 0; d->nblk = 0;
  return SX_OK;
}

static void pack_split_pairs(const PackDesc* descs, int n, std::vector<PackPair>& pairs, std::vector<PackDesc>& rest);

// 0 = launched; 1 = the tables do not fit (the caller takes the separate launches)
int encode_head_launch(int dtype, const PackDesc* descs, int n, const PackPad* pad, const EncodeHeadExtra& E, hipStream_t st) {
  static const int on = env_int("SX_HEAD_MERGED", 1);        // A/B knob
  std::vector<PackDesc> rest;
  std::vector<PackPair> pairs;
  pack_split_pairs(descs, n, pairs, rest);
  if (!on || pairs.size() > (size_t)HEAD_PAIRS || rest.size() > (size_t)HEAD_REST || rest.empty() || E.film.ngen > HEAD_GEN ||
      E.film.Hf > FILM_MAX_HID || E.film.P > 64 || E.xld % 8)
    return 1;
  for (int i = 0; i < E.film.ngen; ++i)
    if (E.film.g[i].C > 1024) return 1;
  HeadArgs A;
  memset(&A, 0, sizeof(A));
  A.split = dtype == SX_F32 ? fp32_planes() : 0;
  A.n_pair = (int)pairs.size(); A.n_rest = (int)rest.size(); A.ngen = E.film.ngen;
  int blk = 0;
  for (int i = 0; i < A.n_pair; ++i) {
    A.pair[i] = pairs[i];
    const int tiles_co = cdiv(std::max(A.pair[i].RTot_f, A.pair[i].KPad_t), 32);
    A.pair[i].tiles_ci = cdiv(std::max(A.pair[i].KPad_f, A.pair[i].RTot_t), 32);
    A.pair[i].blk0 = blk;
    blk += tiles_co * A.pair[i].tiles_ci;
  }
  A.blk_pair = blk;
  A.B = E.B;
  A.blk_film = E.film.ngen * E.B;
  for (int i = 0; i < A.ngen; ++i) {
    const FilmGenDesc& g = E.film.g[i];
    A.gen[i].w1 = g.w1; A.gen[i].b1 = g.b1; A.gen[i].w2 = g.w2; A.gen[i].b2 = g.b2;
    A.gen[i].gf = g.gf; A.gen[i].bf = g.bf; A.gen[i].pre = g.pre; A.gen[i].C = g.C;
  }
  A.P = E.film.P; A.Hf = E.film.Hf; A.fparams = E.film.params; A.zero = E.film.zero; A.zero_n = E.film.zero_n;
  blk = 0;
  for (int i = 0; i < A.n_rest; ++i) {
    A.rest[i] = rest[i];
    A.rest[i].blk0 = blk;
    A.rest[i].nblk = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(A.rest[i].total, 256), 256));
    blk += A.rest[i].nblk;
  }
  A.blk_rest = blk;
  if (pad) A.pad = *pad;
  A.x = E.x; A.sb = E.sb; A.sc = E.sc; A.sh = E.sh; A.sw = E.sw; A.xout = E.xout;
  A.Cx = E.Cx; A.H = E.H; A.W = E.W; A.xld = E.xld;
  const int64_t items = (int64_t)E.B * E.H * E.W * (E.xld / 8);
  A.blk_in = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(items, 256), 2048));
  const int grid = A.blk_pair + A.blk_film + A.blk_rest + A.blk_in;
  if (dtype == SX_BF16) hipLaunchKernelGGL(encode_head_kernel<bf16>, dim3(grid), dim3(256), 0, st, A);
  else hipLaunchKernelGGL(encode_head_kernel<float>, dim3(grid), dim3(256), 0, st, A);
  SX_LAUNCH_CHECK("encode_head");
  return SX_OK;
}

static void pack_split_pairs(const PackDesc* descs, int n, std::vector<PackPair>& pairs, std::vector<PackDesc>& rest) {
  for (int i = 0; i < n; ++i) {
    const PackDesc& f = descs[i];
    if (i + 1 < n && (f.mode == SX_CONV3 || f.mode == SX_CONV1) && !f.tflip && descs[i + 1].tflip && descs[i + 1].w == f.w &&
        descs[i + 1].mode == f.mode && descs[i + 1].Cout == f.Cout && descs[i + 1].Cin == f.Cin) {
      const PackDesc& t = descs[i + 1];
      PackPair q;
      q.w = f.w; q.out_f = f.out; q.out_t = t.out; q.Cout = f.Cout; q.Cin = f.Cin; q.nt = f.mode == SX_CONV3 ? 9 : 1;
      q.RTot_f = f.RTot; q.KPad_f = f.KPad; q.RTot_t = t.RTot; q.KPad_t = t.KPad;
      q.blk0 = 0; q.tiles_ci = 0;
      pairs.push_back(q);
      ++i;
    } else {
      rest.push_back(f);
    }
  }
}

int pack_weights_multi(int dtype, const PackDesc* descs, int n, hipStream_t st, const PackPad* pad) {
  // adjacent (forward, transposed) descriptors of the same 3x3 / 1x1 weight go through the tile kernel, one read for both
  std::vector<PackDesc> rest;
  std::vector<PackPair> pairs;
  pack_split_pairs(descs, n, pairs, rest);
  for (size_t i0 = 0; i0 < pairs.size(); i0 += PAIR_MAX) {
    PackPairArgs A;
    A.split = dtype == SX_F32 ? fp32_planes() : 0;
    A.n = (int)std::min<size_t>(PAIR_MAX, pairs.size() - i0);
    int blk = 0;
    for (int i = 0; i < A.n; ++i) {
      A.d[i] = pairs[i0 + i];
      const int tiles_co = cdiv(std::max(A.d[i].RTot_f, A.d[i].KPad_t), 32);
      A.d[i].tiles_ci = cdiv(std::max(A.d[i].KPad_f, A.d[i].RTot_t), 32);
      A.d[i].blk0 = blk;
      blk += tiles_co * A.d[i].tiles_ci;
    }
    if (dtype == SX_BF16) hipLaunchKernelGGL(pack_pair_kernel<bf16>, dim3(blk), dim3(256), 0, st, A);
    else hipLaunchKernelGGL(pack_pair_kernel<float>, dim3(blk), dim3(256), 0, st, A);
    SX_LAUNCH_CHECK("pack_pair");
  }
  descs = rest.data();
  n = (int)rest.size();
  PackDesc nothing;
  memset(&nothing, 0, sizeof(nothing));
  if (n == 0 && pad) { descs = &nothing; n = 1; }       // the padded copy still needs its launch
  for (int i0 = 0; i0 < n; i0 += PACK_MAX) {
    PackArgs A;
    memset(&A.pad, 0, sizeof(A.pad));
    A.split = dtype == SX_F32 ? fp32_planes() : 0;
    if (pad && i0 == 0) A.pad = *pad;
    A.n = std::min(PACK_MAX, n - i0);
    int blk = 0;
    for (int i = 0; i < A.n; ++i) {
      A.d[i] = descs[i0 + i];
      A.d[i].blk0 = blk;
      A.d[i].nblk = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(A.d[i].total, 256), 256));
      blk += A.d[i].nblk;
    }
    if (dtype == SX_BF16) hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(blk), dim3(256), 0, st, A);
    else hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blk), dim3(256), 0, st, A);
    SX_LAUNCH_CHECK("pack_weight");
  }
  return SX_OK;
}
}  // namespace sx

extern "C" int sx_pack_conv_weight(int dtype, int mode, int transpose_flip, const float* w, int Cout, int Cin,
                                   void* packed, void* stream) {
  PackDesc d;
  SX_TRY(pack_desc_fill(dtype, mode, transpose_flip, w, Cout, Cin, packed, &d));
  return pack_weights_multi(dtype, &d, 1, (hipStream_t)stream);
}

extern "C" int sx_conv_forward(int dtype, int mode, int B, int H, int W, int Cin, int Cout,
                               const void* x, int x_ld, int Hx, int Wx, const void* packed, const float* bias,
                               void* y, int y_ld, int Hy, int Wy, int accumulate, float* gn_stats, int gn_groups,
                               void* stream) {
  SX_REQUIRE(x && packed && y, "sx_conv_forward: null pointer");
  SX_REQUIRE(dtype == SX_F32 || dtype == SX_BF16, "sx_conv_forward: bad dtype %d", dtype);
  SX_REQUIRE(mode >= SX_CONV3 && mode <= SX_CONVT_BWD, "sx_conv_forward: bad mode %d", mode);
  SX_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "sx_conv_forward: empty problem");
  const int EPL = dtype == SX_BF16 ? 8 : 4, CK = 32;
  if (Cin % EPL || x_ld % EPL || Cout % 4 || y_ld % 4) {
    set_error("sx_conv_forward: channel counts must be multiples of %d (in) / 4 (out): Cin=%d x_ld=%d Cout=%d y_ld=%d",
              EPL, Cin, x_ld, Cout, y_ld);
    return SX_ERR_UNSUPPORTED;
  }
  SX_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)y % 8) == 0,
             "sx_conv_forward: misaligned pointer");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = packed; a.y = y; a.bias = bias; a.gn_stats = gn_stats;
  a.B = B; a.H = H; a.W = W; a.Hx = Hx; a.Wx = Wx; a.x_ld = x_ld; a.Hy = Hy; a.Wy = Wy; a.y_ld = y_ld;
  a.K = Cin; a.KPad = round_up(Cin, CK);
  a.R = Cout; a.RPad = round_up(Cout, 16); a.RTot = a.RPad;
  a.accumulate = accumulate;
  switch (mode) {
    case SX_CONV3: a.ntaps = 9; a.tapw = 3; a.in_scale = 1; a.pad = 1; break;
    case SX_CONV1: a.ntaps = 1; a.tapw = 1; a.in_scale = 1; a.pad = 0; break;
    case SX_CONVT_FWD: a.ntaps = 1; a.tapw = 1; a.in_scale = 1; a.pad = 0; a.shuffle = 1; a.RTot = 4 * a.RPad; break;
    default: a.ntaps = 4; a.tapw = 2; a.in_scale = 2; a.pad = 0; break;
  }
  if (mode == SX_CONVT_FWD)
    SX_REQUIRE(Hx >= H && Wx >= W && Hy >= 2 * H && Wy >= 2 * W, "sx_conv_forward: convT fwd buffer dims");
  else if (mode == SX_CONVT_BWD)
    SX_REQUIRE(Hx >= 2 * H && Wx >= 2 * W && Hy >= H && Wy >= W, "sx_conv_forward: convT bwd buffer dims");
  else
    SX_REQUIRE(Hx == H && Wx == W && Hy >= H && Wy >= W, "sx_conv_forward: buffer dims");
  if (gn_stats) {
    SX_REQUIRE(gn_groups > 0 && Cout % gn_groups == 0 && !a.shuffle && !accumulate,
               "sx_conv_forward: bad GroupNorm statistics request");
    a.cpg = Cout / gn_groups;
  }
  a.tw_log2 = W > 8 ? 4 : (W > 4 ? 3 : 2);
  a.TW = 1 << a.tw_log2;
  a.TH = 128 / a.TW;
  const int taph = a.ntaps / a.tapw;
  a.LW = (a.TW - 1) * a.in_scale + a.tapw;
  a.LH = (a.TH - 1) * a.in_scale + taph;
  a.tiles_x = cdiv(W, a.TW);
  a.tiles_y = cdiv(H, a.TH);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16) {
    const int rc = conv_ws_try(a, mode, st);   // 0 = launched, 1 = shape not covered, <0 = error
    if (rc <= 0) return rc;
  }
  if (dtype == SX_F32 && fp32_split()) return conv_x3_dispatch(a, st);
  return dtype == SX_BF16 ? conv_dispatch<bf16>(a, st) : conv_dispatch<float>(a, st);
}

// y = conv(x) where y is a gradient entering GroupNorm + SiLU backward: phase 1 of that backward in the epilogue
namespace sx {
// `deferred` (may be null): set to 1 when the sums were NOT formed because sx_gn_silu_backward will run both phases of this
// shape in one launch (act_bwd_group_kernel, norm_act.hip); the caller then must not pass SX_GN_BWD_SUMS_READY
int conv_forward_gn_bwd_sums(int dtype, int mode, int B, int H, int W, int Cin, int Cout, const void* x, int x_ld,
                             const void* packed, void* y, int y_ld, int gn_mode, const void* gn_y, int gn_y_ld,
                             const void* gn_s, int gn_s_ld, const float* stats, int groups, float eps,
                             const float* gamma, const float* beta, const float* film_g, const float* drop,
                             float* sums, int* deferred, void* stream) {
  if (deferred) *deferred = 0;
  SX_REQUIRE(x && packed && y && gn_y && stats && gamma && beta && sums, "sx_conv_forward_gn_bwd_sums: null pointer");
  SX_REQUIRE(mode == SX_CONV3 || mode == SX_CONV1, "sx_conv_forward_gn_bwd_sums: 3x3 / 1x1 only");
  SX_REQUIRE(gn_mode == 0 || (gn_mode == 2 && gn_s), "sx_conv_forward_gn_bwd_sums: gn_mode 0, or 2 with the shortcut branch");
  SX_REQUIRE(groups > 0 && Cout % groups == 0, "sx_conv_forward_gn_bwd_sums: groups %d do not divide %d channels", groups, Cout);
  // where the fused epilogue is not available, small levels leave the sums to the one-launch backward (both phases).  (Preferring
  // that over the fused epilogue where both exist -- the 16x8 level -- measured the same step time: 1.6223 / 1.6225 ms.)
  const bool defer_ok = deferred && gn_mode == 0 && gn_bwd_group_fused_ok(B, H * W, Cout, groups);
  if (dtype == SX_BF16 && Cin % 8 == 0 && x_ld % 8 == 0 && Cout % 4 == 0 && y_ld % 4 == 0 && B > 0 && H > 0 && W > 0 &&
      ((uintptr_t)x % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)gn_y % 8) == 0 &&
      (!gn_s || ((uintptr_t)gn_s % 8) == 0)) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = packed; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Hx = H; a.Wx = W; a.x_ld = x_ld; a.Hy = H; a.Wy = W; a.y_ld = y_ld;
    a.K = Cin; a.KPad = round_up(Cin, 32);
    a.R = Cout; a.RPad = round_up(Cout, 16); a.RTot = a.RPad;
    if (mode == SX_CONV3) { a.ntaps = 9; a.tapw = 3; a.in_scale = 1; a.pad = 1; }
    else { a.ntaps = 1; a.tapw = 1; a.in_scale = 1; a.pad = 0; }
    a.cpg = Cout / groups;
    a.bs_mode = gn_mode == 2 ? 2 : 1;
    a.bs_y = gn_y; a.bs_y_ld = gn_y_ld; a.bs_s = gn_s; a.bs_s_ld = gn_s_ld;
    a.bs_stats = stats; a.bs_gamma = gamma; a.bs_beta = beta; a.bs_gf = film_g; a.bs_drop = drop; a.bs_sums = sums;
    a.bs_eps = eps; a.bs_inv_n = 1.0f / ((float)a.cpg * (float)H * (float)W);
    const int rc = conv_ws_try(a, mode, (hipStream_t)stream, WS_BS);
    if (rc <= 0) return rc;
  }
  SX_TRY(sx_conv_forward(dtype, mode, B, H, W, Cin, Cout, x, x_ld, H, W, packed, nullptr, y, y_ld, H, W, 0, nullptr, 0, stream));
  if (defer_ok) { *deferred = 1; return SX_OK; }
  return gn_bwd_reduce_only(dtype, gn_mode, B, H * W, Cout, groups, eps, gn_y, gn_y_ld, stats, gamma, beta, y, y_ld, gn_s, gn_s_ld,
                            film_g, drop, sums, (hipStream_t)stream);
}
}  // namespace sx

extern "C" int sx_conv_forward_gn_bwd_sums(int dtype, int mode, int B, int H, int W, int Cin, int Cout, const void* x, int x_ld,
                                           const void* packed, void* y, int y_ld, int gn_mode, const void* gn_y, int gn_y_ld,
                                           const void* gn_s, int gn_s_ld, const float* stats, int groups, float eps,
                                           const float* gamma, const float* beta, const float* film_g, const float* drop,
                                           float* sums, void* stream) {
  return sx::conv_forward_gn_bwd_sums(dtype, mode, B, H, W, Cin, Cout, x, x_ld, packed, y, y_ld, gn_mode, gn_y, gn_y_ld, gn_s, gn_s_ld,
                                      stats, groups, eps, gamma, beta, film_g, drop, sums, nullptr, stream);
}

// ------------------------------------------------------------------ DoubleConv fusions (models.py:60-96)
// conv1 (3x3, + GroupNorm statistics) and the 1x1 shortcut read the same input: one pass over x on the fast path.
extern "C" int sx_conv3_shortcut_forward(int dtype, int B, int H, int W, int Cin, int Cout, const void* x, int x_ld,
                                         const void* packed3, const void* packed1, void* y, int y_ld, float* gn_stats,
                                         int gn_groups, void* s, int s_ld, void* stream) {
  SX_REQUIRE(x && packed3 && packed1 && y && s, "sx_conv3_shortcut_forward: null pointer");
  if (dtype == SX_BF16 && Cin % 8 == 0 && x_ld % 8 == 0 && Cout % 4 == 0 && y_ld % 4 == 0 && s_ld % 4 == 0 && B > 0 && H > 0 && W > 0 &&
      ((uintptr_t)x % 16) == 0 && ((uintptr_t)packed3 % 16) == 0 && ((uintptr_t)packed1 % 16) == 0 && ((uintptr_t)y % 8) == 0 &&
      ((uintptr_t)s % 8) == 0 && (!gn_stats || (gn_groups > 0 && Cout % gn_groups == 0))) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = packed3; a.y = y; a.gn_stats = gn_stats; a.w2 = packed1; a.y2 = s; a.y2_ld = s_ld;
    a.B = B; a.H = H; a.W = W; a.Hx = H; a.Wx = W; a.x_ld = x_ld; a.Hy = H; a.Wy = W; a.y_ld = y_ld;
    a.K = Cin; a.KPad = round_up(Cin, 32);
    a.R = Cout; a.RPad = round_up(Cout, 16); a.RTot = a.RPad;
    a.ntaps = 9; a.tapw = 3; a.in_scale = 1; a.pad = 1;
    if (gn_stats) a.cpg = Cout / gn_groups;
    const int rc = conv_ws_try(a, SX_CONV3, (hipStream_t)stream, WS_DUAL);
    if (rc <= 0) return rc;
  }
  SX_TRY(sx_conv_forward(dtype, SX_CONV3, B, H, W, Cin, Cout, x, x_ld, H, W, packed3, nullptr, y, y_ld, H, W, 0, gn_stats, gn_groups, stream));
  return sx_conv_forward(dtype, SX_CONV1, B, H, W, Cin, Cout, x, x_ld, H, W, packed1, nullptr, s, s_ld, H, W, 0, nullptr, 0, stream);
}

// dx = conv3x3^T(dy) + conv1x1^T(dr), dy = dyr[..., 0:C], dr = dyr[..., C:2C] (one buffer, leading dimension dyr_ld).
// packed3_t / packed1_t: the transposed (+ flipped) packs, rows = Cin of the forward conv, K = C.
// ch_sum (nullable): ch_sum[c] += sum over every pixel of dx[..., c] for c < ch_sum_n (fp32 atomics) -- when the block
// input is a [transposed-conv output | skip] concatenation that IS the transposed convolution's bias gradient.
extern "C" int sx_conv3_shortcut_dgrad(int dtype, int B, int H, int W, int C, int Cin, const void* dyr, int dyr_ld,
                                       const void* packed3_t, const void* packed1_t, void* dx, int dx_ld, float* ch_sum,
                                       int ch_sum_n, void* stream) {
  return sx::conv3_shortcut_dgrad_slots(dtype, B, H, W, C, Cin, dyr, dyr_ld, packed3_t, packed1_t, dx, dx_ld, ch_sum, ch_sum_n, 1, stream);
}

// ch_slots: ch_sum is [ch_slots][ch_sum_n]; the fused epilogue adds into row 0, a separate channel-sum launch spreads its
// blocks over all rows (channel_sum_slots)
int sx::conv3_shortcut_dgrad_slots(int dtype, int B, int H, int W, int C, int Cin, const void* dyr, int dyr_ld,
                                   const void* packed3_t, const void* packed1_t, void* dx, int dx_ld, float* ch_sum,
                                   int ch_sum_n, int ch_slots, void* stream) {
  SX_REQUIRE(dyr && packed3_t && packed1_t && dx, "sx_conv3_shortcut_dgrad: null pointer");
  SX_REQUIRE(dyr_ld >= 2 * C, "sx_conv3_shortcut_dgrad: dyr must hold 2*C channels per pixel");
  if (dtype == SX_BF16 && C % 32 == 0 && dyr_ld % 8 == 0 && Cin % 4 == 0 && dx_ld % 4 == 0 && B > 0 && H > 0 && W > 0 &&
      ((uintptr_t)dyr % 16) == 0 && ((uintptr_t)packed3_t % 16) == 0 && ((uintptr_t)packed1_t % 16) == 0 && ((uintptr_t)dx % 8) == 0) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = dyr; a.w = packed3_t; a.y = dx; a.w2 = packed1_t;
    a.B = B; a.H = H; a.W = W; a.Hx = H; a.Wx = W; a.x_ld = dyr_ld; a.Hy = H; a.Wy = W; a.y_ld = dx_ld;
    a.K = C; a.KPad = C;
    a.R = Cin; a.RPad = round_up(Cin, 16); a.RTot = a.RPad;
    a.ntaps = 9; a.tapw = 3; a.in_scale = 1; a.pad = 1;
    a.ch_sum = ch_sum; a.ch_sum_n = ch_sum ? ch_sum_n : 0;
    const int rc = conv_ws_try(a, SX_CONV3, (hipStream_t)stream, WS_CENTER);
    if (rc <= 0) return rc;
  }
  if (dtype == SX_F32 && fp32_split() && C % 32 == 0 && dyr_ld % 4 == 0 && Cin % 4 == 0 && dx_ld % 4 == 0 && B > 0 && H > 0 && W > 0 &&
      ((uintptr_t)dyr % 16) == 0 && ((uintptr_t)packed3_t % 16) == 0 && ((uintptr_t)packed1_t % 16) == 0 && ((uintptr_t)dx % 16) == 0) {
    // the same joint form in split-bf16 arithmetic: one pass, no read-modify-write of dx
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = dyr; a.w = packed3_t; a.y = dx; a.w2 = packed1_t;
    a.B = B; a.H = H; a.W = W; a.Hx = H; a.Wx = W; a.x_ld = dyr_ld; a.Hy = H; a.Wy = W; a.y_ld = dx_ld;
    a.K = C; a.KPad = C;
    a.R = Cin; a.RPad = round_up(Cin, 16); a.RTot = a.RPad;
    a.ntaps = 9; a.tapw = 3; a.in_scale = 1; a.pad = 1;
    a.tw_log2 = W > 8 ? 4 : (W > 4 ? 3 : 2);
    a.TW = 1 << a.tw_log2;
    a.LW = a.TW + 2;
    a.tiles_x = cdiv(W, a.TW);
    const int rc = conv_x3_dispatch(a, (hipStream_t)stream, WS_CENTER);
    if (rc < 0) return rc;
    if (rc == 0) {
      if (ch_sum) {
        SX_REQUIRE(ch_sum_n > 0 && ch_sum_n <= Cin && ch_sum_n % 8 == 0 && dx_ld % 8 == 0, "sx_conv3_shortcut_dgrad: bad channel-sum request");
        SX_TRY(channel_sum_slots(dtype, B, H, W, H, W, ch_sum_n, ch_sum_n, dx, dx_ld, ch_sum, ch_slots, stream));
      }
      return SX_OK;
    }
  }
  const size_t esz = dtype == SX_BF16 ? 2 : 4;
  SX_TRY(sx_conv_forward(dtype, SX_CONV3, B, H, W, C, Cin, dyr, dyr_ld, H, W, packed3_t, nullptr, dx, dx_ld, H, W, 0, nullptr, 0, stream));
  SX_TRY(sx_conv_forward(dtype, SX_CONV1, B, H, W, C, Cin, (const char*)dyr + (size_t)C * esz, dyr_ld, H, W, packed1_t, nullptr, dx, dx_ld,
                         H, W, 1, nullptr, 0, stream));
  if (ch_sum) {
    SX_REQUIRE(ch_sum_n > 0 && ch_sum_n <= Cin && ch_sum_n % 8 == 0 && dx_ld % 8 == 0, "sx_conv3_shortcut_dgrad: bad channel-sum request");
    SX_TRY(channel_sum_slots(dtype, B, H, W, H, W, ch_sum_n, ch_sum_n, dx, dx_ld, ch_sum, ch_slots, stream));
  }
  return SX_OK;
}
