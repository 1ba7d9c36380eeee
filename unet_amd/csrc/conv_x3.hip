// fp32 tier of sx_conv_forward on the bf16 matrix cores: every fp32 operand is split into two bf16 planes
//     v = hi + lo + r,   hi = bf16(v),  lo = bf16(v - hi),  |r| <= 2^-18 |v|
// and a product is taken as hi*hi + hi*lo + lo*hi with fp32 accumulation (three v_mfma_f32_16x16x32_bf16 instead of
// eight v_mfma_f32_16x16x4_f32 per 16x16x32 block: 48 MFMA cycles against 256).  The dropped lo*lo term and the
// two residuals bound the error of a product by 3 * 2^-18 ~ 1.1e-5 relative; measured on whole convolutions
// 4.4e-6 relative L2 against fp64 (exact fp32 MFMA: 2.1e-7; TF32, what cuDNN runs the reference's fp32 convolutions
// in by default, carries 2^-11).  Tensors stay fp32 in HBM: the split is made once per staged element on its way
// into LDS; the weights are split once per step by the pack kernels (conv_common.h: wpack_off_x3).
// SX_FP32_EXACT=1 / sx_set_fp32_exact(1) selects the exact fp32 MFMA kernels (conv_igemm.hip) instead.
//
// Same geometry, GEMM orientation and epilogue as conv_igemm_kernel (3x3 / 1x1 / transposed forward / transposed
// data gradient); K is walked in chunks of 32 channels, both images sit in LDS as [row][hi 64 B | lo 64 B | pad],
// row stride 160 bytes (conflict-free under ds_read_b128's lane grouping: stride/16 = 10 = 2 * odd).
#include "conv_common.h"

namespace sx {

static constexpr int ROW3 = 160;     // LDS bytes per staged row: 32 bf16 hi, 32 bf16 lo, 32 bytes of padding
static constexpr int LOFF = 64;      // the lo plane inside a row

// 4 floats -> 4 bf16 hi + 4 bf16 lo (round to nearest even; the difference v - hi is exact in fp32)
__device__ __forceinline__ void split4(const f32x4& v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bf16 h = (bf16)v[e];
    hi[e] = h;
    lo[e] = (bf16)(v[e] - (float)h);
  }
}

// NPX: pixel pieces (16 bytes = 4 fp32 channels) a thread stages per chunk
template <int MI, int NI, int WM, int WN, int NPX>
__global__ __launch_bounds__(256) void conv_x3_kernel(ConvArgs a) {
  constexpr int BMC = WM * MI * 16;
  constexpr int NPW = (9 * BMC * 8 + 255) / 256;        // weight pieces per thread (nine taps at most)
  static_assert(WM * WN == 4, "4 waves");
  static_assert(WN * NI * 16 == 128, "128-pixel tile");
  extern __shared__ __align__(16) char smem[];
  char* sm_pix = smem;
  char* sm_w = smem + a.LH * a.LW * ROW3;

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
    pix_off[i] = ((py * a.in_scale) * a.LW + px * a.in_scale) * ROW3 + g * 16;
  }
  int w_off[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) w_off[j] = ((wm * MI + j) * 16 + n) * ROW3 + g * 16;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int j = 0; j < MI; ++j)
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* __restrict__ xb = reinterpret_cast<const float*>(a.x) + (int64_t)b * a.Hx * a.Wx * a.x_ld;   // this sample
  const char* __restrict__ wg = reinterpret_cast<const char*>(a.w);
  const int npix_pieces = a.LH * a.LW * 8;
  const int nw_pieces = a.ntaps * BMC * 8;
  const int iy0 = ty0 * a.in_scale - a.pad, ix0 = tx0 * a.in_scale - a.pad;
  const int kch = a.KPad >> 5;

  // the next chunk's pieces are fetched into registers under this chunk's MFMAs; coordinates are decoded once
  f32x4 pre_x[NPX];
  u32x4 pre_w[NPW];
  int px_off[NPX];          // element offset inside the sample without the chunk term; < 0 = reads as zero
  int64_t pw_off[NPW];      // byte offset without the chunk term; < 0 = reads as zero
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    const int p = tid + i * 256, r = p >> 3, q = p & 7;
    const int ly = r / a.LW, lx = r - ly * a.LW;
    const int iy = iy0 + ly, ix = ix0 + lx;
    const bool ok = p < npix_pieces && iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx;
    px_off[i] = ok ? (iy * a.Wx + ix) * a.x_ld + q * 4 : -1;
  }
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int p = tid + i * 256, r = p >> 3, q = p & 7;
    const int tap = r / BMC, row = r - tap * BMC;
    const bool ok = p < nw_pieces && cb + row < a.RTot;
    pw_off[i] = ok ? (((int64_t)tap * a.RTot + cb + row) * kch << 7) + q * 16 : -1;
  }
  auto fetch = [&](int kc) {      // kc = first channel of the chunk
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (px_off[i] >= 0 && kc + ((tid + i * 256) & 7) * 4 < a.K) v = *reinterpret_cast<const f32x4*>(xb + px_off[i] + kc);
      pre_x[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (pw_off[i] >= 0) v = *reinterpret_cast<const u32x4*>(wg + pw_off[i] + ((int64_t)(kc >> 5) << 7));
      pre_w[i] = v;
    }
  };
  fetch(0);
  for (int kc = 0; kc < a.KPad; kc += 32) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int p = tid + i * 256;
      if (p < npix_pieces) {
        bf16x4 hi, lo;
        split4(pre_x[i], hi, lo);
        char* dst = sm_pix + (p >> 3) * ROW3 + (p & 7) * 8;
        *reinterpret_cast<bf16x4*>(dst) = hi;
        *reinterpret_cast<bf16x4*>(dst + LOFF) = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = tid + i * 256;
      if (p < nw_pieces) *reinterpret_cast<u32x4*>(sm_w + (p >> 3) * ROW3 + (p & 7) * 16) = pre_w[i];
    }
    __syncthreads();
    if (kc + 32 < a.KPad) fetch(kc + 32);
    for (int tap = 0; tap < a.ntaps; ++tap) {
      const int tyy = tap / a.tapw, txx = tap - tyy * a.tapw;
      const int toff = (tyy * a.LW + txx) * ROW3;
      const char* wt = sm_w + tap * BMC * ROW3;
      bf16x8 Ah[MI], Al[MI], Bh[NI], Bl[NI];
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        Ah[j] = *reinterpret_cast<const bf16x8*>(wt + w_off[j]);
        Al[j] = *reinterpret_cast<const bf16x8*>(wt + w_off[j] + LOFF);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        Bh[i] = *reinterpret_cast<const bf16x8*>(sm_pix + pix_off[i] + toff);
        Bl[i] = *reinterpret_cast<const bf16x8*>(sm_pix + pix_off[i] + toff + LOFF);
      }
      // the two small terms first, the leading term last
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al[j], Bh[i], acc[j][i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[j], Bl[i], acc[j][i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[j], Bh[i], acc[j][i], 0, 0, 0);
    }
  }

  // ------------------------------------------------------------------ epilogue (as conv_igemm_kernel<float>)
  float* __restrict__ yg = reinterpret_cast<float*>(a.y);
  const bool do_stats = a.gn_stats != nullptr;
  float* sm_stat = reinterpret_cast<float*>(smem);   // [WN][BMC][2], reused after the K loop
  if (do_stats) __syncthreads();

#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int cl = (wm * MI + j) * 16 + 4 * g;
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
        float* dst = yg + ((int64_t)(b * a.Hy + oy) * a.Wy + ox) * a.y_ld + co;
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
    if (tid < BMC) {
      const int c = cb + tid;
      const int cend = min(a.R, cb + BMC);
      if (c < cend && (tid == 0 || (c % a.cpg) == 0)) {
        const int grp = c / a.cpg;
        float t1 = 0.f, t2 = 0.f;
        for (int cc = c; cc < cend && cc / a.cpg == grp; ++cc)
#pragma unroll
          for (int w = 0; w < WN; ++w) {
            t1 += sm_stat[(w * BMC + (cc - cb)) * 2 + 0];
            t2 += sm_stat[(w * BMC + (cc - cb)) * 2 + 1];
          }
        const int G = a.R / a.cpg;
        atomicAdd(a.gn_stats + ((int64_t)b * G + grp) * 2 + 0, t1);
        atomicAdd(a.gn_stats + ((int64_t)b * G + grp) * 2 + 1, t2);
      }
    }
  }
}

template <int MI, int NI, int WM, int WN, int NPX>
static int launch_x3(const ConvArgs& a, hipStream_t st) {
  constexpr int BMC = WM * MI * 16;
  size_t smem = (size_t)(a.LH * a.LW + a.ntaps * BMC) * ROW3;
  const size_t stat_bytes = (size_t)WN * BMC * 2 * sizeof(float);
  if (smem < stat_bytes) smem = stat_bytes;
  if (smem > 160 * 1024) { set_error("conv tile needs %zu B of LDS", smem); return SX_ERR_UNSUPPORTED; }
  if (a.LH * a.LW * 8 > NPX * 256 || a.ntaps > 9) {
    set_error("conv: fp32 tile of %d pixels x %d taps exceeds the prefetch slots", a.LH * a.LW, a.ntaps);
    return SX_ERR_UNSUPPORTED;
  }
  if ((int64_t)a.Hx * a.Wx * a.x_ld >= ((int64_t)1 << 31)) { set_error("conv: one sample exceeds 2^31 elements"); return SX_ERR_UNSUPPORTED; }
  auto kern = conv_x3_kernel<MI, NI, WM, WN, NPX>;
  static std::atomic<unsigned long long> done{0};
  SX_TRY(ensure_max_lds(reinterpret_cast<const void*>(kern), done, "hipFuncSetAttribute(conv_x3)"));
  dim3 grid(a.tiles_x * a.tiles_y * a.B, cdiv(a.RTot, BMC));
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a);
  SX_LAUNCH_CHECK("conv_x3");
  return SX_OK;
}

template <int NPX>
static int x3_tiles(const ConvArgs& a, hipStream_t st) {
  if (a.RTot <= 16) return launch_x3<1, 2, 1, 4, NPX>(a, st);
  if (a.RTot <= 32) return launch_x3<2, 2, 1, 4, NPX>(a, st);
  return launch_x3<2, 4, 2, 2, NPX>(a, st);
}

int conv_x3_dispatch(const ConvArgs& a, hipStream_t st) {
  if (a.KPad % 32) { set_error("conv_x3: packed K %d is not a multiple of 32", a.KPad); return SX_ERR_INVALID; }
  // the transposed convolution's data gradient gathers from a 16 x 32 pixel image (16 pieces per thread), everything
  // else from at most 34 x 6 pixels (7)
  if (a.in_scale == 2) return x3_tiles<16>(a, st);
  return x3_tiles<7>(a, st);
}

}  // namespace sx
