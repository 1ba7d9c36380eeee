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

// LDS bytes per staged row: PL planes of 32 bf16 (64 bytes each) + 32 bytes of padding.  160 (two planes) and 224 (three):
// stride / 16 = 2 * odd, conflict-free under ds_read_b128's lane grouping (208 and 192 take 8 LDS cycles per read, not 4).
// The WEIGHT image of the three-plane form is stored unpadded (192 bytes per row, 8 cycles per fragment read): nine taps x
// 64 rows at 224 bytes do not fit beside the pixel image (169 KB), and a 32-channel tile stages every pixel twice as often
// (measured: cfg 5 fp32 step 31.1 ms with 32-channel tiles).
template <int PL> struct X3Row { static constexpr int BYTES = PL * 64 + 32; static constexpr int WBYTES = PL == 3 ? 192 : BYTES; };

// THREE planes ("x6", round 3): v = hi + mid + lo + r, |r| <= 2^-27 |v|, and a product keeps the six terms down to 2^-18
// (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the dropped mid*lo, lo*mid, lo*lo are <= 2^-27): fp32-grade products at six
// bf16 MFMAs per block (96 matrix-core cycles against 256 of the exact kernels).  Used where two planes were measured outside
// the 1e-4 contract: networks whose widest layer reaches 512 channels (unet_plan.hip).

// 4 floats -> PL bf16 planes (round to nearest even; every difference is exact in fp32)
template <int PL>
__device__ __forceinline__ void split4(const f32x4& v, bf16x4 (&pl)[PL]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float r = v[e];
#pragma unroll
    for (int q = 0; q < PL; ++q) {
      const bf16 h = (bf16)r;
      pl[q][e] = h;
      r -= (float)h;
    }
  }
}

// acc += A * B over the kept cross terms, small terms first; A / B indexed [plane]
template <int PL>
__device__ __forceinline__ f32x4 x3_mma(const bf16x8 (&A)[PL], const bf16x8 (&B)[PL], f32x4 acc) {
  if constexpr (PL == 3) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2], B[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[1], acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[0], acc, 0, 0, 0);
  return acc;
}

// One block = WM x WN waves, a tile of WM*MI*16 output channels x WN*NI*16 pixels.  Blocks are persistent: a block walks
// tiles blockIdx.x, blockIdx.x + gridDim.x, ... and the (tile, K chunk) pairs form ONE sequence of stages, so the register
// prefetch of the next stage also runs across tile boundaries and under the epilogue; with a single K chunk the weight
// image stays in LDS for every tile of the block.  Fragments of tap t+1 are read while the MFMAs of tap t run.
// NPX: pixel pieces (16 bytes = 4 fp32 channels) a thread stages per chunk.
template <int TAPS, int MI, int NI, int WM, int WN, int NPX, bool CENTER = false, int PL = 2>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && WM * MI <= 2 && TAPS != 4 && PL == 2) ? 2 : 1) void conv_x3_kernel(ConvArgs a) {     // <= 32-channel tiles: two blocks per CU
  constexpr int NT = WM * WN * 64;
  constexpr int BMC = WM * MI * 16;
  constexpr int TAPW = TAPS == 9 ? 3 : (TAPS == 4 ? 2 : 1);
  constexpr int ROW3 = X3Row<PL>::BYTES, WROW = X3Row<PL>::WBYTES;
  constexpr int WPR = 4 * PL;                                // 16-byte pieces of one packed (row, chunk): PL planes of 64 bytes
  constexpr int NPW = (TAPS * BMC * WPR + NT - 1) / NT;      // weight pieces per thread
  extern __shared__ __align__(16) char smem[];
  char* sm_pix = smem;
  char* sm_w = smem + a.LH * a.LW * ROW3;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int n = lane & 15, g = lane >> 4;
  const int cb = blockIdx.y * BMC;
  const int ntiles = a.B * a.tiles_y * a.tiles_x;

  int pix_off[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = (wn * NI + i) * 16 + n;
    const int py = m >> a.tw_log2, px = m & (a.TW - 1);
    pix_off[i] = ((py * a.in_scale) * a.LW + px * a.in_scale) * ROW3 + g * 16;
  }
  int w_off[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) w_off[j] = ((wm * MI + j) * 16 + n) * WROW + g * 16;

  const float* __restrict__ xg = reinterpret_cast<const float*>(a.x);
  const char* __restrict__ wg = reinterpret_cast<const char*>(a.w);
  float* __restrict__ yg = reinterpret_cast<float*>(a.y);
  const int npix_pieces = a.LH * a.LW * 8;
  constexpr int nw_pieces = TAPS * BMC * WPR;
  const int kch = a.KPad >> 5;
  // CENTER (the joint data gradient of conv1 + 1x1 shortcut, sx_conv3_shortcut_dgrad): the input holds 2K channels
  // [dY1 | dR]; chunks of the second half only feed the centre tap, with the 1x1 weights w2 staged in the image of tap 0
  const int kend = CENTER ? 2 * a.KPad : a.KPad;
  const char* __restrict__ wg2 = reinterpret_cast<const char*>(a.w2);
  const bool do_stats = a.gn_stats != nullptr;
  float* sm_stat = reinterpret_cast<float*>(smem);   // [WN][BMC][2], overlays the pixel image between two tiles

  // staging slots: halo coordinates of this thread's pixel pieces and byte offsets of its weight pieces, decoded once
  f32x4 pre_x[NPX];
  u32x4 pre_w[NPW];
  int sl_y[NPX], sl_x[NPX];
  int pw_off[NPW];          // byte offset without the chunk term (a packed weight is a few MB); < 0 = reads as zero
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    const int p = tid + i * NT, r = p >> 3;
    const int ly = r / a.LW;
    sl_y[i] = p < npix_pieces ? ly : (1 << 28);       // pieces past the image fail every bounds test
    sl_x[i] = r - ly * a.LW;
  }
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int p = tid + i * NT, r = p / WPR, q = p - r * WPR;
    const int tap = r / BMC, row = r - tap * BMC;
    const bool ok = p < nw_pieces && cb + row < a.RTot;
    pw_off[i] = ok ? ((tap * a.RTot + cb + row) * kch) * (64 * PL) + q * 16 : -1;
  }
  struct Tile { int b, ty0, tx0; };
  auto tile_at = [&](int t) {
    Tile T;
    T.tx0 = (t % a.tiles_x) * a.TW; t /= a.tiles_x;
    T.ty0 = (t % a.tiles_y) * a.TH; T.b = t / a.tiles_y;
    return T;
  };
  auto fetch = [&](const Tile& T, int kc, bool weights) {      // kc = first channel of the chunk
    const float* xb = xg + (int64_t)T.b * a.Hx * a.Wx * a.x_ld;
    const int iy0 = T.ty0 * a.in_scale - a.pad, ix0 = T.tx0 * a.in_scale - a.pad;
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int iy = iy0 + sl_y[i], ix = ix0 + sl_x[i];
      const int c = kc + ((tid + i * NT) & 7) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx && c < (CENTER ? 2 * a.K : a.K)) v = *reinterpret_cast<const f32x4*>(xb + (iy * a.Wx + ix) * a.x_ld + c);
      pre_x[i] = v;
    }
    if (weights) {
      if (CENTER && kc >= a.KPad) {
        // 1x1 weights [row][chunk]: the pieces of tap 0 (p < BMC * 8) take the same rows of w2
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          const int p = tid + i * NT;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (p < BMC * WPR && pw_off[i] >= 0) v = *reinterpret_cast<const u32x4*>(wg2 + (pw_off[i] + ((kc - a.KPad) >> 5) * (64 * PL)));
          pre_w[i] = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (pw_off[i] >= 0) v = *reinterpret_cast<const u32x4*>(wg + (pw_off[i] + (kc >> 5) * (64 * PL)));
          pre_w[i] = v;
        }
      }
    }
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  Tile cur = tile_at(t);
  fetch(cur, 0, true);
  bool w_in_regs = true;
  while (t < ntiles) {
    f32x4 acc[MI][NI];
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tn = t + (int)gridDim.x;
    const Tile nxt = tile_at(tn < ntiles ? tn : t);

    for (int kc = 0; kc < kend; kc += 32) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const int p = tid + i * NT;
        if (p < npix_pieces) {
          bf16x4 pl[PL];
          split4<PL>(pre_x[i], pl);
          char* dst = sm_pix + (p >> 3) * ROW3 + (p & 7) * 8;
#pragma unroll
          for (int q = 0; q < PL; ++q) *reinterpret_cast<bf16x4*>(dst + 64 * q) = pl[q];
        }
      }
      const bool second = CENTER && kc >= a.KPad;
      if (w_in_regs) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          const int p = tid + i * NT;
          if (p < (second ? BMC * WPR : nw_pieces)) *reinterpret_cast<u32x4*>(sm_w + (p / WPR) * WROW + (p % WPR) * 16) = pre_w[i];
        }
      }
      __syncthreads();
      // next stage: the next chunk of this tile, or the first chunk of the block's next tile
      if (kc + 32 < kend) fetch(cur, kc + 32, true);
      else if (tn < ntiles) { w_in_regs = CENTER || kch > 1; fetch(nxt, 0, w_in_regs); }

      if constexpr (CENTER) {
        if (second) {            // centre tap only: pixel (py + 1, px + 1) of the halo image, weights in the image of tap 0
          const int toff = (a.LW + 1) * ROW3;
          bf16x8 Ac[MI][PL];
#pragma unroll
          for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int q = 0; q < PL; ++q) Ac[j][q] = *reinterpret_cast<const bf16x8*>(sm_w + w_off[j] + 64 * q);
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            bf16x8 Bc[PL];
#pragma unroll
            for (int q = 0; q < PL; ++q) Bc[q] = *reinterpret_cast<const bf16x8*>(sm_pix + pix_off[i] + toff + 64 * q);
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[j][i] = x3_mma<PL>(Ac[j], Bc, acc[j][i]);
          }
          continue;
        }
      }

      // one wave per SIMD: the fragments of tap t+1 are read while the MFMAs of tap t run (two register sets); with two
      // waves per SIMD the other wave covers the LDS latency and one set keeps the kernel inside 256 VGPRs
      constexpr int NS = (NT == 256 && !(PL == 3 && CENTER && MI * NI >= 8)) ? 2 : 1;      // (that one spills 256 bytes per lane with two sets)
      bf16x8 Af[NS][MI][PL], Bf[NS][NI][PL];
      auto read_tap = [&](int tap, int slot) {
        const int tyy = tap / TAPW, txx = tap - tyy * TAPW;
        const int toff = (tyy * a.LW + txx) * ROW3;
        const char* wt = sm_w + tap * BMC * WROW;
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
          for (int q = 0; q < PL; ++q) Af[slot][j][q] = *reinterpret_cast<const bf16x8*>(wt + w_off[j] + 64 * q);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int q = 0; q < PL; ++q) Bf[slot][i][q] = *reinterpret_cast<const bf16x8*>(sm_pix + pix_off[i] + toff + 64 * q);
      };
      if constexpr (NS == 2) read_tap(0, 0);
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int sl = NS == 2 ? (tap & 1) : 0;
        if constexpr (NS == 2) {
          if (tap + 1 < TAPS) read_tap(tap + 1, sl ^ 1);
          __builtin_amdgcn_sched_barrier(0);         // keep the reads of the next tap above this tap's MFMAs
        } else {
          read_tap(tap, 0);
        }
        // the small terms first, the leading term last; term by term over the tile so that consecutive MFMAs hit different accumulators
        if constexpr (PL == 3) {
#pragma unroll
          for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[sl][j][2], Bf[sl][i][0], acc[j][i], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[sl][j][0], Bf[sl][i][2], acc[j][i], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[sl][j][1], Bf[sl][i][1], acc[j][i], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
          for (int i = 0; i < NI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[sl][j][1], Bf[sl][i][0], acc[j][i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
          for (int i = 0; i < NI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[sl][j][0], Bf[sl][i][1], acc[j][i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
          for (int i = 0; i < NI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[sl][j][0], Bf[sl][i][0], acc[j][i], 0, 0, 0);
        if constexpr (NS == 2) __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---------------------------------------------------------------- epilogue of tile `cur` (as conv_igemm_kernel<float>)
    if (do_stats) __syncthreads();        // every wave is done with the pixel image: sm_stat may overlay it
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
        int oy = cur.ty0 + py, ox = cur.tx0 + px;
        const bool pvalid = (oy < a.H) && (ox < a.W);
        if (a.shuffle) { oy = 2 * oy + (tap >> 1); ox = 2 * ox + (tap & 1); }
        f32x4 v = acc[j][i] + bias;
        if (pvalid && cvalid) {
          float* dst = yg + ((int64_t)(cur.b * a.Hy + oy) * a.Wy + ox) * a.y_ld + co;
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
      gn_stats_flush<BMC, WN>(sm_stat, tid, cb, a.R, a.cpg, a.gn_stats + (int64_t)cur.b * (a.R / a.cpg) * 2);
      // (the next stage starts with a barrier: the image is not rewritten before these reads are done)
    }
    t = tn;
    cur = nxt;
  }
}

template <int TAPS, int MI, int NI, int WM, int WN, int NPX, bool CENTER = false, int PL = 2>
static int launch_x3(ConvArgs a, hipStream_t st) {
  constexpr int ROW3 = X3Row<PL>::BYTES;
  constexpr int NT = WM * WN * 64;
  constexpr int BMC = WM * MI * 16, TP = WN * NI * 16;
  const int taph = TAPS == 9 ? 3 : (TAPS == 4 ? 2 : 1);
  a.TH = TP / a.TW;
  a.LH = (a.TH - 1) * a.in_scale + taph;
  a.tiles_y = cdiv(a.H, a.TH);
  size_t smem = (size_t)a.LH * a.LW * ROW3 + (size_t)TAPS * BMC * X3Row<PL>::WBYTES;
  const size_t stat_bytes = (size_t)WN * BMC * 2 * sizeof(float);
  if (smem < stat_bytes) smem = stat_bytes;
  if (smem > 160 * 1024) { set_error("conv tile needs %zu B of LDS", smem); return SX_ERR_UNSUPPORTED; }
  if (a.LH * a.LW * 8 > NPX * NT) {
    set_error("conv: fp32 tile of %d pixels exceeds the prefetch slots", a.LH * a.LW);
    return SX_ERR_UNSUPPORTED;
  }
  if ((int64_t)a.Hx * a.Wx * a.x_ld >= ((int64_t)1 << 31)) { set_error("conv: one sample exceeds 2^31 elements"); return SX_ERR_UNSUPPORTED; }
  auto kern = conv_x3_kernel<TAPS, MI, NI, WM, WN, NPX, CENTER, PL>;
  static std::atomic<unsigned long long> done{0};
  SX_TRY(ensure_max_lds(reinterpret_cast<const void*>(kern), done, "hipFuncSetAttribute(conv_x3)"));
  // persistent grid: as many blocks as the LDS images allow on 256 CUs (2 in flight per CU at least when they are small)
  const int ntiles = a.tiles_x * a.tiles_y * a.B;
  const int gy = cdiv(a.RTot, BMC);
  // resident blocks per CU as the runtime computes it from registers and LDS (cached per 4 KB bucket of LDS: the query is a host call)
  static std::atomic<int> occ[41];      // (per instantiation)
  int per_cu = occ[smem >> 12].load(std::memory_order_relaxed);
  if (per_cu == 0) {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), NT, smem);
    if (e != hipSuccess) return hip_fail(e, "hipOccupancyMaxActiveBlocksPerMultiprocessor(conv_x3)");
    per_cu = std::max(1, std::min(4, nb));
    occ[smem >> 12].store(per_cu, std::memory_order_relaxed);
  }
  static const int oversub = env_int("SX_X3_WAVES", 1);         // tuning knob: tiles per resident block slot
  int gx = std::max(1, std::min(ntiles, (256 * per_cu * oversub) / gy));
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(NT), smem, st, a);
  SX_LAUNCH_CHECK("conv_x3");
  return SX_OK;
}

template <int TAPS, int NPX, bool CENTER = false>
static int x3_tiles(const ConvArgs& a, hipStream_t st) {
  if (fp32_planes() == 3) {
    // three planes: 64 channels x 128 pixels on four waves (the 256-pixel form's images exceed the LDS)
    if (a.RTot <= 16) return launch_x3<TAPS, 1, 2, 1, 4, NPX, CENTER, 3>(a, st);
    if (a.RTot <= 32 || TAPS == 4) return launch_x3<TAPS, 2, 2, 1, 4, NPX, CENTER, 3>(a, st);     // (transposed data gradient: a 16 x 32 pixel image)
    return launch_x3<TAPS, 2, 4, 2, 2, NPX, CENTER, 3>(a, st);
  }
  if (a.RTot <= 16) return launch_x3<TAPS, 1, 2, 1, 4, NPX, CENTER>(a, st);
  if (a.RTot <= 32) return launch_x3<TAPS, 2, 2, 1, 4, NPX, CENTER>(a, st);
  // 64 channels x 256 pixels on 8 waves where a sample has that many pixels in tiles of this width (half the weight
  // traffic per pixel: at one block per CU the 128-pixel form was bound by re-fetching 74 KB of weights per chunk)
  if constexpr (TAPS != 4) {
    constexpr int NPX8 = TAPS == 9 ? 6 : 4;
    const int th = 256 / a.TW;
    if (a.H * a.TW >= 256 && (th + (TAPS == 9 ? 2 : 0)) * a.LW * 8 <= NPX8 * 512) return launch_x3<TAPS, 2, 4, 2, 4, NPX8, CENTER>(a, st);
  }
  return launch_x3<TAPS, 2, 4, 2, 2, NPX, CENTER>(a, st);
}

int conv_x3_dispatch(const ConvArgs& a, hipStream_t st, int var) {
  if (a.KPad % 32) { set_error("conv_x3: packed K %d is not a multiple of 32", a.KPad); return SX_ERR_INVALID; }
  if (var == WS_CENTER) {
    if (a.ntaps != 9 || !a.w2 || a.K != a.KPad || a.x_ld < 2 * a.K || a.accumulate || a.gn_stats || a.bias || a.shuffle) return 1;   // 1 = not covered
    return x3_tiles<9, 7, true>(a, st);
  }
  if (var != WS_PLAIN) return 1;
  // pieces per thread at 256 threads: the transposed convolution's data gradient gathers from a 16 x 32 pixel image (16),
  // everything else from at most 34 x 6 pixels (7)
  if (a.ntaps == 9) return x3_tiles<9, 7>(a, st);
  if (a.ntaps == 4) return x3_tiles<4, 16>(a, st);
  return x3_tiles<1, 7>(a, st);
}

}  // namespace sx
