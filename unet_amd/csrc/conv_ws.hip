// Weights-stationary persistent implicit-GEMM convolution, bf16, gfx950 (the fast tier of sx_conv_forward
// for 3x3 / 1x1 convolutions on grids at least 16 wide).
//
// This U-Net is "few channels, many pixels" (32..256 channels, up to 524k pixels per layer at batch 64), so the
// weight slice of a channel tile fits in REGISTERS: every wave loads its MFMA A-fragments (packed weights,
// [KCH chunks][taps][MI] x 8 bf16 per lane, up to 288 VGPRs of the 512 a wave may own at one wave per SIMD) once,
// and the block then walks pixel tiles persistently.  Consequences:
//   * no weight traffic through LDS at all (it was 3/4 of the LDS bytes of the generic kernel);
//   * the only LDS image is the input halo tile (rows of 16 pixels + halo, 64 bytes per pixel and K chunk), filled by
//     LDS-DMA (global_load_lds_dwordx4: no staging VGPRs) into a ring of 4 buffers, three stages in flight, one raw
//     s_barrier + counted s_waitcnt vmcnt per stage -> HBM latency is covered by ~3 stages of MFMAs, not 1;
//   * fragments of one halo row serve all three vertical taps: (NI+2)*3 LDS reads feed 9*MI*NI MFMAs per chunk
//     (0.25 reads per MFMA at MI=2, NI=4; the generic kernel needs 0.75).
// Output orientation / epilogue (4 consecutive channels per lane, fused GroupNorm partial sums) as in conv_igemm.hip.
#include "conv_common.h"
#include <stdlib.h>

namespace sx {

// 64 zero bytes: the source of every halo / padding piece (LDS-DMA cannot write immediates)
__device__ __attribute__((aligned(64))) unsigned g_zero_page[16];

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int KCH, int TAPS, int MI, int NI, int WM, int WN, int NBUF>
__global__ __launch_bounds__(256, 1) void conv_ws_kernel(ConvArgs a) {
  constexpr int TAPW = TAPS == 9 ? 3 : 1;
  constexpr int PAD = TAPS == 9 ? 1 : 0;
  constexpr int BMC = WM * MI * 16;
  constexpr int TH = WN * NI, TW = 16;
  constexpr int LW = TW + TAPW - 1, LH = TH + TAPW - 1;
  constexpr int NPIX = LH * LW, PIECES = NPIX * 4, NP = (PIECES + 255) / 256;
  constexpr int BUF = NP * 256 * 16;            // lane-linear image: piece p (pixel p>>2, part p&3) at byte 16*p
  constexpr int DIST = NBUF - 1;                // stages in flight
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __align__(16) char smem[];
  float* sm_stat = reinterpret_cast<float*>(smem + NBUF * BUF);   // [WN][BMC][2]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int n = lane & 15, g = lane >> 4;
  const int cb = blockIdx.y * BMC;
  const int ntiles = a.B * a.tiles_y * a.tiles_x;
  int t = blockIdx.x;
  if (t >= ntiles) return;
  // stages of this block: s = 0 .. nst-1, stage s = (tile t0 + (s / KCH) * gridDim.x, chunk s % KCH)
  const int my_tiles = (ntiles - t + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nst = my_tiles * KCH;

  const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(a.x);
  const bf16* __restrict__ wg = reinterpret_cast<const bf16*>(a.w);
  bf16* __restrict__ yg = reinterpret_cast<bf16*>(a.y);

  // ---- stationary weights: A[kc][tap][j] = W[tap][cb + (wm*MI+j)*16 + n][kc*32 + 8g .. +8]
  bf16x8 A[KCH][TAPS][MI];
#pragma unroll
  for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int row = cb + (wm * MI + j) * 16 + n;
        bf16x8 v = {};
        if (row < a.RTot) v = *reinterpret_cast<const bf16x8*>(wg + ((int64_t)tap * a.RTot + row) * a.KPad + kc * 32 + g * 8);
        A[kc][tap][j] = v;
      }

  // per-thread staging slots: piece p = tid + q*256 -> halo pixel r = p>>2, 16-byte part p&3
  int st_ly[NP], st_lx[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int r = (tid + q * 256) >> 2;
    st_ly[q] = (tid + q * 256) < PIECES ? r / LW : -100000;    // padding lanes never pass the bounds test
    st_lx[q] = r - (r / LW) * LW;
  }
  const int part = tid & 3;
  const int frag_off = (wn * NI * LW + n) * 64 + g * 16;
  const bf16* zero = reinterpret_cast<const bf16*>(g_zero_page);

  // LDS-DMA: every wave issues exactly NP global_load_lds per stage (dummy stages read the zero page), so the
  // vmcnt bookkeeping is uniform: before computing stage s at most (DIST-1)*NP younger loads may be pending.
  auto issue_stage = [&](int s) {
    char* buf = smem + (s % NBUF) * BUF + wid * 1024;
    const bool live = s < nst;
    const int kc = s % KCH;
    int tt = blockIdx.x + (s / KCH) * (int)gridDim.x;
    const int tx_i = tt % a.tiles_x; tt /= a.tiles_x;
    const int ty_i = tt % a.tiles_y; tt /= a.tiles_y;
    const int iy0 = ty_i * TH - PAD, ix0 = tx_i * TW - PAD;
    const int c = kc * 32 + part * 8;
    const bf16* base = xg + (int64_t)tt * a.Hx * a.Wx * a.x_ld + c;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int iy = iy0 + st_ly[q], ix = ix0 + st_lx[q];
      const bf16* src = zero;
      if (live && iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx && c < a.K) src = base + ((int64_t)iy * a.Wx + ix) * a.x_ld;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + q * 4096), 16, 0, 0);
    }
  };

#pragma unroll
  for (int s = 0; s < DIST; ++s) issue_stage(s);
  const bool do_stats = a.gn_stats != nullptr;
  int s = 0;

  while (true) {
    f32x4 acc[MI][NI];
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int kc = 0; kc < KCH; ++kc, ++s) {
      wait_vmcnt<(DIST - 1) * NP>();                 // this wave's pieces of stage s have landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // and its LDS reads of stage s-1 have returned
      __builtin_amdgcn_s_barrier();                  // ... for every wave: buffer (s-1) % NBUF is free
      issue_stage(s + DIST);
      const char* buf = smem + (s % NBUF) * BUF + frag_off;
#pragma unroll
      for (int rr = 0; rr < NI + TAPW - 1; ++rr) {
        bf16x8 Bf[TAPW];
#pragma unroll
        for (int tx = 0; tx < TAPW; ++tx) Bf[tx] = *reinterpret_cast<const bf16x8*>(buf + (rr * LW + tx) * 64);
#pragma unroll
        for (int ty = 0; ty < TAPW; ++ty) {
          const int i = rr - ty;
          if (i >= 0 && i < NI) {
#pragma unroll
            for (int tx = 0; tx < TAPW; ++tx)
#pragma unroll
              for (int j = 0; j < MI; ++j)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kc][ty * TAPW + tx][j], Bf[tx], acc[j][i], 0, 0, 0);
          }
        }
      }
    }

    // ---- epilogue of tile t
    int tt = t;
    const int tx_i = tt % a.tiles_x; tt /= a.tiles_x;
    const int ty_i = tt % a.tiles_y; tt /= a.tiles_y;
    const int b = tt;
    const int ox = tx_i * TW + n;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int cl = (wm * MI + j) * 16 + 4 * g;
      const int c0 = cb + cl;
      const bool cvalid = c0 < a.R;
      f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int oy = ty_i * TH + wn * NI + i;
        f32x4 v = acc[j][i];
        if (cvalid && oy < a.H && ox < a.W) {
          bf16* dst = yg + ((int64_t)(b * a.Hy + oy) * a.Wy + ox) * a.y_ld + c0;
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
      // raw barrier: __syncthreads() would drain the LDS-DMA prefetches (vmcnt(0)); only LDS writes matter here
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
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
    t += gridDim.x;
    if (t >= ntiles) break;
  }
  wait_vmcnt<0>();     // drain the dummy prefetches before the wave (and its LDS allocation) retires
}

template <int KCH, int TAPS, int MI, int NI, int WM, int WN>
static int launch_ws(ConvArgs a, hipStream_t st) {
  constexpr int TAPW = TAPS == 9 ? 3 : 1;
  constexpr int BMC = WM * MI * 16, TH = WN * NI, TW = 16;
  constexpr int NPIX = (TH + TAPW - 1) * (TW + TAPW - 1);
  constexpr int NP = (NPIX * 4 + 255) / 256;
  constexpr int NBUF = 4;                                       // three stages in flight per block
  const size_t smem = (size_t)NBUF * NP * 4096 + (size_t)WN * BMC * 2 * sizeof(float);
  auto kern = conv_ws_kernel<KCH, TAPS, MI, NI, WM, WN, NBUF>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_ws)");
    configured = true;
  }
  a.TH = TH; a.TW = TW;
  a.tiles_x = cdiv(a.W, TW);
  a.tiles_y = cdiv(a.H, TH);
  const int ntiles = a.B * a.tiles_x * a.tiles_y;
  const int gy = cdiv(a.RTot, BMC);
  // persistent grid: about one block per CU (registers allow one wave per SIMD); light variants get two
  int per_cu = (KCH * TAPS * MI <= 36) ? 2 : 1;
  if (const char* e = getenv("SX_WS_PER_CU")) per_cu = atoi(e);     // tuning knob
  const int gx = std::max(1, std::min(ntiles, (256 * per_cu) / gy));
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), smem, st, a);
  SX_LAUNCH_CHECK("conv_ws");
  return SX_OK;
}

template <int KCH, int TAPS>
static int ws_pick_tile(const ConvArgs& a, hipStream_t st) {
  if (a.RTot <= 32) {
    if constexpr (KCH <= 4) return launch_ws<KCH, TAPS, 2, 4, 1, 4>(a, st);   // 32 ch x 256 px (16 rows)
    else return 1;
  } else if (a.RTot <= 64) {
    if constexpr (KCH <= 4) return launch_ws<KCH, TAPS, 2, 4, 2, 2>(a, st);   // 64 ch x 128 px (8 rows)
    else return launch_ws<KCH, TAPS, 1, 8, 4, 1>(a, st);                      // 64 ch x 128 px
  } else {
    if constexpr (KCH <= 4) return launch_ws<KCH, TAPS, 2, 8, 4, 1>(a, st);   // 128 ch x 128 px
    else return launch_ws<KCH, TAPS, 1, 8, 4, 1>(a, st);                      // 64 ch x 128 px, grid.y = RTot/64
  }
}

template <int TAPS>
static int ws_pick_k(const ConvArgs& a, hipStream_t st) {
  switch (a.KPad / 32) {
    case 1: return ws_pick_tile<1, TAPS>(a, st);
    case 2: return ws_pick_tile<2, TAPS>(a, st);
    case 4: return ws_pick_tile<4, TAPS>(a, st);
    case 8: return ws_pick_tile<8, TAPS>(a, st);
  }
  return 1;
}

// 0 = launched, 1 = not covered (caller falls back to the generic kernel), < 0 = error
int conv_ws_try(const ConvArgs& a, int mode, hipStream_t st) {
  if (mode != SX_CONV3 && mode != SX_CONV1) return 1;
  if (a.W < 16 || a.bias || a.shuffle || a.R % 4) return 1;
  if (a.KPad % 32) return 1;
  return mode == SX_CONV3 ? ws_pick_k<9>(a, st) : ws_pick_k<1>(a, st);
}

}  // namespace sx
