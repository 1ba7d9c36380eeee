// Weight gradients of the 3x3 / 1x1 / ConvTranspose2d(2,2) layers as a split-K MFMA GEMM whose
// reduction dimension is the pixel index:
//     dW[tap][a][b] = sum_{n,py,px} P[n,py,px,a] * Q[n, qy(py,tap), qx(px,tap), b]
//   conv3x3 : P = dY (a = co), Q = X  (b = ci), q = p + tap - 1 (zero outside)   -> W[co][ci][ty][tx]
//   conv1x1 : same with one tap
//   convT   : P = X  (a = ci), Q = dY (b = co), q = 2p + tap                    -> W[ci][co][ty][tx]
// Replaces the weight half of aten::convolution_backward for models.py:66,69,73-77,127.
//
// Both operands are pixel-major in memory (NHWC) but the MFMA wants 8 consecutive K (= pixels)
// per lane, i.e. the transposed image.  bf16: tiles are staged pixel-major in LDS and fragments are
// read with ds_read_b64_tr_b16 (hardware transpose).  fp32: v_mfma_f32_16x16x4_f32 takes ONE k per
// lane, so a plain ds_read_b32 of [pixel][channel] rows is already the right fragment.
//
// Grid: (a-tiles of 16*MI) x (b-tiles of 64) x nsplit.  Every split walks its share of the
// 128-pixel tiles and writes an fp32 slab; a second kernel sums the slabs in a fixed order
// (deterministic) straight into the parameter-gradient tensor in PyTorch layout.
#include "wgrad_common.h"
#include <algorithm>
#include <mutex>

namespace sx {

template <typename T> struct WgTraits;
template <> struct WgTraits<bf16> { static constexpr int PADB = 16; };
template <> struct WgTraits<float> { static constexpr int PADB = 64; };

// A/B fragment for the 32-pixel k-step starting at tile pixel k0, 16 channels starting at byte
// offset ch_off of each pixel row.  pixel_row(k) maps a tile pixel to its LDS row.
template <typename T> struct Frag;
template <> struct Frag<bf16> {
  typedef bf16x8 V;
  // rowaddr[jb] : LDS byte address this lane supplies for k = 8g + 4jb + q, columns 4p..4p+3
  __device__ static __forceinline__ V read(const char* a0, const char* a1) {
    typedef s16x4 __attribute__((address_space(3))) * lp;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)a1);
    union { struct { s16x4 l, h; } s; bf16x8 v; } u;
    u.s.l = lo; u.s.h = hi;
    return u.v;
  }
};

template <typename T, int MI, int ITEMS>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  typedef typename Elem<T>::Vec Vec;
  constexpr int EPL = Elem<T>::EPL;
  constexpr int BA = MI * 16, BB = 64;
  constexpr int PROW = BA * (int)sizeof(T) + WgTraits<T>::PADB;
  constexpr int QROW = BB * (int)sizeof(T) + WgTraits<T>::PADB;
  extern __shared__ __align__(16) char smem[];
  char* sP = smem;
  char* sQ = smem + a.TP * PROW;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int a0 = blockIdx.x * BA, b0 = blockIdx.y * BB;
  const int split = blockIdx.z;
  const int nitems = a.ntaps * a.nbf;

  // per-item constants
  int it_tap_off[ITEMS], it_bf[ITEMS];
  bool it_valid[ITEMS];
#pragma unroll
  for (int t = 0; t < ITEMS; ++t) {
    const int it = wid * ITEMS + t;
    it_valid[t] = it < nitems;
    const int itc = it_valid[t] ? it : 0;
    const int tap = itc / a.nbf;
    it_bf[t] = itc - tap * a.nbf;
    const int tyy = tap / a.tapw, txx = tap - tyy * a.tapw;
    it_tap_off[t] = (tyy * a.LW + txx) * QROW;
  }

  f32x4 acc[ITEMS][MI];
#pragma unroll
  for (int t = 0; t < ITEMS; ++t)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* __restrict__ Pg = reinterpret_cast<const T*>(a.P);
  const T* __restrict__ Qg = reinterpret_cast<const T*>(a.Q);
  constexpr int PPIECES = BA * (int)sizeof(T) / 16, QPIECES = BB * (int)sizeof(T) / 16;

  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
  // fp32: the next tile's pieces are fetched into registers under this tile's MFMAs (the loop below was load -> barrier ->
  // multiply with nothing in flight in between); piece coordinates are tile-independent and decoded once
  constexpr bool PREFETCH = sizeof(T) == 4;
  constexpr int NPP = PREFETCH ? 4 : 1, NPQ = PREFETCH ? 8 : 1;      // pieces per thread: <= 64 px x 16 / 256, <= 128 px x 16 / 256
  Vec pre_p[NPP], pre_q[NPQ];
  int pp_y[NPP], pp_x[NPP], pq_y[NPQ], pq_x[NPQ];
  const int np_p = a.TP * PPIECES, np_q = a.LH * a.LW * QPIECES;
  if constexpr (PREFETCH) {
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int p = tid + i * 256, m = p / PPIECES;
      pp_y[i] = p < np_p ? (m >> a.tw_log2) : (1 << 28);       // out-of-range pieces fail every bounds test
      pp_x[i] = m & (a.TW - 1);
    }
#pragma unroll
    for (int i = 0; i < NPQ; ++i) {
      const int p = tid + i * 256, r = p / QPIECES;
      pq_y[i] = p < np_q ? r / a.LW : (1 << 28);
      pq_x[i] = r - (r / a.LW) * a.LW;
    }
  }
  auto fetch = [&](int tile) {
    int tt = tile;
    const int tx_i = tt % a.tiles_x; tt /= a.tiles_x;
    const int ty_i = tt % a.tiles_y; tt /= a.tiles_y;
    const int bimg = tt;
    const int ty0 = ty_i * a.TH, tx0 = tx_i * a.TW;
    const int qy0 = ty0 * a.q_scale - a.pad, qx0 = tx0 * a.q_scale - a.pad;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int py = ty0 + pp_y[i], px = tx0 + pp_x[i];
      const int c = a0 + ((tid + i * 256) % PPIECES) * EPL;
      Vec v = {};
      if (py < a.H && px < a.W && c < a.Ca)
        v = *reinterpret_cast<const Vec*>(Pg + ((int64_t)(bimg * a.H + py) * a.W + px) * a.p_ld + c);
      pre_p[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NPQ; ++i) {
      const int qy = qy0 + pq_y[i], qx = qx0 + pq_x[i];
      const int c = b0 + ((tid + i * 256) % QPIECES) * EPL;
      Vec v = {};
      if (qy >= 0 && qy < a.Hq && qx >= 0 && qx < a.Wq && c < a.Cb)
        v = *reinterpret_cast<const Vec*>(Qg + ((int64_t)(bimg * a.Hq + qy) * a.Wq + qx) * a.q_ld + c);
      pre_q[i] = v;
    }
  };
  if constexpr (PREFETCH) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    if constexpr (PREFETCH) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NPP; ++i) {
        const int p = tid + i * 256;
        if (p < np_p) *reinterpret_cast<Vec*>(sP + (p / PPIECES) * PROW + (p % PPIECES) * 16) = pre_p[i];
      }
#pragma unroll
      for (int i = 0; i < NPQ; ++i) {
        const int p = tid + i * 256;
        if (p < np_q) *reinterpret_cast<Vec*>(sQ + (p / QPIECES) * QROW + (p % QPIECES) * 16) = pre_q[i];
      }
      __syncthreads();
      if (tile + 1 < t_end) fetch(tile + 1);
    } else {
    int tt = tile;
    const int tx_i = tt % a.tiles_x; tt /= a.tiles_x;
    const int ty_i = tt % a.tiles_y; tt /= a.tiles_y;
    const int bimg = tt;
    const int ty0 = ty_i * a.TH, tx0 = tx_i * a.TW;
    __syncthreads();
    for (int p = tid; p < a.TP * PPIECES; p += 256) {
      const int m = p / PPIECES, q = p - m * PPIECES;
      const int py = ty0 + (m >> a.tw_log2), px = tx0 + (m & (a.TW - 1));
      const int c = a0 + q * EPL;
      Vec v = {};
      if (py < a.H && px < a.W && c < a.Ca)
        v = *reinterpret_cast<const Vec*>(Pg + ((int64_t)(bimg * a.H + py) * a.W + px) * a.p_ld + c);
      *reinterpret_cast<Vec*>(sP + m * PROW + q * 16) = v;
    }
    const int qy0 = ty0 * a.q_scale - a.pad, qx0 = tx0 * a.q_scale - a.pad;
    for (int p = tid; p < a.LH * a.LW * QPIECES; p += 256) {
      const int r = p / QPIECES, q = p - r * QPIECES;
      const int ly = r / a.LW, lx = r - ly * a.LW;
      const int qy = qy0 + ly, qx = qx0 + lx;
      const int c = b0 + q * EPL;
      Vec v = {};
      if (qy >= 0 && qy < a.Hq && qx >= 0 && qx < a.Wq && c < a.Cb)
        v = *reinterpret_cast<const Vec*>(Qg + ((int64_t)(bimg * a.Hq + qy) * a.Wq + qx) * a.q_ld + c);
      *reinterpret_cast<Vec*>(sQ + r * QROW + q * 16) = v;
    }
    __syncthreads();
    }

    if constexpr (sizeof(T) == 2) {
      const int q4 = n >> 2, p4 = n & 3;
      for (int ks = 0; ks < (a.TP >> 5); ++ks) {
        // this lane supplies rows k = ks*32 + 8g + 4jb + q4, jb = 0,1
        const int k0 = ks * 32 + 8 * g + q4, k1 = k0 + 4;
        const char* pa0 = sP + k0 * PROW + p4 * 8;
        const char* pa1 = sP + k1 * PROW + p4 * 8;
        bf16x8 A[MI];
#pragma unroll
        for (int j = 0; j < MI; ++j) A[j] = Frag<bf16>::read(pa0 + j * 32, pa1 + j * 32);
        const int r0 = (((k0 >> a.tw_log2) * a.q_scale) * a.LW + (k0 & (a.TW - 1)) * a.q_scale) * QROW + p4 * 8;
        const int r1 = (((k1 >> a.tw_log2) * a.q_scale) * a.LW + (k1 & (a.TW - 1)) * a.q_scale) * QROW + p4 * 8;
#pragma unroll
        for (int t = 0; t < ITEMS; ++t) {
          // every lane issues the transposed read (EXEC must be full); invalid items read item 0
          bf16x8 Bf = Frag<bf16>::read(sQ + r0 + it_tap_off[t] + it_bf[t] * 32, sQ + r1 + it_tap_off[t] + it_bf[t] * 32);
#pragma unroll
          for (int j = 0; j < MI; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[j], Bf, acc[t][j], 0, 0, 0);
        }
      }
    } else {
#pragma unroll 4
      for (int ks = 0; ks < (a.TP >> 2); ++ks) {
        const int k = ks * 4 + g;      // one pixel per lane group
        float A[MI];
#pragma unroll
        for (int j = 0; j < MI; ++j) A[j] = *reinterpret_cast<const float*>(sP + k * PROW + (j * 16 + n) * 4);
        const int r = (((k >> a.tw_log2) * a.q_scale) * a.LW + (k & (a.TW - 1)) * a.q_scale) * QROW + n * 4;
#pragma unroll
        for (int t = 0; t < ITEMS; ++t) {
          const float Bv = *reinterpret_cast<const float*>(sQ + r + it_tap_off[t] + it_bf[t] * 64);
#pragma unroll
          for (int j = 0; j < MI; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], Bv, acc[t][j], 0, 0, 0);
        }
      }
    }
  }

  // slab[split][tap][a][b]: lane holds rows a = 4g + r, column b = n of each 16x16 tile
  float* slab = a.slab + (int64_t)split * a.ntaps * a.CaPad * a.CbPad;
#pragma unroll
  for (int t = 0; t < ITEMS; ++t) {
    if (!it_valid[t]) continue;
    const int it = wid * ITEMS + t;
    const int tap = it / a.nbf, bf = it - tap * a.nbf;
    const int bcol = b0 + bf * 16 + n;
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int arow = a0 + j * 16 + 4 * g + r;
        if (arow < a.CaPad && bcol < a.CbPad)
          slab[((int64_t)tap * a.CaPad + arow) * a.CbPad + bcol] = acc[t][j][r];
      }
  }
}

// fp32 tier on the bf16 matrix cores (see conv_x3.hip): both operand tiles are split into bf16 hi / lo planes on their way
// into LDS (pixel-major rows, exactly the images of the bf16 branch above, two of each), fragments come through
// ds_read_b64_tr_b16 and every product block is three MFMAs: lo*hi + hi*lo + hi*hi.  The next tile's pieces are fetched
// into registers under this tile's MFMAs.
// NW waves per block: the 36 items of a 64 x 64 x 9 problem take 9 x 16 accumulator registers per wave on four waves (spills at
// two blocks per CU); on eight waves they take 5 x 16.
// (Measured for the eight-wave multi-problem launch and dropped: a second LDS image with one barrier per tile, 3.645 against
// 3.645-3.65 ms per fp32 step; plus a second register set, loads two tiles ahead: 3.913 against 3.902 ms in one box.)
template <int MI, int ITEMS, int NW>
__device__ __forceinline__ void wgrad_x3_body(const WgradArgs& a, int blk_x, int blk_y, int blk_z) {
  constexpr int NT = NW * 64;
  constexpr int BA = MI * 16, BB = 64;
  constexpr int PROW = BA * 2 + 16, QROW = BB * 2 + 16;
  constexpr int PPIECES = BA / 4, QPIECES = BB / 4;        // 16-byte pieces of 4 fp32 per pixel
  constexpr int NPP = 1024 / NT, NPQ = (MI == 2 ? 3072 : 2048) / NT;      // MI = 2 also takes 128-pixel tiles (180-pixel halo)
  extern __shared__ __align__(16) char smem[];
  const int p_img = a.TP * PROW, q_img = a.LH * a.LW * QROW;      // one plane of each operand

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int a0 = blk_x * BA, b0 = blk_y * BB;
  const int split = blk_z;
  const int nitems = a.ntaps * a.nbf;

  int it_tap_off[ITEMS], it_bf[ITEMS];
  bool it_valid[ITEMS];
#pragma unroll
  for (int t = 0; t < ITEMS; ++t) {
    const int it = wid * ITEMS + t;
    it_valid[t] = it < nitems;
    const int itc = it_valid[t] ? it : 0;
    const int tap = itc / a.nbf;
    it_bf[t] = itc - tap * a.nbf;
    const int tyy = tap / a.tapw, txx = tap - tyy * a.tapw;
    it_tap_off[t] = (tyy * a.LW + txx) * QROW;
  }
  f32x4 acc[ITEMS][MI];
#pragma unroll
  for (int t = 0; t < ITEMS; ++t)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* __restrict__ Pg = reinterpret_cast<const float*>(a.P);
  const float* __restrict__ Qg = reinterpret_cast<const float*>(a.Q);
  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
  struct Pre { f32x4 p[NPP]; f32x4 q[NPQ]; };      // one tile's pieces on their way from HBM to LDS
  int pp_y[NPP], pp_x[NPP], pq_y[NPQ], pq_x[NPQ];
  const int np_p = a.TP * PPIECES, np_q = a.LH * a.LW * QPIECES;
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int p = tid + i * NT, m = p / PPIECES;
    pp_y[i] = p < np_p ? (m >> a.tw_log2) : (1 << 28);       // out-of-range pieces fail every bounds test
    pp_x[i] = m & (a.TW - 1);
  }
#pragma unroll
  for (int i = 0; i < NPQ; ++i) {
    const int p = tid + i * NT, r = p / QPIECES;
    pq_y[i] = p < np_q ? r / a.LW : (1 << 28);
    pq_x[i] = r - (r / a.LW) * a.LW;
  }
  auto fetch = [&](Pre& R, int tile) {
    int tt = tile;
    const int tx_i = tt % a.tiles_x; tt /= a.tiles_x;
    const int ty_i = tt % a.tiles_y; tt /= a.tiles_y;
    const int bimg = tt;
    const int ty0 = ty_i * a.TH, tx0 = tx_i * a.TW;
    const int qy0 = ty0 * a.q_scale - a.pad, qx0 = tx0 * a.q_scale - a.pad;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int py = ty0 + pp_y[i], px = tx0 + pp_x[i];
      const int c = a0 + ((tid + i * NT) % PPIECES) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (py < a.H && px < a.W && c < a.Ca)
        v = *reinterpret_cast<const f32x4*>(Pg + ((int64_t)(bimg * a.H + py) * a.W + px) * a.p_ld + c);
      R.p[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NPQ; ++i) {
      const int qy = qy0 + pq_y[i], qx = qx0 + pq_x[i];
      const int c = b0 + ((tid + i * NT) % QPIECES) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (qy >= 0 && qy < a.Hq && qx >= 0 && qx < a.Wq && c < a.Cb)
        v = *reinterpret_cast<const f32x4*>(Qg + ((int64_t)(bimg * a.Hq + qy) * a.Wq + qx) * a.q_ld + c);
      R.q[i] = v;
    }
  };
  auto put = [&](char* hi_img, char* lo_img, int off, const f32x4& v) {
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h[e] = (bf16)v[e]; l[e] = (bf16)(v[e] - (float)h[e]); }
    *reinterpret_cast<bf16x4*>(hi_img + off) = h;
    *reinterpret_cast<bf16x4*>(lo_img + off) = l;
  };
  const int q4 = n >> 2, p4 = n & 3;
  auto put_tile = [&](const Pre& R, char* buf) {
    char* sPh = buf;
    char* sPl = sPh + p_img;
    char* sQh = sPl + p_img;
    char* sQl = sQh + q_img;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int p = tid + i * NT;
      if (p < np_p) put(sPh, sPl, (p / PPIECES) * PROW + (p % PPIECES) * 8, R.p[i]);
    }
#pragma unroll
    for (int i = 0; i < NPQ; ++i) {
      const int p = tid + i * NT;
      if (p < np_q) put(sQh, sQl, (p / QPIECES) * QROW + (p % QPIECES) * 8, R.q[i]);
    }
  };
  auto mma_tile = [&](const char* buf) {
    const char* sPh = buf;
    const char* sPl = sPh + p_img;
    const char* sQh = sPl + p_img;
    const char* sQl = sQh + q_img;
    for (int ks = 0; ks < (a.TP >> 5); ++ks) {
      // this lane supplies rows k = ks*32 + 8g + 4jb + q4, jb = 0,1
      const int k0 = ks * 32 + 8 * g + q4, k1 = k0 + 4;
      const int pa0 = k0 * PROW + p4 * 8, pa1 = k1 * PROW + p4 * 8;
      bf16x8 Ah[MI], Al[MI];
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        Ah[j] = Frag<bf16>::read(sPh + pa0 + j * 32, sPh + pa1 + j * 32);
        Al[j] = Frag<bf16>::read(sPl + pa0 + j * 32, sPl + pa1 + j * 32);
      }
      const int r0 = (((k0 >> a.tw_log2) * a.q_scale) * a.LW + (k0 & (a.TW - 1)) * a.q_scale) * QROW + p4 * 8;
      const int r1 = (((k1 >> a.tw_log2) * a.q_scale) * a.LW + (k1 & (a.TW - 1)) * a.q_scale) * QROW + p4 * 8;
      // every lane issues the transposed reads (EXEC must be full); invalid items read item 0.  The fragments of item t+1
      // are read under the MFMAs of item t (back to back, each item exposed the LDS latency: 18 times per tile)
      bf16x8 Bh[2], Bl[2];
      auto read_item = [&](int t, int slot) {
        const int o0 = r0 + it_tap_off[t] + it_bf[t] * 32, o1 = r1 + it_tap_off[t] + it_bf[t] * 32;
        Bh[slot] = Frag<bf16>::read(sQh + o0, sQh + o1);
        Bl[slot] = Frag<bf16>::read(sQl + o0, sQl + o1);
      };
      read_item(0, 0);
#pragma unroll
      for (int t = 0; t < ITEMS; ++t) {
        const int sl = t & 1;
        if (t + 1 < ITEMS) read_item(t + 1, sl ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al[j], Bh[sl], acc[t][j], 0, 0, 0);
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[j], Bl[sl], acc[t][j], 0, 0, 0);
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[j], Bh[sl], acc[t][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  Pre R;
  if (t_begin < t_end) fetch(R, t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();
    put_tile(R, smem);
    __syncthreads();
    if (tile + 1 < t_end) fetch(R, tile + 1);
    mma_tile(smem);
  }

  float* slab = a.slab + (int64_t)split * a.ntaps * a.CaPad * a.CbPad;
#pragma unroll
  for (int t = 0; t < ITEMS; ++t) {
    if (!it_valid[t]) continue;
    const int it = wid * ITEMS + t;
    const int tap = it / a.nbf, bf = it - tap * a.nbf;
    const int bcol = b0 + bf * 16 + n;
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int arow = a0 + j * 16 + 4 * g + r;
        if (arow < a.CaPad && bcol < a.CbPad)
          slab[((int64_t)tap * a.CaPad + arow) * a.CbPad + bcol] = acc[t][j][r];
      }
  }
}

template <int MI, int ITEMS, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void wgrad_x3_kernel(WgradArgs a) {
  wgrad_x3_body<MI, ITEMS, NW>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Every fp32-tier problem of a backward half in ONE launch (the bf16 tier's wgrad_ws_multi_kernel, applied to this kernel):
// block ranges per problem, all on eight waves (the item count per wave follows from taps x b fragments), longest blocks first.
static constexpr int WG_X3_MULTI_MAX = 16;
struct WgX3MultiArgs {
  WgradArgs p[WG_X3_MULTI_MAX];
  int variant[WG_X3_MULTI_MAX], a_tiles[WG_X3_MULTI_MAX], ab_tiles[WG_X3_MULTI_MAX];
  int blk0[WG_X3_MULTI_MAX + 1];
  int n;
};
static_assert(sizeof(WgX3MultiArgs) <= 4096, "kernel-argument segment");
#define SX_X3_VARIANTS(X) X(2, 1) X(2, 2) X(2, 3) X(2, 4) X(2, 5) X(4, 1) X(4, 2) X(4, 3) X(4, 4) X(4, 5)
__global__ __launch_bounds__(512, 1) void wgrad_x3_multi_kernel(const WgX3MultiArgs) {
  typedef const WgX3MultiArgs __attribute__((address_space(4))) * TablePtr;      // (see wgrad_ws_multi_kernel)
  TablePtr A = (TablePtr)__builtin_amdgcn_kernarg_segment_ptr();
  int pi = 0;
  const int n = A->n;
  while (pi + 1 < n && (int)blockIdx.x >= A->blk0[pi + 1]) ++pi;
  const int id = (int)blockIdx.x - A->blk0[pi];
  WgradArgs a;
  __builtin_memcpy(&a, &A->p[pi], sizeof(a));
  const int at = A->a_tiles[pi], ab = A->ab_tiles[pi];
  const int split = id / ab, r = id - split * ab;
  const int by = r / at, bx = r - by * at;
  switch (A->variant[pi]) {
#define SX_X3_BODY(mi, items) case mi * 16 + items: wgrad_x3_body<mi, items, 8>(a, bx, by, split); break;
    SX_X3_VARIANTS(SX_X3_BODY)
#undef SX_X3_BODY
    default: break;
  }
}

// out[(a*Cb + b)*ntaps + tap] (+)= sum_s slab[s][tap][a][b]   (PyTorch layouts: conv W[co][ci][tap], convT W[ci][co][tap])
// One launch reduces the slabs of many layers (descriptor table in the kernel arguments).  Unit of work = one
// 256-byte segment of a slab row (64 consecutive b) x all splits: 16 threads x float4 along b, 16 split lanes whose
// partial sums are combined through LDS in a fixed order (deterministic).
struct WgReduceArgs { WgReduceDesc d[WG_MAX_REDUCE]; int n; };

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(WgReduceArgs A) {
  int di = 0;
  while (di + 1 < A.n && (int)blockIdx.x >= A.d[di + 1].blk0) ++di;
  wgrad_reduce_block(A.d[di], (int)blockIdx.x - A.d[di].blk0);
}

int wgrad_reduce_multi(const WgReduceDesc* descs, int n, hipStream_t st) {
  for (int i0 = 0; i0 < n; i0 += WG_MAX_REDUCE) {
    WgReduceArgs A;
    A.n = std::min(WG_MAX_REDUCE, n - i0);
    int blk = 0;
    for (int i = 0; i < A.n; ++i) {
      A.d[i] = descs[i0 + i];
      A.d[i].blk0 = blk;
      A.d[i].nblk = wgrad_reduce_blocks(A.d[i]);
      blk += A.d[i].nblk;
    }
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blk), dim3(256), 0, st, A);
    SX_LAUNCH_CHECK("wgrad_reduce");
  }
  return SX_OK;
}

struct WgPlan { int MI, nbf, items, a_tiles, b_tiles, nsplit, tiles_per_split; size_t ws_bytes; };

static WgPlan wgrad_plan(int mode, int B, int H, int W, int Ca, int Cb, const WgradArgs& g, int blocks = 1024) {
  WgPlan p;
  const int CaPad = round_up(Ca, 16), CbPad = round_up(Cb, 16);
  p.MI = CaPad >= 64 ? 4 : 2;
  p.a_tiles = cdiv(CaPad, p.MI * 16);
  p.b_tiles = cdiv(CbPad, 64);
  p.nbf = std::min(4, CbPad / 16);
  p.items = cdiv(g.ntaps * p.nbf, 4);
  const int ntiles = B * g.tiles_x * g.tiles_y;
  const size_t slab1 = (size_t)g.ntaps * CaPad * CbPad * sizeof(float);
  // ~4 blocks per CU: with the 64-pixel fp32 tiles two blocks share a CU, one loading while the other multiplies
  // (split-bf16 arithmetic: the MFMAs are 5x cheaper, the slab traffic is not -- fewer, longer splits, `blocks` in total)
  int ns = std::max(1, blocks / (p.a_tiles * p.b_tiles));
  ns = std::min(ns, ntiles);
  ns = std::min<int>(ns, std::max<size_t>(1, ((size_t)32 << 20) / slab1));
  p.tiles_per_split = cdiv(ntiles, ns);
  p.nsplit = cdiv(ntiles, p.tiles_per_split);
  p.ws_bytes = slab1 * p.nsplit;
  return p;
}

static int wgrad_geom(int mode, int dtype, int B, int H, int W, WgradArgs& a, int tp = 0) {
  switch (mode) {
    case SX_CONV3: a.ntaps = 9; a.tapw = 3; a.q_scale = 1; a.pad = 1; break;
    case SX_CONV1: a.ntaps = 1; a.tapw = 1; a.q_scale = 1; a.pad = 0; break;
    case SX_CONVT_FWD: a.ntaps = 4; a.tapw = 2; a.q_scale = 2; a.pad = 0; break;
    default: set_error("sx_conv_wgrad: bad mode %d", mode); return SX_ERR_INVALID;
  }
  a.tw_log2 = W > 8 ? 4 : (W > 4 ? 3 : 2);
  a.TW = 1 << a.tw_log2;
  // the transposed conv's Q tile is 4x the pixels of the P tile: keep the fp32 image inside LDS
  // fp32: 64-pixel tiles keep a block's two images under 80 KB of LDS, i.e. two blocks per CU (8.68 -> 8.29 ms per step)
  a.TP = (mode == SX_CONVT_FWD && dtype == SX_F32) ? 32 : (dtype == SX_F32 ? 64 : 128);
  if (tp) a.TP = tp;
  a.TH = a.TP / a.TW;
  const int taph = a.ntaps / a.tapw;
  a.LW = (a.TW - 1) * a.q_scale + a.tapw;
  a.LH = (a.TH - 1) * a.q_scale + taph;
  a.tiles_x = cdiv(W, a.TW);
  a.tiles_y = cdiv(H, a.TH);
  a.ntiles = B * a.tiles_x * a.tiles_y;
  return SX_OK;
}

template <typename T, int MI, int ITEMS>
static int launch_wgrad(const WgradArgs& a, const WgPlan& p, hipStream_t st) {
  constexpr int PROW = MI * 16 * (int)sizeof(T) + WgTraits<T>::PADB;
  constexpr int QROW = 64 * (int)sizeof(T) + WgTraits<T>::PADB;
  const size_t smem = (size_t)a.TP * PROW + (size_t)a.LH * a.LW * QROW;
  if (smem > 160 * 1024) { set_error("wgrad tile needs %zu B of LDS", smem); return SX_ERR_UNSUPPORTED; }
  if (sizeof(T) == 4 && (a.TP * MI * 4 > 4 * 256 || a.LH * a.LW * 16 > 8 * 256)) {     // the kernel's register-prefetch slots
    set_error("wgrad: fp32 tile of %d + %d pixels exceeds the prefetch slots", a.TP, a.LH * a.LW);
    return SX_ERR_UNSUPPORTED;
  }
  auto kern = wgrad_kernel<T, MI, ITEMS>;
  static std::atomic<unsigned long long> done{0};
  SX_TRY(ensure_max_lds(reinterpret_cast<const void*>(kern), done, "hipFuncSetAttribute(wgrad)"));
  hipLaunchKernelGGL(kern, dim3(p.a_tiles, p.b_tiles, p.nsplit), dim3(256), smem, st, a);
  SX_LAUNCH_CHECK("wgrad");
  return SX_OK;
}

template <typename T, int MI>
static int wgrad_items(const WgradArgs& a, const WgPlan& p, hipStream_t st) {
  switch (p.items) {
    case 1: return launch_wgrad<T, MI, 1>(a, p, st);
    case 2: return launch_wgrad<T, MI, 2>(a, p, st);
    case 3: return launch_wgrad<T, MI, 3>(a, p, st);
    case 4: return launch_wgrad<T, MI, 4>(a, p, st);
    case 5: return launch_wgrad<T, MI, 5>(a, p, st);
    case 7: return launch_wgrad<T, MI, 7>(a, p, st);      // 9 taps x 3 b fragments (33..48 input channels)
    case 9: return launch_wgrad<T, MI, 9>(a, p, st);
  }
  set_error("wgrad: unexpected item count %d", p.items);
  return SX_ERR_UNSUPPORTED;
}

template <int MI, int ITEMS, int NW = 4>
static int launch_wgrad_x3(const WgradArgs& a, const WgPlan& p, hipStream_t st) {
  constexpr int PROW = MI * 16 * 2 + 16, QROW = 64 * 2 + 16;
  const size_t smem = 2 * ((size_t)a.TP * PROW + (size_t)a.LH * a.LW * QROW);
  if (smem > 160 * 1024) { set_error("wgrad tile needs %zu B of LDS", smem); return SX_ERR_UNSUPPORTED; }
  if (a.TP * MI * 4 > 4 * 256 || a.LH * a.LW * 16 > (MI == 2 ? 12 : 8) * 256 || a.TP % 32) {     // the kernel's register-prefetch slots
    set_error("wgrad: fp32 tile of %d + %d pixels exceeds the prefetch slots", a.TP, a.LH * a.LW);
    return SX_ERR_UNSUPPORTED;
  }
  auto kern = wgrad_x3_kernel<MI, ITEMS, NW>;
  static std::atomic<unsigned long long> done{0};
  SX_TRY(ensure_max_lds(reinterpret_cast<const void*>(kern), done, "hipFuncSetAttribute(wgrad_x3)"));
  hipLaunchKernelGGL(kern, dim3(p.a_tiles, p.b_tiles, p.nsplit), dim3(NW * 64), smem, st, a);
  SX_LAUNCH_CHECK("wgrad_x3");
  return SX_OK;
}

template <int MI>
static int wgrad_items_x3(const WgradArgs& a, const WgPlan& p, hipStream_t st) {
  switch (p.items) {
    case 1: return launch_wgrad_x3<MI, 1>(a, p, st);
    case 2: return launch_wgrad_x3<MI, 2>(a, p, st);
    case 3: return launch_wgrad_x3<MI, 3>(a, p, st);
    case 4: return launch_wgrad_x3<MI, 4>(a, p, st);
    case 5: return launch_wgrad_x3<MI, 5>(a, p, st);
    case 7: return launch_wgrad_x3<MI, 7>(a, p, st);      // 9 taps x 3 b fragments (33..48 input channels)
    case 9:
      if constexpr (MI == 4) return launch_wgrad_x3<MI, 5, 8>(a, p, st);      // 36 items on eight waves
      else return launch_wgrad_x3<MI, 9>(a, p, st);
  }
  set_error("wgrad: unexpected item count %d", p.items);
  return SX_ERR_UNSUPPORTED;
}

int wgrad_x3_launch(WgX3Launch* l, int n, hipStream_t st) {
  std::stable_sort(l, l + n, [](const WgX3Launch& x, const WgX3Launch& y) { return x.cost > y.cost; });
  static std::atomic<unsigned long long> done{0};
  SX_TRY(ensure_max_lds(reinterpret_cast<const void*>(wgrad_x3_multi_kernel), done, "hipFuncSetAttribute(wgrad_x3_multi)"));
  static const int per_launch = std::max(1, std::min(WG_X3_MULTI_MAX, env_int("SX_X3_WG_PER_LAUNCH", WG_X3_MULTI_MAX)));   // diagnostic: 1 = a launch per problem
  for (int i0 = 0; i0 < n; i0 += per_launch) {
    WgX3MultiArgs A;
    memset(&A, 0, sizeof(A));
    A.n = std::min(per_launch, n - i0);
    int blk = 0, smem = 0;
    for (int i = 0; i < A.n; ++i) {
      const WgX3Launch& x = l[i0 + i];
      A.p[i] = x.a; A.variant[i] = x.variant; A.a_tiles[i] = x.a_tiles; A.ab_tiles[i] = x.a_tiles * x.b_tiles;
      A.blk0[i] = blk;
      blk += x.a_tiles * x.b_tiles * x.nsplit;
      smem = std::max(smem, x.smem);
    }
    A.blk0[A.n] = blk;
    hipLaunchKernelGGL(wgrad_x3_multi_kernel, dim3(blk), dim3(512), (size_t)smem, st, A);
    SX_LAUNCH_CHECK("wgrad_x3_multi");
  }
  return SX_OK;
}

}  // namespace sx

using namespace sx;

extern "C" int64_t sx_conv_wgrad_workspace_bytes(int mode, int B, int H, int W, int Ca, int Cb) {
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  // slab size does not depend on dtype; the tile count does, so size for the finer (fp32) tiling.  The bf16 tier
  // (wgrad_ws.hip) may fall back to the generic kernel, so the workspace covers both.
  if (wgrad_geom(mode, SX_F32, B, H, W, a) != SX_OK) return -1;
  return std::max<int64_t>((int64_t)wgrad_plan(mode, B, H, W, Ca, Cb, a).ws_bytes, wgrad_ws_slab_bytes(mode, B, H, W, Ca, Cb));
}

namespace sx {
int wgrad_launch(int dtype, int mode, int B, int H, int W, int Ca, int Cb, const void* P, int p_ld, const void* Q,
                 int q_ld, int Hq, int Wq, void* slab, int64_t slab_bytes, WgReduceDesc* desc, hipStream_t st, WgWsLaunch* defer,
                 WgX3Launch* defer_x3) {
  SX_REQUIRE(P && Q && slab && desc, "sx_conv_wgrad: null pointer");
  SX_REQUIRE(dtype == SX_F32 || dtype == SX_BF16, "sx_conv_wgrad: bad dtype %d", dtype);
  const int EPL = dtype == SX_BF16 ? 8 : 4;
  if (Ca % EPL || Cb % EPL || p_ld % EPL || q_ld % EPL) {
    set_error("sx_conv_wgrad: channel counts must be multiples of %d: Ca=%d Cb=%d p_ld=%d q_ld=%d", EPL, Ca, Cb, p_ld, q_ld);
    return SX_ERR_UNSUPPORTED;
  }
  SX_REQUIRE(((uintptr_t)P % 16) == 0 && ((uintptr_t)Q % 16) == 0, "sx_conv_wgrad: misaligned pointer");
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  SX_TRY(wgrad_geom(mode, dtype, B, H, W, a));
  if (mode == SX_CONVT_FWD) SX_REQUIRE(Hq >= 2 * H && Wq >= 2 * W, "sx_conv_wgrad: convT Q dims");
  else SX_REQUIRE(Hq == H && Wq == W, "sx_conv_wgrad: Q dims");
  if (dtype == SX_BF16 && (int64_t)Hq * Wq * std::max(p_ld, q_ld) * 2 < ((int64_t)1 << 30)) {
    const int64_t need = wgrad_ws_slab_bytes(mode, B, H, W, Ca, Cb);
    if (need > 0) {
      SX_REQUIRE(need <= slab_bytes, "sx_conv_wgrad: workspace too small (%lld < %lld)", (long long)slab_bytes, (long long)need);
      if (defer) {          // the caller batches persistent-tier problems into one launch (wgrad_ws_launch)
        const int rc = wgrad_ws_prepare(mode, B, H, W, Ca, Cb, P, p_ld, Q, q_ld, Hq, Wq, (float*)slab, desc, defer);
        if (rc == 0) return WG_DEFERRED;
        if (rc < 0) return rc;
      } else {
        const int rc = wgrad_ws_try(mode, B, H, W, Ca, Cb, P, p_ld, Q, q_ld, Hq, Wq, (float*)slab, desc, st);
        if (rc <= 0) return rc;
      }
    }
  }
  // experiment knob: SX_X3_WGRAD = 0 / 1 forces exact / split arithmetic for the weight gradients alone (default: as the convolutions)
  const int wg_force = env_int("SX_X3_WGRAD", -1);
  const bool wsplit = wg_force < 0 ? fp32_split_process() != 0 : wg_force != 0;
  static const int x3_blocks = env_int("SX_X3_WG_BLOCKS", 512);
  // split-bf16 arithmetic, <= 48 output rows (the level-0 layers): 128-pixel tiles.  A 64-pixel tile's MFMAs (0.8 us) are
  // shorter than the HBM latency of the next tile's register prefetch, and its halo over-fetch is 1.7x against 1.4x
  static const int x3_tp = env_int("SX_X3_WG_TP", 128);
  if (dtype == SX_F32 && wsplit && mode != SX_CONVT_FWD && round_up(Ca, 16) < 64 && x3_tp != 64) {
    SX_TRY(wgrad_geom(mode, dtype, B, H, W, a, x3_tp));
    if (a.LH * a.LW * 16 > 12 * 256 || a.TP % 32) SX_TRY(wgrad_geom(mode, dtype, B, H, W, a));      // 4-wide grids: the halo of a 32 x 4 tile exceeds the staging slots
  }
  // a problem that will share its launch with the rest of its backward half need not fill the chip alone: fewer, longer
  // splits (less slab traffic), SX_X3_WG_BATCH_BLOCKS per problem
  static const int x3_multi = env_int("SX_X3_WG_MULTI", 1), x3_batch_blocks = env_int("SX_X3_WG_BATCH_BLOCKS", 128);
  const bool batch = defer_x3 && dtype == SX_F32 && wsplit && x3_multi;
  WgPlan p = wgrad_plan(mode, B, H, W, Ca, Cb, a, (dtype == SX_F32 && wsplit) ? (batch ? x3_batch_blocks : x3_blocks) : 1024);
  SX_REQUIRE((int64_t)p.ws_bytes <= slab_bytes, "sx_conv_wgrad: workspace too small (%lld < %zu)",
             (long long)slab_bytes, p.ws_bytes);
  a.P = P; a.Q = Q; a.slab = (float*)slab;
  a.B = B; a.H = H; a.W = W; a.Hq = Hq; a.Wq = Wq; a.p_ld = p_ld; a.q_ld = q_ld;
  a.Ca = Ca; a.Cb = Cb; a.CaPad = round_up(Ca, 16); a.CbPad = round_up(Cb, 16);
  a.nbf = p.nbf; a.tiles_per_split = p.tiles_per_split;
  if (batch) {
    const int PROW = p.MI * 16 * 2 + 16, QROW = 64 * 2 + 16;
    const size_t smem = 2 * ((size_t)a.TP * PROW + (size_t)a.LH * a.LW * QROW);
    const int items8 = cdiv(a.ntaps * p.nbf, 8);
    const bool fits = smem <= 160 * 1024 && a.TP * p.MI * 4 <= 4 * 256 && a.LH * a.LW * 16 <= (p.MI == 2 ? 12 : 8) * 256 &&
                      a.TP % 32 == 0 && items8 >= 1 && items8 <= 5;
    if (fits) {
      defer_x3->a = a; defer_x3->variant = p.MI * 16 + items8;
      defer_x3->a_tiles = p.a_tiles; defer_x3->b_tiles = p.b_tiles; defer_x3->nsplit = p.nsplit; defer_x3->smem = (int)smem;
      // one block: its tiles' MFMA work at 4 TFLOP/s per CU + its operand bytes at 60 GB/s per CU
      const double px = (double)p.tiles_per_split * a.TP;
      defer_x3->cost = 3.0 * 2.0 * px * a.ntaps * (p.MI * 16) * (p.nbf * 16) / 4e12 +
                       4.0 * p.tiles_per_split * ((double)a.TP * p.MI * 16 + (double)a.LH * a.LW * 64) / 6e10;
      desc->slab = (const float*)slab; desc->nsplit = p.nsplit; desc->ntaps = a.ntaps; desc->CaPad = a.CaPad; desc->CbPad = a.CbPad;
      return WG_DEFERRED_X3;
    }
  }
  int rc;
  if (dtype == SX_BF16) rc = p.MI == 4 ? wgrad_items<bf16, 4>(a, p, st) : wgrad_items<bf16, 2>(a, p, st);
  else if (wsplit) rc = p.MI == 4 ? wgrad_items_x3<4>(a, p, st) : wgrad_items_x3<2>(a, p, st);
  else rc = p.MI == 4 ? wgrad_items<float, 4>(a, p, st) : wgrad_items<float, 2>(a, p, st);
  SX_TRY(rc);
  desc->slab = (const float*)slab; desc->nsplit = p.nsplit; desc->ntaps = a.ntaps; desc->CaPad = a.CaPad; desc->CbPad = a.CbPad;
  return SX_OK;
}
}  // namespace sx

extern "C" int sx_conv_wgrad(int dtype, int mode, int B, int H, int W, int Ca, int Cb, int Ca_out, int Cb_out,
                             const void* P, int p_ld, const void* Q, int q_ld, int Hq, int Wq,
                             float* dw, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  SX_REQUIRE(dw, "sx_conv_wgrad: null pointer");
  SX_REQUIRE(Ca_out > 0 && Ca_out <= Ca && Cb_out > 0 && Cb_out <= Cb, "sx_conv_wgrad: bad output channel counts");
  WgReduceDesc d;
  memset(&d, 0, sizeof(d));
  SX_TRY(wgrad_launch(dtype, mode, B, H, W, Ca, Cb, P, p_ld, Q, q_ld, Hq, Wq, workspace, workspace_bytes, &d, (hipStream_t)stream, nullptr));
  d.out = dw; d.Ca = Ca_out; d.Cb = Cb_out; d.accumulate = accumulate;
  return wgrad_reduce_multi(&d, 1, (hipStream_t)stream);
}

// ------------------------------------------------------------------ sx_conv_wgrad_static
// The same weight gradient for callers that cannot allocate device memory (code under HIP graph capture, tight serving
// loops, harnesses that forbid hipMalloc): the split-K slabs live in a file-scope device array instead of a caller
// workspace.  Every GPU the library runs on holds its own copy of the array (SX_WGRAD_STATIC_BYTES) until the process
// ends; `scratch` below is only the host-side handle of the symbol -- the per-device address comes from
// hipGetSymbolAddress, once per device, and that address is what the kernels receive.
namespace sx {
static __device__ __align__(256) unsigned char scratch[SX_WGRAD_STATIC_BYTES];
static constexpr int MAX_DEVICES = 64;
static void* g_scratch_ptr[MAX_DEVICES];      // device address of `scratch` on each device, filled on first use
static std::mutex g_scratch_mu;

static int scratch_on_current_device(void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
  SX_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "sx_conv_wgrad_static: device ordinal %d outside 0..%d", dev, MAX_DEVICES - 1);
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return hip_fail(e, "hipGetSymbolAddress(scratch)");
    SX_REQUIRE(p && ((uintptr_t)p % 256) == 0, "sx_conv_wgrad_static: bad scratch address");
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return SX_OK;
}
}  // namespace sx

extern "C" int sx_conv_wgrad_static(int dtype, int mode, int B, int H, int W, int Ca, int Cb, int Ca_out, int Cb_out,
                                    const void* P, int p_ld, const void* Q, int q_ld, int Hq, int Wq,
                                    float* dw, int accumulate, void* stream) {
  const int64_t need = sx_conv_wgrad_workspace_bytes(mode, B, H, W, Ca, Cb);
  SX_REQUIRE(need >= 0, "sx_conv_wgrad_static: bad mode %d", mode);
  if (need > (int64_t)SX_WGRAD_STATIC_BYTES) {
    set_error("sx_conv_wgrad_static: this problem needs %lld bytes of split-K slabs, the static scratch holds %lld "
              "(use sx_conv_wgrad with a caller workspace)", (long long)need, (long long)SX_WGRAD_STATIC_BYTES);
    return SX_ERR_UNSUPPORTED;
  }
  void* ptr = nullptr;
  SX_TRY(scratch_on_current_device(&ptr));
  // every slab element the reduction reads is written by the split-K kernel first: nothing to clear between calls
  return sx_conv_wgrad(dtype, mode, B, H, W, Ca, Cb, Ca_out, Cb_out, P, p_ld, Q, q_ld, Hq, Wq, dw, accumulate, ptr,
                       (int64_t)SX_WGRAD_STATIC_BYTES, stream);
}
