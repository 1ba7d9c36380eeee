// Persistent split-K weight-gradient kernel, bf16, gfx950: the fast tier of sx_conv_wgrad for 3x3 / 1x1 convolutions.
//
//     dW[tap][a][b] = sum over pixels p of  P[p][a] * Q[p + tap - pad][b]        (P = dY, Q = layer input)
//     ConvTranspose2d(2,2) (TAPS = 4): P = layer input (a = ci), Q = dY, Q pixel = 2p + tap: the LDS-DMA de-interleaves
//     the four parity classes of the dY tile into four sub-images with P's pixel order, so a tap is a plain 1x1 product.
//
// The reduction dimension is the pixel index (up to 524k per layer at batch 64) and the result is tiny
// (taps x Ca x Cb), so a block keeps ALL its accumulators in registers while it walks a contiguous range of
// 128-pixel tiles, and writes one fp32 slab at the very end:
//   * both operand tiles are filled by LDS-DMA (raw buffer loads, 16 bytes per lane straight into LDS, hardware
//     zero fill outside the image) into a ring of NBUF stages: no staging registers, no address arithmetic beyond
//     one 32-bit add per piece, loads of the next tiles stay in flight under the MFMAs of the current one;
//   * pixel-major tiles are consumed through ds_read_b64_tr_b16 (hardware transpose: 8 consecutive pixels of one
//     channel per lane = the MFMA k layout).  The 16-byte pieces of a pixel are XOR-swizzled by the pixel's
//     position so the 8 rows a 32-lane half reads land on 8 distinct 32-byte bank segments (the DMA applies the
//     permutation for free: it only decides which global piece a lane fetches);
//   * the (a tile, b tile, tap group) space is divided among R blocks that share a pixel range and are placed on
//     the same XCD (same L2), and the pixel range is split S ways with S*R ~ number of CUs: fewer, larger slabs
//     than a plain split-K over every CU (slab traffic, not operand traffic, dominated the first version).
#include "wgrad_common.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace sx {


template <int N> __device__ __forceinline__ void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Byte offset of 16-byte piece c of pixel (y, x) inside a tile image with `pitch` pixel slots per row and RB bytes
// per pixel.  An involution on (x, c) for fixed y, so the same function maps "slot -> what to fetch" for the DMA.
template <int RB, int TWL>
__device__ __forceinline__ int wg_tile_off(int y, int x, int pitch, int c) {
  const int sb = TWL == 4 ? ((x >> 3) & 1) : (y & 1);
  if constexpr (RB == 128) {
    const int s2 = ((x >> 1) & 1) | (sb << 1);
    return ((y * pitch + x) * 8 + (c ^ (2 * s2))) * 16;
  } else if constexpr (RB == 64) {
    return ((y * pitch + x) * 4 + (c ^ (2 * sb))) * 16;
  } else {
    return ((y * pitch + (x ^ (sb << 2))) * 2 + c) * 16;
  }
}

// One MFMA operand = two ds_read_b64_tr_b16 results (k = 8g .. 8g+3 and 8g+4 .. 8g+7 of one channel per lane).
struct WgFrag {
  s16x4 lo, hi;
  __device__ __forceinline__ bf16x8 v() const {
    union { struct { s16x4 l, h; } s; bf16x8 v; } u;
    u.s.l = lo; u.s.h = hi;
    return u.v;
  }
};
template <int OFF>
__device__ __forceinline__ void wg_issue_tr(WgFrag& f, unsigned a0, unsigned a1) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(a0), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(a1), "n"(OFF));
}
// placed after the s_waitcnt that covers the reads: ties the consumers of f behind that wait
__device__ __forceinline__ void wg_landed(WgFrag& f) {
  asm volatile("" : "+v"(f.lo));
  asm volatile("" : "+v"(f.hi));
}
template <int N, typename F, int I = 0>
__device__ __forceinline__ void wg_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    wg_static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

struct WgTilePos {
  int b, ty, tx;
  __device__ __forceinline__ void set(int t, int tiles_x, int tiles_y) {
    tx = t % tiles_x; t /= tiles_x;
    ty = t % tiles_y; b = t / tiles_y;
  }
  __device__ __forceinline__ void next(int tiles_x, int tiles_y) {
    if (++tx == tiles_x) { tx = 0; if (++ty == tiles_y) { ty = 0; ++b; } }
  }
};

// TAPS: 9 (3x3, pad 1) or 1.  TGH x TGW: the rectangle of taps one block owns.  MI / NBF: 16-channel fragments of
// the block's a (P) / b (Q) range.  TWL: log2 of the tile width (16x8 or 8x16 pixels).
static constexpr bool DBG = false;  // SX_WG_DBG ablation switches (bit 0: no MFMA loop, bit 1: no operand traffic)

template <int TAPS, int TGH, int TGW, int MI, int NBF, int TWL, int NBUF, bool DUAL = false>
__device__ __forceinline__ void wgrad_ws_body(const WgWsArgs& a, const int block_id) {
  constexpr int PAD = TAPS == 9 ? 1 : 0;
  constexpr bool CT = TAPS == 4;                       // transposed conv: four parity sub-images of Q, no halo
  static_assert(!CT || (TGH == 1 && TGW == 1), "transposed conv: taps are sub-images, not shifts");
  constexpr int TW = 1 << TWL, TH = 128 / TW, YS = 32 / TW;
  constexpr int LW = TW + TGW - 1, LH = TH + TGH - 1;
  constexpr int PB = MI * 32, QB = NBF * 32;
  constexpr int PP = PB / 16, QP = QB / 16;
  constexpr int QPITCH = NBF == 1 ? ((LW + 7) & ~7) : LW;
  constexpr int NPP = (128 * PP) / 256;
  constexpr int QSUB = LH * QPITCH * QB;                // bytes of one Q (sub-)image
  constexpr int QSLOTS = (CT ? 4 : 1) * LH * QPITCH * QP;
  constexpr int NPQ = (QSLOTS + 255) / 256;
  constexpr int NPT = NPP + NPQ;
  constexpr int BUF = NPT * 4096;
  constexpr int DIST = NBUF - 1;
  constexpr int NIT = (CT ? 4 : TGH * TGW) * NBF, ITEMS = (NIT + 3) / 4;
  static_assert(NPP >= 1, "P tile smaller than one DMA round");
  extern __shared__ __align__(16) char smem[];
  const unsigned long long t_entry = DBG ? __builtin_amdgcn_s_memtime() : 0;

  // waves 0-3 compute (one per SIMD), waves 4-7 only feed the LDS ring: a wave that issues an LDS-DMA instruction is
  // held for ~50 cycles per KB, which would otherwise come straight out of the MFMA stream of a one-wave SIMD
  const int wid8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = wid8 >= 4;
  const int wid = wid8 & 3, tid = threadIdx.x & 255, lane = tid & 63;
  const int n = lane & 15, g = lane >> 4, q4 = n >> 2, p4 = n & 3;

  // ---- which part of the problem: partners (same pixel range, different r) are 8 block ids apart = same XCD
  const int id = block_id, xcd = id & 7, wq = id >> 3;
  const int split = (wq / a.R) * 8 + xcd;
  int r = wq % a.R;
  if (split >= a.S) return;
  const int tg = r % a.tg_count; r /= a.tg_count;
  const int tb = r % a.b_tiles, ta = r / a.b_tiles;
  const int a0 = ta * (MI * 16), b0 = tb * (NBF * 16);
  const int oy = TGH == 3 ? 0 : (TGW == 3 ? tg : tg / 3);
  const int ox = TGW == 3 ? 0 : (TAPS == 9 ? tg % 3 : 0);

  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
  const int nst = t_end - t_begin;

  // ---- DMA setup: slot e = tid + q*256 of the P / Q image -> (pixel, piece) it holds -> global offset
  const int prow_bytes = a.W * a.p_ld * 2, qrow_bytes = a.Wq * a.q_ld * 2;
  const int pimg_bytes = a.H * prow_bytes, qimg_bytes = a.Hq * qrow_bytes;
  int p_rel[NPP], p_x[NPP], q_rel[NPQ], q_x[NPQ];
#pragma unroll
  for (int q = 0; q < NPP; ++q) {
    const int e = tid + q * 256;
    const int ps = e / PP, cs = e % PP;
    const int y = ps >> TWL, xs = ps & (TW - 1);
    // invert the swizzle: the involution maps the slot's (xs, cs) to the (x, c) stored there
    const int o = wg_tile_off<PB, TWL>(y, xs, TW, cs) >> 4;
    const int x = (o / PP) & (TW - 1), c = o % PP;
    const bool ok = a0 + c * 8 < a.Ca;
    p_x[q] = ok ? x : -100000;
    p_rel[q] = y * prow_bytes + x * a.p_ld * 2 + (a0 + c * 8) * 2;
  }
#pragma unroll
  for (int q = 0; q < NPQ; ++q) {
    const int e = tid + q * 256;
    const int sub = CT ? e / (LH * QPITCH * QP) : 0;          // parity class = tap (ty, tx) of the transposed conv
    const int es = e - sub * (LH * QPITCH * QP);
    const int ps = es / QP, cs = es % QP;
    const int y = ps / QPITCH, xs = ps - y * QPITCH;
    const int o = wg_tile_off<QB, TWL>(y, xs, QPITCH, cs) >> 4;
    const int x = (o / QP) - y * QPITCH, c = o % QP;
    const bool ok = sub < 4 && y < LH && x < LW && b0 + c * 8 < a.Cb;
    if constexpr (CT) {
      q_x[q] = ok ? x : -100000;                               // bounds are checked on the P (low-resolution) grid
      q_rel[q] = (2 * y + (sub >> 1)) * qrow_bytes + (2 * x + (sub & 1)) * a.q_ld * 2 + (b0 + c * 8) * 2;
    } else {
      q_x[q] = ok ? x + ox - PAD : -100000;
      q_rel[q] = (y + oy - PAD) * qrow_bytes + (x + ox - PAD) * a.q_ld * 2 + (b0 + c * 8) * 2;
    }
  }

  // A stage's NPT pieces are issued one by one (stage_begin, issue_piece<q> ..., stage_end) so the main loop can
  // spread them between its k-steps: an LDS-DMA instruction holds the issuing wave for ~50 cycles of the CU's one
  // texture-address unit, which other waves' MFMAs cover only if the four waves do not all issue at once.
  WgTilePos ip;
  ip.set(t_begin, a.tiles_x, a.tiles_y);
  int is = 0;
  __amdgpu_buffer_rsrc_t rp, rq;
  int ix0 = 0, ipoff = 0, iqoff = 0;
  char* ibuf = smem;
  auto stage_begin = [&]() {
    ibuf = smem + (is % NBUF) * BUF + wid * 1024;
    const bool live = is < nst && !(DBG && (a.dbg & 2));
    rp = __builtin_amdgcn_make_buffer_rsrc((void*)(a.P + (int64_t)ip.b * a.H * a.W * a.p_ld), 0, live ? pimg_bytes : 0, 0x00020000);
    rq = __builtin_amdgcn_make_buffer_rsrc((void*)(a.Q + (int64_t)ip.b * a.Hq * a.Wq * a.q_ld), 0, live ? qimg_bytes : 0, 0x00020000);
    ix0 = ip.tx * TW;
    const int y0 = ip.ty * TH;
    ipoff = y0 * prow_bytes + ix0 * a.p_ld * 2;
    iqoff = (CT ? 2 : 1) * (y0 * qrow_bytes + ix0 * a.q_ld * 2);
  };
  auto issue_piece = [&](auto QI) {
    constexpr int q = decltype(QI)::value;
    if constexpr (q < NPP) {
      const unsigned off = (unsigned)(ix0 + p_x[q]) < (unsigned)a.W ? (unsigned)(p_rel[q] + ipoff) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (__attribute__((address_space(3))) void*)(ibuf + q * 4096), 16, off, 0, 0, 0);
    } else {
      const unsigned off = (unsigned)(ix0 + q_x[q - NPP]) < (unsigned)a.W ? (unsigned)(q_rel[q - NPP] + iqoff) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void*)(ibuf + q * 4096), 16, off, 0, 0, 0);
    }
  };
  auto stage_end = [&]() {
    ++is;
    ip.next(a.tiles_x, a.tiles_y);
  };
  auto issue_stage = [&]() {
    stage_begin();
    wg_static_for<NPT>(issue_piece);
    stage_end();
  };

  // ---- fragment addressing (stage-relative byte offsets, invariant over tiles)
  const int yl = TWL == 4 ? (g >> 1) : g;
  const int xl = (TWL == 4 ? 8 * (g & 1) : 0) + q4;
  const int hh = p4 >> 1, lo = (p4 & 1) * 8;
  int offP[MI][2];
#pragma unroll
  for (int j = 0; j < MI; ++j)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) offP[j][rd] = wg_tile_off<PB, TWL>(yl, xl + 4 * rd, TW, 2 * j + hh) + lo;
  int offQ[ITEMS][2], it_tap[ITEMS], it_bf[ITEMS];
  bool it_valid[ITEMS];
#pragma unroll
  for (int t = 0; t < ITEMS; ++t) {
    // DUAL: "is this the centre tap" must be known at compile time.  NBF == 4: item t of every wave is tap t (bf = wave).
    // NBF == 1: items are dealt t-major and the taps rotated by 4, so the centre tap is item 0 of wave 0 (waves 1-3 then
    // multiply the shortcut rows for taps 5-7 as well: rows nobody reads).
    const int it = DUAL ? t * 4 + wid : wid * ITEMS + t;
    it_valid[t] = it < NIT;
    const int itc = it_valid[t] ? it : 0;
    const int tl = (DUAL && NBF == 1) ? (itc + 4) % 9 : itc / NBF, bf = (DUAL && NBF == 1) ? 0 : itc - tl * NBF;
    const int dy = CT ? 0 : tl / TGW, dx = CT ? 0 : tl - dy * TGW;
    it_tap[t] = TAPS == 9 ? (oy + dy) * 3 + (ox + dx) : (CT ? tl : 0);
    it_bf[t] = bf;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd)
      offQ[t][rd] = NPP * 4096 + (CT ? tl * QSUB : 0) + wg_tile_off<QB, TWL>(yl + dy, xl + dx + 4 * rd, QPITCH, 2 * bf + hh) + lo;
  }

  f32x4 acc[ITEMS][MI];
#pragma unroll
  for (int t = 0; t < ITEMS; ++t)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (loader) {
#pragma unroll
    for (int k = 0; k < DIST; ++k) issue_stage();
    for (int s = 0; s < nst; ++s) {
      wg_wait_vmcnt<(DIST - 1) * NPT>();                 // this wave's pieces of stage s have landed
      __builtin_amdgcn_s_barrier();                      // ... everyone's have, and buffer (s-1) % NBUF is free
      issue_stage();                                     // stage s + DIST (dummy past the end: keeps vmcnt uniform)
    }
    wg_wait_vmcnt<0>();   // drain the dummy prefetches before the wave (and the block's LDS allocation) retires
    return;
  }

  const unsigned sm0 = (unsigned)(uintptr_t)smem;
  const unsigned long long t_setup = DBG ? __builtin_amdgcn_s_memtime() : 0;
  unsigned long long t_first = 0;
  for (int s = 0; s < nst; ++s) {
    __builtin_amdgcn_s_barrier();                        // stage s is in LDS (the loaders waited for it)
    if (DBG && s == 0) t_first = __builtin_amdgcn_s_memtime();
    const unsigned buf = sm0 + (s % NBUF) * BUF;
    if (DBG && (a.dbg & 1)) continue;
    unsigned aP[MI][2], aQ[ITEMS][2];
#pragma unroll
    for (int j = 0; j < MI; ++j) { aP[j][0] = buf + offP[j][0]; aP[j][1] = buf + offP[j][1]; }
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) { aQ[t][0] = buf + offQ[t][0]; aQ[t][1] = buf + offQ[t][1]; }
    // Transposed reads are issued from inline asm (the compiler would put s_waitcnt vmcnt(0) in front of any LDS
    // read it can see while LDS-DMA is in flight) one k-step ahead of the MFMAs that consume them; every k-step
    // starts with one lgkmcnt(0) for reads issued a whole k-step of MFMAs earlier.
    WgFrag Af[2][MI], Bq[2][ITEMS];
#pragma unroll
    for (int j = 0; j < MI; ++j) wg_issue_tr<0>(Af[0][j], aP[j][0], aP[j][1]);
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) wg_issue_tr<0>(Bq[0][t], aQ[t][0], aQ[t][1]);
    wg_static_for<4>([&](auto KS) {
      constexpr int ks = decltype(KS)::value, cur = ks & 1, nxt = cur ^ 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < MI; ++j) wg_landed(Af[cur][j]);
#pragma unroll
      for (int t = 0; t < ITEMS; ++t) wg_landed(Bq[cur][t]);
      if constexpr (ks < 3) {
#pragma unroll
        for (int j = 0; j < MI; ++j) wg_issue_tr<(ks + 1) * 32 * PB>(Af[nxt][j], aP[j][0], aP[j][1]);
      }
#pragma unroll
      for (int t = 0; t < ITEMS; ++t) {
        if constexpr (ks < 3) wg_issue_tr<(ks + 1) * YS * QPITCH * QB>(Bq[nxt][t], aQ[t][0], aQ[t][1]);
#pragma unroll
        for (int j = 0; j < MI; ++j)
          if (!DUAL || j < MI / 2 || t == (NBF == 4 ? 4 : 0))   // DUAL: fragments 2, 3 of P are the shortcut's dR -- centre tap only (static)
            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bq[cur][t].v(), Af[cur][j].v(), acc[t][j], 0, 0, 0);
      }
    });
  }

  const unsigned long long t_loop = DBG ? __builtin_amdgcn_s_memtime() : 0;
  // ---- slab[split][tap][a/16][b/16][a%16][b%16]: lane (n, g) of tile (t, j) holds b = b0 + 16*bf + 4g .. +3 of row
  // a = a0 + 16*j + n, so one store instruction of a wave fills one contiguous 1 KB tile (row-major slabs made it sixteen
  // 64-byte pieces and the store tail ran at 15 B/cycle/CU: 9.6k cycles of a 44k-cycle launch)
  const int ntaps = TAPS;
  float* slab = a.slab + (int64_t)split * ntaps * a.CaPad * a.CbPad;
  const int B16 = a.CbPad >> 4, A16 = a.CaPad >> 4;
#pragma unroll
  for (int t = 0; t < ITEMS; ++t) {
    if (!it_valid[t]) continue;
    const int b16 = (b0 >> 4) + it_bf[t];
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int a16 = (a0 >> 4) + j;
      if (a16 < A16 && b16 < B16)
        *reinterpret_cast<f32x4*>(slab + ((((int64_t)it_tap[t] * A16 + a16) * B16 + b16) << 8) + n * 16 + 4 * g) = acc[t][j];
    }
  }
  if (DBG && a.stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = a.stamps + (size_t)block_id * 8;
    o[0] = t_entry; o[1] = t_setup; o[2] = t_first; o[3] = t_loop; o[4] = __builtin_amdgcn_s_memtime();
    o[5] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int TAPS, int TGH, int TGW, int MI, int NBF, int TWL, int NBUF, bool DUAL = false>
__global__ __launch_bounds__(512, 1) void wgrad_ws_kernel(WgWsArgs a) {
  wgrad_ws_body<TAPS, TGH, TGW, MI, NBF, TWL, NBUF, DUAL>(a, (int)blockIdx.x);
}

// ---- every weight gradient of a backward half in ONE launch.  The layers' split-K problems are independent of each other
// and of the data-gradient chain, and each of them alone is a 7-45 us launch of which ~8 us are ramp, first touch and slab
// store tail: queued behind one another in a single grid (block ranges per problem, every range a multiple of 8 blocks so
// that "block id & 7 = XCD" holds inside each), the tail of one problem runs under the head of the next and the
// per-dispatch cost is paid once.  A block looks up its problem and jumps to that problem's instantiation of the body.
template <int TAPS, int TGH, int TGW, int MI, int NBF, int TWL>
struct WgGeom {
  static constexpr int TW = 1 << TWL, TH = 128 / TW;
  static constexpr int LW = TW + TGW - 1, LH = TH + TGH - 1;
  static constexpr int QPITCH = NBF == 1 ? ((LW + 7) & ~7) : LW;
  static constexpr int NPT = (128 * MI * 2) / 256 + ((TAPS == 4 ? 4 : 1) * LH * QPITCH * NBF * 2 + 255) / 256;
  static constexpr int NBUF = NPT * 4096 * 4 <= 160 * 1024 ? 4 : 3;
  static constexpr int SMEM = NBUF * NPT * 4096;
};

// variant code = DUAL << 12 | TAPS << 8 | MI << 5 | NBF << 2 | (TWL - 3)   (all-taps blocks only: TGH x TGW = the whole stencil)
#define SX_WG_VARIANTS(X)                                                                                     \
  X(9, 3, 3, 1, 1, 4, false) X(9, 3, 3, 1, 2, 4, false) X(9, 3, 3, 1, 4, 4, false)                             \
  X(9, 3, 3, 2, 1, 4, false) X(9, 3, 3, 2, 2, 4, false) X(9, 3, 3, 2, 4, 4, false)                             \
  X(9, 3, 3, 4, 1, 4, false) X(9, 3, 3, 4, 2, 4, false) X(9, 3, 3, 4, 4, 4, false)                             \
  X(9, 3, 3, 4, 4, 3, false) X(9, 3, 3, 4, 2, 3, false) X(9, 3, 3, 2, 4, 3, false) X(9, 3, 3, 2, 2, 3, false) \
  X(1, 1, 1, 1, 1, 4, false) X(1, 1, 1, 1, 2, 4, false) X(1, 1, 1, 1, 4, 4, false)                             \
  X(1, 1, 1, 2, 1, 4, false) X(1, 1, 1, 2, 2, 4, false) X(1, 1, 1, 2, 4, 4, false)                             \
  X(1, 1, 1, 4, 1, 4, false) X(1, 1, 1, 4, 2, 4, false) X(1, 1, 1, 4, 4, 4, false)                             \
  X(1, 1, 1, 4, 4, 3, false) X(1, 1, 1, 4, 2, 3, false) X(1, 1, 1, 2, 4, 3, false) X(1, 1, 1, 2, 2, 3, false) \
  X(4, 1, 1, 4, 2, 4, false) X(4, 1, 1, 4, 2, 3, false)                                                       \
  X(9, 3, 3, 4, 4, 4, true) X(9, 3, 3, 4, 1, 4, true)

static constexpr int wg_variant_code(int taps, int mi, int nbf, int twl, bool dual) {
  return (dual ? 1 << 12 : 0) | taps << 8 | mi << 5 | nbf << 2 | (twl - 3);
}

static constexpr int WG_MULTI_MAX = 16;
struct WgWsMultiArgs {
  WgWsArgs p[WG_MULTI_MAX];
  int variant[WG_MULTI_MAX];
  int blk0[WG_MULTI_MAX + 1];
  int n;
};

__global__ __launch_bounds__(512, 1) void wgrad_ws_multi_kernel(const WgWsMultiArgs) {
  // The table is read straight from the kernel-argument segment (scalar loads at a uniform dynamic offset): indexing the
  // by-value parameter itself made the compiler copy all 2 KB of it to scratch, per lane.
  typedef const WgWsMultiArgs __attribute__((address_space(4))) * TablePtr;
  TablePtr A = (TablePtr)__builtin_amdgcn_kernarg_segment_ptr();
  int pi = 0;
  const int n = A->n;
  while (pi + 1 < n && (int)blockIdx.x >= A->blk0[pi + 1]) ++pi;
  const int id = (int)blockIdx.x - A->blk0[pi];
  WgWsArgs a;
  __builtin_memcpy(&a, &A->p[pi], sizeof(a));
  switch (A->variant[pi]) {
#define SX_WG_BODY(taps, tgh, tgw, mi, nbf, twl, dual)                                                             \
    case wg_variant_code(taps, mi, nbf, twl, dual):                                                                 \
      wgrad_ws_body<taps, tgh, tgw, mi, nbf, twl, WgGeom<taps, tgh, tgw, mi, nbf, twl>::NBUF, dual>(a, id);         \
      break;
    SX_WG_VARIANTS(SX_WG_BODY)
#undef SX_WG_BODY
    default: break;
  }
}

// ------------------------------------------------------------------ host side
struct WgWsPlan {
  bool ok;
  int taps;
  int MI, NBF, TWL;
  int a_tiles, b_tiles, tg_count, R, S;
  int tiles_x, tiles_y, ntiles, tiles_per_split;
  int CaPad, CbPad;
  int64_t slab_bytes;
};

// tuning knobs, read once per process
static int wg_min_tiles(bool ct) { static const int v = env_int("SX_WG_MIN_TILES", 0); return v > 0 ? v : (ct ? 4 : 8); }
// blocks one problem aims for (S * R).  Problems run many to a launch (wgrad_ws_launch), so none of them has to fill the
// 256 CUs alone: fewer, longer splits mean fewer slabs to write and to reduce
// (measured on cfg A, batch 64: 256 blocks per problem 1.906 ms/step, 128: 1.846, 64: 1.807, 32: 1.820, 16: 1.97); a problem
// launched on its own (sx_conv_wgrad) still spreads over the whole chip
// Large problems (cfg 5: 34-155 GFLOP per layer against <= 19 on cfg A) get blocks in proportion to their arithmetic instead:
// at 64 blocks each the 13 problems of a half left the chip to a few 0.5 ms blocks (decoder half 1.45 ms at 22 % of the
// MFMA peak; cfg 5 step 7.28 -> 6.70 ms with 100 MFLOP per block, 6.89 with 200, 6.95 with 400).  cfg A wants the opposite
// (a block's fixed cost of ~8 us against 10-20 GFLOP problems: 1.68 ms per step at 400 MFLOP per block = 64 blocks each,
// 1.69 at 200, 1.72 at 100), so the finer split starts at 45 GFLOP (cfg A's largest problem, the fused conv1 + shortcut of
// the last decoder, counts 38.6 with its centre-tap-only rows).  SX_WG_BLOCK_MFLOP overrides both; <= 512 blocks per
// problem (the slab reservation, wgrad_ws_slab_bytes, covers whichever plan is larger).
static constexpr int WG_MAX_BLOCKS = 512;
static int wg_target_blocks(bool batched, double mflop) {
  static const int v = std::min(std::max(env_int("SX_WG_TARGET", 0), 0), 256);
  static const int per_block_env = env_int("SX_WG_BLOCK_MFLOP", 0);
  if (!batched) return 256;
  static const int fine_gf = env_int("SX_WG_FINE_GF", 45);
  const int per_block = per_block_env > 0 ? per_block_env : (mflop > 1e3 * fine_gf ? 100 : 400);
  const int by_cost = (int)std::min<double>(WG_MAX_BLOCKS, mflop / per_block + 1.0);
  return std::max(v > 0 ? v : 64, by_cost);
}

static WgWsPlan wgws_plan(int mode, int B, int H, int W, int Ca, int Cb, bool batched = false) {
  WgWsPlan p;
  memset(&p, 0, sizeof(p));
  if (mode != SX_CONV3 && mode != SX_CONV1 && mode != SX_CONVT_FWD) return p;
  if (W < 8 || Ca % 8 || Cb % 8) return p;
  const bool ct = mode == SX_CONVT_FWD;
  if ((int64_t)H * W * std::max(Ca, Cb) * 2 >= ((int64_t)1 << 30)) return p;   // 32-bit offsets inside one image
  p.taps = mode == SX_CONV3 ? 9 : (ct ? 4 : 1);
  p.CaPad = round_up(Ca, 16); p.CbPad = round_up(Cb, 16);
  p.MI = p.CaPad >= 64 ? 4 : p.CaPad / 16;      // 1, 2, (3 -> 4), 4
  if (p.MI == 3) p.MI = 4;
  p.NBF = p.CbPad >= 64 ? 4 : p.CbPad / 16;
  if (p.NBF == 3) p.NBF = 4;
  if (ct) {                                      // four Q sub-images per stage: 32 b channels keep three stages in LDS
    if (p.CaPad < 64 || p.CbPad < 32) { return p; }
    p.MI = 4; p.NBF = 2;
  }
  p.a_tiles = cdiv(p.CaPad, p.MI * 16);
  p.b_tiles = cdiv(p.CbPad, p.NBF * 16);
  p.TWL = W >= 16 ? 4 : 3;
  const int TW = 1 << p.TWL, TH = 128 / TW;
  p.tiles_x = cdiv(W, TW); p.tiles_y = cdiv(H, TH);
  p.ntiles = B * p.tiles_x * p.tiles_y;
  // every block owns ALL taps of its (a tile, b tile) (splitting the taps over blocks was measured and never paid on cfg A).
  // At least MIN_TILES tiles per block: slab traffic (S slabs written, then read by the reduction) is what small-K
  // layers pay for parallelism, and a block's fixed cost (~8 us) wants amortising anyway
  p.tg_count = 1;
  p.R = p.a_tiles * p.b_tiles;
  int S = std::max(1, std::min(std::max(1, p.ntiles / wg_min_tiles(ct)), std::max(1, wg_target_blocks(batched, 2e-6 * p.ntiles * 128.0 * p.taps * p.CaPad * p.CbPad) / p.R)));
  // partners of one split share an XCD (block id & 7), so a plan with S < 8 leaves XCDs without a block of this problem and
  // queues its R partners on the few that have one (cfg 5, transposed conv 512 -> 256: R = 64, S = 1: 377 us on one XCD;
  // with >= 8 splits cfg 5's decoder half 1263 -> 810 us, step 7.00 -> 6.54 ms; cfg A 1.715 -> 1.704 ms, same box)
  static const int min_splits = env_int("SX_WG_MIN_SPLITS", 8);
  if (S < min_splits) S = std::max(S, std::min(min_splits, std::max(1, p.ntiles / wg_min_tiles(ct))));
  p.tiles_per_split = cdiv(p.ntiles, S);
  p.S = cdiv(p.ntiles, p.tiles_per_split);
  p.slab_bytes = (int64_t)p.S * p.taps * p.CaPad * p.CbPad * 4;
  p.ok = true;
  return p;
}

// per-variant launch facts (dynamic LDS) and the one-time function attribute, keyed by the variant code
static int wg_variant_smem(int code) {
  switch (code) {
#define SX_WG_SMEM(taps, tgh, tgw, mi, nbf, twl, dual) \
    case wg_variant_code(taps, mi, nbf, twl, dual): return WgGeom<taps, tgh, tgw, mi, nbf, twl>::SMEM;
    SX_WG_VARIANTS(SX_WG_SMEM)
#undef SX_WG_SMEM
  }
  return -1;
}

static int launch_single(const WgWsLaunch& l, hipStream_t st) {
  switch (l.variant) {
#define SX_WG_LAUNCH(taps, tgh, tgw, mi, nbf, twl, dual)                                                           \
    case wg_variant_code(taps, mi, nbf, twl, dual): {                                                                \
      auto kern = wgrad_ws_kernel<taps, tgh, tgw, mi, nbf, twl, WgGeom<taps, tgh, tgw, mi, nbf, twl>::NBUF, dual>;   \
      static std::atomic<unsigned long long> done{0};                                                                \
      SX_TRY(ensure_max_lds(reinterpret_cast<const void*>(kern), done, "hipFuncSetAttribute(wgrad_ws)"));            \
      hipLaunchKernelGGL(kern, dim3(l.blocks), dim3(512), (size_t)l.smem, st, l.a);                                   \
      break;                                                                                                         \
    }
    SX_WG_VARIANTS(SX_WG_LAUNCH)
#undef SX_WG_LAUNCH
    default: set_error("wgrad_ws: unknown variant %d", l.variant); return SX_ERR_INVALID;
  }
  SX_LAUNCH_CHECK("wgrad_ws");
  return SX_OK;
}

int64_t wgrad_ws_slab_bytes(int mode, int B, int H, int W, int Ca, int Cb) {
  const WgWsPlan p = wgws_plan(mode, B, H, W, Ca, Cb), q = wgws_plan(mode, B, H, W, Ca, Cb, true);
  return p.ok ? std::max(p.slab_bytes, q.slab_bytes) : 0;
}

// 0 = prepared (launch with wgrad_ws_launch), 1 = shape not covered
int wgrad_ws_prepare(int mode, int B, int H, int W, int Ca, int Cb, const void* P, int p_ld, const void* Q, int q_ld, int Hq, int Wq,
                     float* slab, WgReduceDesc* desc, WgWsLaunch* out, int dual, int batched) {
  const WgWsPlan p = wgws_plan(mode, B, H, W, Ca, Cb, batched != 0);
  if (!p.ok) return 1;
  // callers reserve wgrad_ws_slab_bytes() per problem: whatever plan runs must fit it (knobs included)
  if (p.slab_bytes > wgrad_ws_slab_bytes(mode, B, H, W, Ca, Cb)) { set_error("wgrad_ws: plan needs %lld slab bytes, more than reserved", (long long)p.slab_bytes); return SX_ERR_INVALID; }
  if (dual && (mode != SX_CONV3 || p.MI != 4 || (p.NBF != 4 && p.NBF != 1) || p.a_tiles != 1 || p.b_tiles != 1 || p.TWL != 4 || Ca != 64)) return 1;
  WgWsArgs& a = out->a;
  memset(&a, 0, sizeof(a));
  a.P = (const bf16*)P; a.Q = (const bf16*)Q; a.slab = slab;
  a.B = B; a.H = H; a.W = W; a.p_ld = p_ld; a.q_ld = q_ld; a.Hq = Hq; a.Wq = Wq;
  a.Ca = Ca; a.Cb = Cb; a.CaPad = p.CaPad; a.CbPad = p.CbPad;
  a.a_tiles = p.a_tiles; a.b_tiles = p.b_tiles; a.tg_count = p.tg_count; a.R = p.R; a.S = p.S;
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.ntiles = p.ntiles; a.tiles_per_split = p.tiles_per_split;
  out->variant = wg_variant_code(p.taps, p.MI, p.NBF, p.TWL, dual != 0);
  out->smem = wg_variant_smem(out->variant);
  if (out->smem < 0) return 1;
  out->blocks = cdiv(p.S, 8) * 8 * p.R;
  desc->slab = slab; desc->nsplit = p.S; desc->ntaps = p.taps; desc->CaPad = p.CaPad; desc->CbPad = p.CbPad; desc->tiled = 1;
  return SX_OK;
}

// One launch for up to WG_MULTI_MAX prepared problems (more are split over several launches); a single problem uses its
// own instantiation.  Problems run in the order given: put the long ones first.
int wgrad_ws_launch(const WgWsLaunch* l, int n, hipStream_t st) {
  static const int no_multi = env_int("SX_WG_NO_MULTI", 0);        // A/B knob: one launch per problem
  for (int i0 = 0; i0 < n;) {
    const int m = no_multi ? 1 : std::min(WG_MULTI_MAX, n - i0);
    if (m == 1) { SX_TRY(launch_single(l[i0], st)); ++i0; continue; }
    WgWsMultiArgs A;
    memset(&A, 0, sizeof(A));
    A.n = m;
    int blk = 0, smem = 0;
    for (int i = 0; i < m; ++i) {
      A.p[i] = l[i0 + i].a; A.variant[i] = l[i0 + i].variant; A.blk0[i] = blk;
      blk += l[i0 + i].blocks;                 // a multiple of 8: block id & 7 stays the XCD inside every problem
      smem = std::max(smem, l[i0 + i].smem);
    }
    A.blk0[m] = blk;
    static std::atomic<unsigned long long> done{0};
    SX_TRY(ensure_max_lds(reinterpret_cast<const void*>(wgrad_ws_multi_kernel), done, "hipFuncSetAttribute(wgrad_ws_multi)"));
    hipLaunchKernelGGL(wgrad_ws_multi_kernel, dim3(blk), dim3(512), (size_t)smem, st, A);
    SX_LAUNCH_CHECK("wgrad_ws_multi");
    i0 += m;
  }
  return SX_OK;
}

int wgrad_ws_try(int mode, int B, int H, int W, int Ca, int Cb, const void* P, int p_ld, const void* Q, int q_ld, int Hq, int Wq,
                 float* slab, WgReduceDesc* desc, hipStream_t st, int dual) {
  WgWsLaunch l;
  const int rc = wgrad_ws_prepare(mode, B, H, W, Ca, Cb, P, p_ld, Q, q_ld, Hq, Wq, slab, desc, &l, dual, 0);
  if (rc != 0) return rc;
  return wgrad_ws_launch(&l, 1, st);
}

}  // namespace sx
