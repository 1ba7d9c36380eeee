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

#ifndef SX_WS_NBUF
#define SX_WS_NBUF 4
#endif
#ifndef SX_WS_PFD
#define SX_WS_PFD 1
#endif
#ifndef SX_WS_DBG
#define SX_WS_DBG 0        // 1: in-kernel s_memtime stamps (tools/conv_stamps.py); never in the shipped library
#endif

namespace sx {

static constexpr bool WS_DBG = SX_WS_DBG != 0;
#ifndef SX_WS_ABL
#define SX_WS_ABL 0        // diagnostic builds only (wrong results): bit 0 no epilogue stores, bit 1 no statistics arithmetic
#endif

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// four bf16 outputs as exactly ONE vector-memory instruction (the stage waits below count the stores of full tiles)
__device__ __forceinline__ void store4_counted(bf16* p, f32x4 v) {
  const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  union { bf16x4 b; u32x2 u; } c;
  c.b = o;
  asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(c.u) : "memory");
}
// s_waitcnt vmcnt(BASE + c * STEP), c in 0..3 (wave-uniform)
template <int BASE, int STEP> __device__ __forceinline__ void wait_vmcnt_plus(int c) {
  static_assert(BASE + 3 * STEP <= 63, "vmcnt field");
  if (c == 0) wait_vmcnt<BASE>();
  else if (c == 1) wait_vmcnt<BASE + STEP>();
  else if (c == 2) wait_vmcnt<BASE + 2 * STEP>();
  else wait_vmcnt<BASE + 3 * STEP>();
}

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4); every lane ends with the total
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x128, 0xf, 0xf, false);   // row_ror:8
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x124, 0xf, 0xf, false);   // row_ror:4
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x122, 0xf, 0xf, false);   // row_ror:2
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x121, 0xf, 0xf, false);   // row_ror:1
  return v;
}

// position of a tile inside the batch, advanced incrementally (no divisions in the loops)
struct TilePos {
  int b, ty, tx;
  __device__ __forceinline__ void set(int t, int tiles_x, int tiles_y) {
    tx = t % tiles_x; t /= tiles_x;
    ty = t % tiles_y; b = t / tiles_y;
  }
  __device__ __forceinline__ void next(int tiles_x, int tiles_y) {
    if (++tx == tiles_x) { tx = 0; if (++ty == tiles_y) { ty = 0; ++b; } }
  }
};

// TWL: log2 of the tile width.  16-wide tiles: one MFMA pixel fragment = one tile row.  8-wide tiles (grids narrower
// than 16, the deepest U-Net levels): a fragment = two rows of 8.
// VAR: WS_PLAIN / WS_DUAL / WS_CENTER (conv_common.h): the 1x1 shortcut of a DoubleConv block rides along with the 3x3.
template <int KCH, int TAPS, int MI, int NI, int WM, int WN, int NBUF, int TWL, int VAR>
__global__ __launch_bounds__(256, NBUF == 3 ? 2 : 1) void conv_ws_kernel(ConvArgs a) {     // 3-stage ring = the two-blocks-per-CU form
  constexpr int TAPW = TAPS == 9 ? 3 : 1;
  constexpr int PAD = TAPS == 9 ? 1 : 0;
  constexpr int BMC = WM * MI * 16;
  constexpr int TW = 1 << TWL, RPF = 16 / TW;      // rows per pixel fragment
  constexpr int TH = WN * NI * RPF;
  constexpr int LW = TW + TAPW - 1, LH = TH + TAPW - 1;
  constexpr int NPIX = LH * LW, PIECES = NPIX * 4, NP = (PIECES + 255) / 256;
  constexpr int BUF = NP * 256 * 16;            // lane-linear image: piece p (pixel p>>2, part p&3) at byte 16*p
  constexpr int DIST = NBUF - 1;                // stages in flight
  constexpr int KTOT = VAR == WS_CENTER ? 2 * KCH : KCH;   // K chunks (= pipeline stages) per tile
  constexpr int CT = TAPS / 2;                  // centre tap
  static_assert(VAR == WS_PLAIN || VAR == WS_BS || VAR == WS_GATHER || TAPS == 9, "fused variants are 3x3 only");
  static_assert(VAR != WS_GATHER || (TAPS == 1 && KCH % 4 == 0), "the gather variant is a 1x1 over 4 taps x K chunks");
  constexpr bool GA = VAR == WS_GATHER;
  constexpr bool BS = VAR == WS_BS;
  constexpr int PFD = SX_WS_PFD;                // fragment rows in flight ahead of the MFMAs
  constexpr int SWZ_SH = (TWL == 3 && TAPS == 9) ? 1 : 2;     // halo-column bit that flips the channel-group pair (bank swizzle)
  // (block-output sums -- four per channel, FiLM / dropout factors, for the 1x1 head's data gradient -- were built and
  // measured: 45 us fused against 17 + 26 us for the two launches.  A 1x1 launch is all epilogue, and at one wave per SIMD
  // nothing hides the sigmoid's latency; the stand-alone reduction runs the same arithmetic at 8 waves per SIMD.)
  constexpr bool BS2 = false;
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __align__(16) char smem[];
  float* sm_stat = reinterpret_cast<float*>(smem + NBUF * BUF);   // [WN][BMC][2]

  const unsigned long long t_entry = WS_DBG ? __builtin_amdgcn_s_memtime() : 0;
  unsigned long long t_first = 0;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int n = lane & 15, g = lane >> 4;
  const int cb = blockIdx.y * BMC;
  const int ntiles = a.B * a.tiles_y * a.tiles_x;
  // contiguous tile range per block: neighbouring tiles share halo rows in L2 and belong to the same sample,
  // so the GroupNorm partial sums are flushed once per (block, sample) instead of once per tile
  const int tpb = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  // blocks are dealt to the 8 XCDs round-robin: give each XCD a contiguous eighth of the tile space, so that neighbouring
  // tile ranges (shared halo rows, same sample) meet in one L2
  int bx = blockIdx.x;
  if ((gridDim.x & 7) == 0) bx = (bx & 7) * ((int)gridDim.x >> 3) + (bx >> 3);
  const int t_begin = bx * tpb;
  const int t_end = min(ntiles, t_begin + tpb);
  if (t_begin >= t_end) return;
  const int nst = (t_end - t_begin) * KTOT;

  const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(a.x);
  const bf16* __restrict__ wg = reinterpret_cast<const bf16*>(a.w);
  bf16* __restrict__ yg = reinterpret_cast<bf16*>(a.y);

  // ---- stationary weights: A[kc][tap][j] = W[tap][cb + (wm*MI+j)*16 + n][kc*32 + 8g .. +8], one contiguous 1 KB
  // unit of the packed layout per (j, kc, tap) (conv_common.h: wpack_off_bf16); units past the last row block read as zero
  const int kct = (a.KRow ? a.KRow : a.KPad) >> 5;     // K chunks per packed row block
  const int r16 = (cb >> 4) + wm * MI;                 // first 16-row block of this wave
  bf16x8 A[KCH][TAPS][MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const bool in = (r16 + j) * 16 < a.RTot;
    const bf16* wj = wg + (((int64_t)(r16 + j) * kct + a.kc0) * TAPS << 9) + lane * 8;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        bf16x8 v = {};
        if (in) v = *reinterpret_cast<const bf16x8*>(wj + ((kc * TAPS + tap) << 9));
        A[kc][tap][j] = v;
      }
  }
  // 1x1 weights of the fused variants: A2[kc][j] = W2[cb + (wm*MI+j)*16 + n][kc*32 + 8g .. +8]
  bf16x8 A2[(VAR == WS_PLAIN || BS || GA) ? 1 : KCH][MI];
  if constexpr (VAR == WS_DUAL || VAR == WS_CENTER) {
    const bf16* __restrict__ wg2 = reinterpret_cast<const bf16*>(a.w2);
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const bool in = (r16 + j) * 16 < a.RTot;
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) {
        bf16x8 v = {};
        if (in) v = *reinterpret_cast<const bf16x8*>(wg2 + (((int64_t)(r16 + j) * kct + a.kc0 + kc) << 9) + lane * 8);
        A2[kc][j] = v;
      }
    }
  }

  // ---- staging: piece p = tid + q*256 -> halo pixel r = p>>2 (ly, lx), 16-byte part p&3.
  // Per image one buffer resource (num_records = image bytes): rows above / below the image fall outside it and
  // the hardware returns zeros; columns left / right of the image are forced out of range by hand.
  // (K chunks beyond the real channel count read neighbouring finite data; their packed weights are zero.)
  // (The resource ends with the last pixel's K channels, not with its row of x_ld channels: when the input is a channel slice
  // of a wider buffer, the bytes behind that point belong to the next image or to whatever follows the tensor in memory --
  // fp32 scratch read as bf16 has NaN patterns, and 0 x NaN poisoned the last pixels of the last image.)
  const int row_bytes = a.Wx * a.x_ld * 2;
  const int img_bytes = a.Hx * row_bytes - (a.x_ld - (VAR == WS_CENTER ? 2 : 1) * a.K) * 2;
  int st_rel[NP], st_lx[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int p = tid + q * 256;
    const int r = p >> 2;
    const int ly = r / LW, lx = r - ly * LW;
    // LDS slot (pixel r, 16-byte part p&3) holds channel group part ^ swz(lx): see SWZ_SH below
    const int gpart = (p & 3) ^ (((lx >> SWZ_SH) & 1) << 1);
    if constexpr (GA) {        // stage kk = 4*kc + tap of output pixel (ly, lx): input pixel (2*ly + tap/2, 2*lx + tap%2)
      st_lx[q] = p < PIECES ? 2 * lx : -100000;
      st_rel[q] = 2 * ly * row_bytes + 2 * lx * a.x_ld * 2 + gpart * 16;
    } else {
      st_lx[q] = p < PIECES ? lx - PAD : -100000;
      st_rel[q] = (ly - PAD) * row_bytes + (lx - PAD) * a.x_ld * 2 + gpart * 16;
    }
  }
  const int nr = n >> TWL, nc = n & (TW - 1);      // this lane's pixel inside a fragment
  // Fragment reads: lane (n, g) wants the 8 channels of group g of its pixel.  With the groups at their natural 16-byte
  // parts, each of ds_read_b128's four 16-lane groups hit every 16-byte slot of the 256-byte bank row twice (8 LDS cycles
  // per read instead of 4: the C >= 128 stages are LDS-read-bound).  Storing group g of halo column lx at part
  // g ^ 2*bit(lx) -- bit 2 of lx for 16-wide tiles and 8-wide 1x1 tiles, bit 1 for the 10-wide halo of 8-wide 3x3 tiles -- makes
  // every group cover all 16 slots (enumerated over all rows / taps: tools/lds_bank_model.py).  The LDS-DMA applies the
  // permutation for free by choosing which global piece a lane fetches; a reading lane's column is nc + tx.
  int frag_off[TAPW];
#pragma unroll
  for (int tx = 0; tx < TAPW; ++tx)
    frag_off[tx] = ((wn * NI * RPF + nr) * LW + nc) * 64 + ((g ^ ((((nc + tx) >> SWZ_SH) & 1) << 1)) * 16);

  TilePos ip;                     // issue position
  ip.set(t_begin, a.tiles_x, a.tiles_y);
  int ikc = 0, is = 0;
  auto issue_stage = [&]() {
    char* buf = smem + (is % NBUF) * BUF + wid * 1024;
    const bool live = is < nst;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(xg + (int64_t)ip.b * a.Hx * a.Wx * a.x_ld), 0, live ? img_bytes : 0, 0x00020000);
    const int x0 = GA ? 2 * ip.tx * TW + (ikc & 1) : ip.tx * TW;
    const int tile_off = GA ? (2 * ip.ty * TH + ((ikc >> 1) & 1)) * row_bytes + x0 * a.x_ld * 2 + (ikc >> 2) * 64
                            : ip.ty * TH * row_bytes + x0 * a.x_ld * 2 + ikc * 64;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const unsigned ix = (unsigned)(x0 + st_lx[q]);
      const unsigned off = ix < (unsigned)a.Wx ? (unsigned)(st_rel[q] + tile_off) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(buf + q * 4096), 16, off, 0, 0, 0);
    }
    ++is;
    if (++ikc == KTOT) { ikc = 0; ip.next(a.tiles_x, a.tiles_y); }
  };

#pragma unroll
  for (int k = 0; k < DIST; ++k) issue_stage();
  const unsigned long long t_setup = WS_DBG ? __builtin_amdgcn_s_memtime() : 0;
  const bool do_stats = a.gn_stats != nullptr;
  const bool do_chsum = a.ch_sum != nullptr;       // per-channel output sums (a bias gradient riding on a data gradient)
  int s = 0;
  TilePos cp;                     // compute position
  cp.set(t_begin, a.tiles_x, a.tiles_y);

  // epilogue addressing: lane (n, g) of frag (j, i) stores channels c0(j)..+3 of pixel (row wn*NI+i, column n)
  const int out_row = a.Wy * a.y_ld;
  const int lane_out = ((wn * NI * RPF + nr) * a.Wy + nc) * a.y_ld + cb + wm * MI * 16 + 4 * g;
  const int out_row2 = a.Wy * a.y2_ld;
  const int lane_out2 = ((wn * NI * RPF + nr) * a.Wy + nc) * a.y2_ld + cb + wm * MI * 16 + 4 * g;
  f32x4 s1[MI], s2[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) { s1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[j] = s1[j]; }

  // ---- WS_BS: fused GroupNorm-backward sums.  A lane's channels never change, so gamma / beta stay in registers; the
  // statistics of its channels' group and the FiLM / dropout factors are reloaded when the sample changes.
  constexpr bool bs2 = BS2;
  f32x4 s3[BS2 ? MI : 1], s4[BS2 ? MI : 1];
  f32x4 bs_ga[BS ? MI : 1], bs_be[BS ? MI : 1], bs_fac[BS2 ? MI : 1], bs_drp[BS2 ? MI : 1];
  float bs_rs[BS ? MI : 1], bs_mr[BS ? MI : 1];
  const bf16* __restrict__ bsY = reinterpret_cast<const bf16*>(a.bs_y);
  const bf16* __restrict__ bsS = reinterpret_cast<const bf16*>(a.bs_s);
  const int lane_bsy = ((wn * NI * RPF + nr) * a.Wy + nc) * a.bs_y_ld + cb + wm * MI * 16 + 4 * g;
  const int lane_bss = ((wn * NI * RPF + nr) * a.Wy + nc) * a.bs_s_ld + cb + wm * MI * 16 + 4 * g;
  auto bs_sample = [&](int b) {
    if constexpr (BS) {
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int c = min(cb + (wm * MI + j) * 16 + 4 * g, a.R - 4);
        const float* stp = a.bs_stats + ((int64_t)b * (a.R / a.cpg) + c / a.cpg) * 2;
        const float mean = stp[0] * a.bs_inv_n;
        const float var = fmaxf(stp[1] * a.bs_inv_n - mean * mean, 0.f);
        const float r = rsqrtf(var + a.bs_eps);
        bs_rs[j] = r;
        bs_mr[j] = mean * r;
        if constexpr (BS2) {
          f32x4 f = {1.f, 1.f, 1.f, 1.f}, d = f;
          if (a.bs_drop) d = *reinterpret_cast<const f32x4*>(a.bs_drop + (int64_t)b * a.R + c);
          if (a.bs_gf) f = *reinterpret_cast<const f32x4*>(a.bs_gf + (int64_t)b * a.R + c);
          bs_drp[j] = d;
          bs_fac[j] = f * d;
        }
      }
    }
  };
  if constexpr (BS) {
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int c = min(cb + (wm * MI + j) * 16 + 4 * g, a.R - 4);
      bs_ga[j] = *reinterpret_cast<const f32x4*>(a.bs_gamma + c);
      bs_be[j] = *reinterpret_cast<const f32x4*>(a.bs_beta + c);
      if constexpr (BS2) { s3[j] = f32x4{0.f, 0.f, 0.f, 0.f}; s4[j] = s3[j]; }
    }
    bs_sample(cp.b);
  }
  // per-(sample, channel) flush: DPP row sums -> LDS -> one atomic per channel and sum
  auto flush_bs = [&](int b) {
    if constexpr (BS) {
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int cl = (wm * MI + j) * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t1 = s1[j][r], t2 = s2[j][r], t3 = 0.f, t4 = 0.f;
          if constexpr (BS2) { t1 *= bs_fac[j][r]; t2 *= bs_fac[j][r]; t3 = row_sum16(s3[j][r]); t4 = row_sum16(s4[j][r]); }
          t1 = row_sum16(t1); t2 = row_sum16(t2);
          if (n == 0) *reinterpret_cast<f32x4*>(sm_stat + (wn * BMC + cl + r) * 4) = f32x4{t1, t2, t3, t4};
        }
        s1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        s2[j] = s1[j];
        if constexpr (BS2) { s3[j] = s1[j]; s4[j] = s1[j]; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int ns = bs2 ? 4 : 2;
      for (int o = tid; o < BMC * ns; o += 256) {
        const int cl = o / ns, k = o - cl * ns;
        if (cb + cl < a.R) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < WN; ++w) t += sm_stat[(w * BMC + cl) * 4 + k];
          atomicAdd(a.bs_sums + ((int64_t)b * a.R + cb + cl) * 4 + k, t);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();            // sm_stat may be rewritten by the next flush
    }
  };

  auto flush_stats = [&](int b) {
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int cl = (wm * MI + j) * 16 + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t1 = row_sum16(s1[j][r]), t2 = row_sum16(s2[j][r]);
        if (n == 0) {
          sm_stat[(wn * BMC + cl + r) * 2 + 0] = t1;
          sm_stat[(wn * BMC + cl + r) * 2 + 1] = t2;
        }
      }
      s1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      s2[j] = s1[j];
    }
    // raw barrier: __syncthreads() would drain the LDS-DMA prefetches (vmcnt(0)); only LDS writes matter here
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    gn_stats_flush<BMC, WN>(sm_stat, tid, cb, a.R, a.cpg, a.gn_stats + (int64_t)b * (a.R / a.cpg) * 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // sm_stat may be rewritten by the next flush
  };

  unsigned long long dbg_wait = 0, dbg_issue = 0, dbg_epi = 0, dbg_last = 0;
  constexpr int NST = MI * NI * (VAR == WS_DUAL ? 2 : 1);                    // stores of one full-tile epilogue
  constexpr bool FASTEPI = !BS && (DIST - 1) * NP + 3 * NST <= 63 && DIST <= 3;
  int epf0 = 0, epf1 = 0, epf2 = 0;                // 1: the epilogue 0 / 1 / 2 tiles back issued exactly NST stores (wave-uniform)
  for (int t = t_begin; t < t_end; ++t) {
    f32x4 acc[MI][NI], acc2[VAR == WS_DUAL ? MI : 1][VAR == WS_DUAL ? NI : 1];
    // register-heavy instantiations (288 weight VGPRs and two channel fragments per wave) prefetch the first fragment row
    // only; the second is fetched at the top of the epilogue, under the arithmetic of the first
    constexpr bool BS_HEAVY = BS && MI == 2 && KCH * TAPS * MI * 4 >= 288;
    constexpr int PJ = BS ? (BS_HEAVY ? 1 : MI) : 1;
    bf16x4 ypre[PJ][BS ? NI : 1], ylate[BS_HEAVY ? NI : 1];
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (VAR == WS_DUAL) acc2[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }

#pragma unroll
    for (int kc = 0; kc < KTOT; ++kc, ++s) {
      const unsigned long long d0 = WS_DBG ? __builtin_amdgcn_s_memtime() : 0;
      // this wave's pieces of stage s have landed.  vmcnt retires in issue order and counts stores too: what may still be
      // in flight is the DIST - 1 newer stages AND the output stores of every epilogue since stage s was requested (DIST
      // stages ago) -- without that term each tile start waited for the previous tile's write acknowledgements.  Only
      // epilogues that issued exactly NST stores (full tiles, store4_counted) are counted; anything else counts as zero,
      // which only waits longer.
      if constexpr (FASTEPI) {
        int c = 0;
        if (kc <= DIST - 1) {
          const int M = (DIST - 1 - kc) / KTOT;     // epilogues 0..M tiles back fall inside the window (compile-time after unrolling)
          c = epf0 + (M >= 1 ? epf1 : 0) + (M >= 2 ? epf2 : 0);
        }
        wait_vmcnt_plus<(DIST - 1) * NP, NST>(c);
      } else {
        wait_vmcnt<(DIST - 1) * NP>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // and its LDS reads of stage s-1 have returned
      __builtin_amdgcn_s_barrier();                  // ... for every wave: buffer (s-1) % NBUF is free
      const unsigned long long d1 = WS_DBG ? __builtin_amdgcn_s_memtime() : 0;
      if (WS_DBG && s == 0) t_first = d1;
      if constexpr (BS) {
        // the GroupNorm input at this tile's output pixels, fetched under the last stage's MFMAs.  Issued BEFORE the stage's
        // DMA: vmcnt retires in order, so waiting for these loads later must not wait for the newest prefetch stage too
        if (kc == KTOT - 1) {
          const bf16* yt = bsY + ((int64_t)(cp.b * a.Hy + cp.ty * TH) * a.Wy + cp.tx * TW) * a.bs_y_ld;
          const bool xv = cp.tx * TW + nc < a.W;
          const int rl = a.H - cp.ty * TH - wn * NI * RPF - nr;
#pragma unroll
          for (int j = 0; j < PJ; ++j) {
            const bool cv = cb + (wm * MI + j) * 16 + 4 * g < a.R;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
              bf16x4 v = {};
              if (cv && xv && i * RPF < rl) v = *reinterpret_cast<const bf16x4*>(yt + (unsigned)(lane_bsy + i * RPF * a.Wy * a.bs_y_ld + j * 16));
              ypre[j][i] = v;
            }
          }
        }
      }
      issue_stage();                                 // stage s + DIST (dummy past the end: keeps vmcnt uniform)
      const unsigned long long d2 = WS_DBG ? __builtin_amdgcn_s_memtime() : 0;
      if (WS_DBG) { dbg_wait += d1 - d0; dbg_issue += d2 - d1; dbg_last = d2; }
      const char* buf = smem + (s % NBUF) * BUF;
      if constexpr (VAR == WS_CENTER) {
        if (kc >= KCH) {                             // shortcut channels: centre tap only
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const bf16x8 Bc = *reinterpret_cast<const bf16x8*>(buf + frag_off[1] + ((i * RPF + 1) * LW + 1) * 64);
#pragma unroll
            for (int j = 0; j < MI; ++j)
              acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[kc - KCH][j], Bc, acc[j][i], 0, 0, 0);
          }
          continue;
        }
      }
      constexpr int KC = KCH;                        // bound for indexing A (kc < KCH here)
      // Fragment rows are fetched PFD rows ahead of the MFMAs that consume them (register ring Bq): issued and awaited back
      // to back -- what the compiler schedules on its own at this register pressure -- every row exposed the full LDS
      // latency, 18 times per stage at the deep levels (3.7k cycles per stage against 1152 of MFMA, s_memtime stamps).
      constexpr int NR = NI * RPF + TAPW - 1;        // halo rows of the fragments' first rows
      bf16x8 Bq[PFD + 1][TAPW];
#pragma unroll
      for (int p = 0; p < PFD && p < NR; ++p)
#pragma unroll
        for (int tx = 0; tx < TAPW; ++tx) Bq[p][tx] = *reinterpret_cast<const bf16x8*>(buf + frag_off[tx] + (p * LW + tx) * 64);
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {
        if (rr + PFD < NR) {
#pragma unroll
          for (int tx = 0; tx < TAPW; ++tx)
            Bq[(rr + PFD) % (PFD + 1)][tx] = *reinterpret_cast<const bf16x8*>(buf + frag_off[tx] + ((rr + PFD) * LW + tx) * 64);
        }
        __builtin_amdgcn_sched_barrier(0);           // keep the prefetch above this row's MFMAs
        const bf16x8 (&Bf)[TAPW] = Bq[rr % (PFD + 1)];
#pragma unroll
        for (int ty = 0; ty < TAPW; ++ty) {
          const int i = (rr - ty) / RPF;
          if (rr - ty >= 0 && (rr - ty) % RPF == 0 && i < NI) {
#pragma unroll
            for (int tx = 0; tx < TAPW; ++tx)
#pragma unroll
              for (int j = 0; j < MI; ++j)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kc % KC][ty * TAPW + tx][j], Bf[tx], acc[j][i], 0, 0, 0);
            if constexpr (VAR == WS_DUAL) {
              if (ty == 1) {
#pragma unroll
                for (int j = 0; j < MI; ++j)
                  acc2[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[kc % KC][j], Bf[1], acc2[j][i], 0, 0, 0);
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    const unsigned long long d3 = WS_DBG ? __builtin_amdgcn_s_memtime() : 0;
    // ---- epilogue of tile t = (cp.b, cp.ty, cp.tx)
    bf16* ytile = yg + ((int64_t)(cp.b * a.Hy + cp.ty * TH) * a.Wy + cp.tx * TW) * a.y_ld;   // wave-uniform
    bf16* y2tile = VAR == WS_DUAL ? reinterpret_cast<bf16*>(a.y2) + ((int64_t)(cp.b * a.Hy + cp.ty * TH) * a.Wy + cp.tx * TW) * a.y2_ld : nullptr;
    const bool xvalid = cp.tx * TW + nc < a.W;
    const int rows_left = a.H - cp.ty * TH - wn * NI * RPF - nr;   // valid rows below this lane's first row
    bool full_tile = false;
    if (TAPS == 1 && VAR == WS_PLAIN && a.shuffle) {
      // ConvTranspose2d(2,2) as four 1x1 GEMMs: row = tap*RPad + co, the output pixel of input pixel (y, x) under tap
      // (ty, tx) is (2y + ty, 2x + tx); bias added here (models.py:190)
      bf16* ybase = yg + ((int64_t)(cp.b * a.Hy + 2 * cp.ty * TH) * a.Wy + 2 * cp.tx * TW) * a.y_ld;
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int row = cb + (wm * MI + j) * 16 + 4 * g;
        const int tp = row / a.RPad, co = row - tp * a.RPad;
        if (row >= a.RTot || co >= a.R) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          if (xvalid && i * RPF < rows_left) {
            const int oy = 2 * (wn * NI * RPF + nr + i * RPF) + (tp >> 1), ox = 2 * nc + (tp & 1);
            store4(ybase + (unsigned)((oy * a.Wy + ox) * a.y_ld + co), acc[j][i] + bv);
          }
        }
      }
    } else if (FASTEPI && a.epi_counted && !a.accumulate && (cp.tx + 1) * TW <= a.W && (cp.ty + 1) * TH <= a.H && cb + BMC <= a.R) {
      // full tile, every channel valid: NST unconditional stores, counted by the stage waits of the next tile(s)
      full_tile = true;
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const f32x4 v = acc[j][i];
          if (!(SX_WS_ABL & 1)) store4_counted(ytile + (unsigned)(lane_out + i * RPF * out_row + j * 16), v);
          if (SX_WS_ABL & 2) { s1[j] += v; }
          else if (do_stats) { s1[j] += v; s2[j] += v * v; }
          else if (do_chsum) s1[j] += v;
          if constexpr (VAR == WS_DUAL) { if (!(SX_WS_ABL & 1)) store4_counted(y2tile + (unsigned)(lane_out2 + i * RPF * out_row2 + j * 16), acc2[j][i]); else s1[j] += acc2[j][i]; }
        }
    } else {
    if constexpr (BS_HEAVY) {
      const bf16* yt = bsY + ((int64_t)(cp.b * a.Hy + cp.ty * TH) * a.Wy + cp.tx * TW) * a.bs_y_ld;
      const bool cv = cb + (wm * MI + 1) * 16 + 4 * g < a.R;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        bf16x4 v = {};
        if (cv && xvalid && i * RPF < rows_left) v = *reinterpret_cast<const bf16x4*>(yt + (unsigned)(lane_bsy + i * RPF * a.Wy * a.bs_y_ld + 16));
        ylate[i] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const bool cvalid = cb + (wm * MI + j) * 16 + 4 * g < a.R;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        f32x4 v = acc[j][i];
        if (cvalid && xvalid && i * RPF < rows_left) {
          bf16* dst = ytile + (unsigned)(lane_out + i * RPF * out_row + j * 16);
          if (a.accumulate) v += load4(dst);
          if constexpr (BS) {
            // sums from the value as STORED (bf16), exactly what the element-wise apply pass will read back
            const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *reinterpret_cast<bf16x4*>(dst) = o;
            f32x4 sv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (bs2) sv = load4(bsS + ((int64_t)(cp.b * a.Hy + cp.ty * TH) * a.Wy + cp.tx * TW) * a.bs_s_ld + (unsigned)(lane_bss + i * RPF * a.Wy * a.bs_s_ld + j * 16));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float gq = (float)o[r];
              const float yq = (float)((BS_HEAVY && j >= PJ) ? ylate[i][r] : ypre[j < PJ ? j : 0][i][r]);
              const float nh = yq * bs_rs[j] - bs_mr[j];
              const float z = nh * bs_ga[j][r] + bs_be[j][r];
              const float sg = sigmoidf_(z);
              const float dz = gq * (sg * (1.0f + z * (1.0f - sg)));
              s1[j][r] += dz * nh;
              s2[j][r] += dz;
              if constexpr (bs2) {
                s3[j][r] += gq * ((z * sg + sv[r]) * bs_drp[j][r]);
                s4[j][r] += gq;
              }
            }
          } else if (!(SX_WS_ABL & 1))
          store4(dst, v);
          if (SX_WS_ABL & 2) { s1[j] += v; }
          else if (do_stats) { s1[j] += v; s2[j] += v * v; }
          else if (do_chsum) s1[j] += v;
          if constexpr (VAR == WS_DUAL) { if (!(SX_WS_ABL & 1)) store4(y2tile + (unsigned)(lane_out2 + i * RPF * out_row2 + j * 16), acc2[j][i]); else s1[j] += acc2[j][i]; }
        }
      }
    }
    }
    epf2 = epf1; epf1 = epf0; epf0 = (full_tile && !(SX_WS_ABL & 1)) ? 1 : 0;
    if (WS_DBG) dbg_epi += __builtin_amdgcn_s_memtime() - d3;
    const int b_done = cp.b;
    cp.next(a.tiles_x, a.tiles_y);
    if (do_stats && (t + 1 == t_end || cp.b != b_done)) flush_stats(b_done);
    if constexpr (BS) {
      if (t + 1 == t_end || cp.b != b_done) {
        flush_bs(b_done);
        if (t + 1 < t_end) bs_sample(cp.b);
      }
    }
  }
  if (do_chsum && !do_stats) {       // one flush per block: the sums run over every sample
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int cl = (wm * MI + j) * 16 + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t1 = row_sum16(s1[j][r]);
        if (n == 0) sm_stat[(wn * BMC + cl + r) * 2] = t1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid < BMC && cb + tid < min(a.R, a.ch_sum_n)) {
      float t1 = 0.f;
#pragma unroll
      for (int w = 0; w < WN; ++w) t1 += sm_stat[(w * BMC + tid) * 2];
      atomicAdd(a.ch_sum + cb + tid, t1);
    }
  }
  const unsigned long long t_loop = WS_DBG ? __builtin_amdgcn_s_memtime() : 0;
  wait_vmcnt<0>();     // drain the dummy prefetches before the wave (and its LDS allocation) retires
  if (WS_DBG && a.stamps && tid == 0) {
    unsigned long long* o = a.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
    o[0] = t_entry; o[1] = t_setup; o[2] = t_first; o[3] = t_loop; o[4] = __builtin_amdgcn_s_memtime();
    o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = (unsigned long long)(t_end - t_begin);
    o[7] = dbg_wait; o[8] = dbg_issue; o[9] = dbg_epi; o[10] = (unsigned long long)nst;   // summed over the block's stages / tiles (wave 0)
  }
}

template <int KCH, int TAPS, int MI, int NI, int WM, int WN, int TWL, int VAR>
static int launch_ws(ConvArgs a, hipStream_t st) {
  constexpr int TAPW = TAPS == 9 ? 3 : 1;
  constexpr int TW = 1 << TWL;
  constexpr int BMC = WM * MI * 16, TH = WN * NI * (16 / TW);
  constexpr int NPIX = (TH + TAPW - 1) * (TW + TAPW - 1);
  constexpr int NP = (NPIX * 4 + 255) / 256;
  // Light configurations (few weight registers, 8 or more tiles per block) run TWO blocks per CU with a 3-stage ring each
  // (<= 80 KB of LDS, < 256 VGPRs): the second wave on a SIMD hides the LDS latency the first one cannot (profiles/
  // r02_conv_ws_stamps.txt).  Measured per configuration (tools/conv_layers.py, SX_WS_PER_CU=2 against 1): level-0 3x3
  // 22.4 -> 19.9 us, level-0 / level-1 joint data gradients 34.4 -> 30.2 and 29.4 -> 23.5, 1x1 head 15.3 -> 11.8,
  // transposed forward at 64x32 18.4 -> 16.7, level-0 fused-sums data gradient 30.0 -> 26.9 (all 32-channel 3x3 forms
  // once __launch_bounds__(256, 2) keeps them under 256 VGPRs); the 64-channel forms spill 300-400 bytes per lane at that
  // budget (21 -> 40 us; the 128-channel fused-shortcut forward 27 -> 49 us) and the deeper ones have too few tiles per
  // block: they keep one block per CU.  128-channel joint gradient 24.9 -> 22.9, 64-channel fused forward 15.3 -> 14.5 us.
  constexpr bool TWO = (TAPS == 9 && KCH == 1 && MI == 2 && NI == 4 && TWL == 4) ||
                       (VAR == WS_CENTER && KCH == 1 && MI == 2) || (TAPS == 9 && KCH == 2 && MI == 1 && NI == 8 && TWL == 4) ||
                       (TAPS == 9 && KCH == 4 && MI == 1 && NI == 8 && TWL == 4 && (VAR == WS_PLAIN || VAR == WS_CENTER)) ||
                       (VAR == WS_PLAIN && TAPS == 1 && KCH <= 2 && TWL == 4) ||
                       (VAR == WS_DUAL && KCH == 2 && MI == 1 && NI == 4 && WM == 2 && TWL == 4);
  constexpr int NBUF = TWO ? 3 : SX_WS_NBUF;                    // NBUF - 1 stages in flight per block
  const size_t smem = (size_t)NBUF * NP * 4096 + (size_t)WN * BMC * (VAR == WS_BS ? 4 : 2) * sizeof(float);
  auto kern = conv_ws_kernel<KCH, TAPS, MI, NI, WM, WN, NBUF, TWL, VAR>;
  static std::atomic<unsigned long long> done{0};
  SX_TRY(ensure_max_lds(reinterpret_cast<const void*>(kern), done, "hipFuncSetAttribute(conv_ws)"));
  a.TH = TH; a.TW = TW;
  static const int epi_counted = env_int("SX_WS_EPI_COUNTED", 1);
  a.epi_counted = epi_counted;
  if (WS_DBG) a.stamps = getenv("SX_WS_STAMPS") ? (unsigned long long*)strtoull(getenv("SX_WS_STAMPS"), nullptr, 0) : nullptr;   // diagnostic build only
  a.tiles_x = cdiv(a.W, TW);
  a.tiles_y = cdiv(a.H, TH);
  const int ntiles = a.B * a.tiles_x * a.tiles_y;
  const int gy = cdiv(a.RTot, BMC);
  // persistent grid: about one block per CU (registers allow one wave per SIMD); light variants get two
  static const int per_cu_env = env_int("SX_WS_PER_CU", 0);     // tuning knob (0 = the per-configuration choice), read once per process
  const int per_cu = per_cu_env > 0 ? per_cu_env : (TWO ? 2 : 1);
  int gx = std::max(1, std::min(ntiles, (256 * per_cu) / gy));
  gx = cdiv(ntiles, cdiv(ntiles, gx));       // contiguous tile ranges: no empty trailing blocks
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), smem, st, a);
  SX_LAUNCH_CHECK("conv_ws");
  return SX_OK;
}

template <int KCH, int TAPS, int VAR>
static int ws_pick_tile(const ConvArgs& a, hipStream_t st) {
  if (a.W < 16) {
    // 8-wide tiles, 64 channels per block (grid.y = RTot/64): few pixels, many channels -- the deepest levels.
    // 16-row tiles unless that leaves most of the chip without a block.
    const int gy = cdiv(a.RTot, 64);
    const int nt16 = a.B * cdiv(a.H, 16) * cdiv(a.W, 8);
    if constexpr (KCH == 8 && VAR == WS_DUAL) return 1;      // would spill: two launches instead
    else {
      if (nt16 * gy >= 256) return launch_ws<KCH, TAPS, 1, 8, 4, 1, 3, VAR>(a, st);
      return launch_ws<KCH, TAPS, 1, 4, 4, 1, 3, VAR>(a, st);
    }
  }
  // register budget: the stationary weights take KCH*TAPS*MI*4 VGPRs (+ KCH*MI*4 for a fused 1x1), the accumulators
  // MI*NI*4 (twice that with a second output); configurations that would spill use 16-channel wave tiles (MI = 1)
  constexpr bool LIGHT = VAR == WS_DUAL && KCH >= 4;
  if (a.RTot <= 32) {
    if constexpr (KCH > 4) return 1;
    else if constexpr (LIGHT) return launch_ws<KCH, TAPS, 1, 4, 2, 2, 4, VAR>(a, st);   // 32 ch x 128 px
    else if constexpr (VAR == WS_DUAL && KCH == 2) {
      static const int light2 = env_int("SX_WS_DUAL2", 1);
      if (light2) return launch_ws<KCH, TAPS, 1, 4, 2, 2, 4, VAR>(a, st);
      return launch_ws<KCH, TAPS, 2, 4, 1, 4, 4, VAR>(a, st);
    }
    else return launch_ws<KCH, TAPS, 2, 4, 1, 4, 4, VAR>(a, st);                        // 32 ch x 256 px (16 rows)
  } else if (a.RTot <= 64) {
    // 64 input channels: 16-channel wave tiles (72 weight registers) fit two blocks per CU: 17.6 -> 15.9 us, with the
    // fused sums 21.7 -> 18.5 us against the 32-channel wave tiles at one block per CU
    if constexpr (KCH == 2 && TAPS == 9 && (VAR == WS_PLAIN || VAR == WS_BS)) return launch_ws<KCH, TAPS, 1, 8, 4, 1, 4, VAR>(a, st);
    else if constexpr (KCH <= 4 && !LIGHT) return launch_ws<KCH, TAPS, 2, 4, 2, 2, 4, VAR>(a, st);   // 64 ch x 128 px (8 rows)
    else return launch_ws<KCH, TAPS, 1, 8, 4, 1, 4, VAR>(a, st);                                // 64 ch x 128 px
  } else {
    // (the fused-sums variant of the 288-weight-register tile spills and its one-tile blocks expose all of the extra
    // arithmetic: measured 37.6 us against 19.9 + 10 us for convolution + separate reduction -> two launches there)
    // 128 input channels: 16-channel wave tiles, two blocks per CU (16.5 -> 15.6 us against 32-channel wave tiles)
    // (its fused-sums form spills 184 bytes per lane at the two-block register budget: step 1.728 against 1.720 ms)
    if constexpr (VAR == WS_BS && KCH * TAPS * 2 * 4 >= 288) return 1;
    else if constexpr (KCH == 4 && VAR == WS_PLAIN && TAPS == 9) return launch_ws<KCH, TAPS, 1, 8, 4, 1, 4, VAR>(a, st);
    else if constexpr (KCH <= 4 && (VAR == WS_PLAIN || VAR == WS_BS)) return launch_ws<KCH, TAPS, 2, 8, 4, 1, 4, VAR>(a, st);   // 128 ch x 128 px
    else return launch_ws<KCH, TAPS, 1, 8, 4, 1, 4, VAR>(a, st);                      // 64 ch x 128 px, grid.y = RTot/64
  }
}

template <int TAPS, int VAR>
static int ws_pick_k(const ConvArgs& a, hipStream_t st) {
  switch (a.KPad / 32) {
    case 1: return ws_pick_tile<1, TAPS, VAR>(a, st);
    case 2: return ws_pick_tile<2, TAPS, VAR>(a, st);
    case 4: return ws_pick_tile<4, TAPS, VAR>(a, st);
    case 8: return ws_pick_tile<8, TAPS, VAR>(a, st);
  }
  return 1;
}

// 0 = launched, 1 = not covered (caller falls back to the generic kernel), < 0 = error
int conv_ws_try(const ConvArgs& a, int mode, hipStream_t st, int var) {
  if (mode == SX_CONVT_FWD) {
    // four 1x1 GEMMs with a pixel-shuffle store (+ bias); bias needs 16-byte loads of 4 channels
    // (512 input channels as two K halves, the second accumulating through the pixel-shuffle store: 32 + 51 us against 58 us
    // on the generic kernel for cfg 5's first transposed convolution -- not kept)
    if (var != WS_PLAIN || a.W < 8 || a.R % 4 || a.accumulate || a.gn_stats || a.KPad % 32 || a.KPad > 256) return 1;
    if (a.bias && ((uintptr_t)a.bias % 16)) return 1;
    return ws_pick_k<1, WS_PLAIN>(a, st);
  }
  if (mode == SX_CONVT_BWD) {
    // 1x1 over the 4 taps x K/32 chunks of the packed [row block][chunk][tap] layout (K = 128 -> 16 stationary units)
    if (var != WS_PLAIN || a.W < 8 || a.R % 4 || a.accumulate || a.gn_stats || a.ch_sum || a.bias || a.KPad % 32 || a.K != a.KPad) return 1;
    ConvArgs h = a;
    h.KRow = 4 * a.KPad;
    switch (a.KPad / 32) {
      case 1: return ws_pick_tile<4, 1, WS_GATHER>(h, st);
      case 2: return ws_pick_tile<8, 1, WS_GATHER>(h, st);
      case 4: return ws_pick_tile<16, 1, WS_GATHER>(h, st);
      case 8: {
        // 256 channels = 32 stationary units: two K halves of 16 (units are [chunk][tap], so the second half starts at unit 16
        // and 128 channels into every input pixel), the second accumulating (cfg 5: 21.7 + 36.3 us against 85 on the generic kernel)
        static const int gather_halves = env_int("SX_WS_GATHER_HALVES", 1);
        if (!gather_halves) return 1;
        h.K = h.KPad = 128;
        const int rc = ws_pick_tile<16, 1, WS_GATHER>(h, st);
        if (rc != 0) return rc;
        h.x = reinterpret_cast<const bf16*>(a.x) + 128;
        h.kc0 = 16; h.accumulate = 1;
        return ws_pick_tile<16, 1, WS_GATHER>(h, st);
      }
    }
    return 1;
  }
  if (mode != SX_CONV3 && mode != SX_CONV1) return 1;
  if (a.W < 8 || a.bias || a.shuffle || a.R % 4) return 1;
  if (a.KPad % 32) return 1;
  if (var == WS_DUAL) {
    if (mode != SX_CONV3 || a.accumulate || !a.w2 || !a.y2 || a.y2_ld % 4) return 1;
    return ws_pick_k<9, WS_DUAL>(a, st);
  }
  if (var == WS_BS) {
    // fused GroupNorm-backward sums: a lane's four channels must share a group; no other epilogue extras
    if (a.accumulate || a.gn_stats || a.ch_sum || a.KPad > 256 || a.cpg <= 0 || a.cpg % 4 || a.R % a.cpg) return 1;
    if (a.bs_mode != 1 || mode != SX_CONV3) return 1;          // instantiated: 3x3 convolutions with GroupNorm + SiLU sums
    if (!a.bs_y || !a.bs_stats || !a.bs_gamma || !a.bs_beta || !a.bs_sums || a.bs_y_ld % 4 || (a.bs_mode == 2 && (!a.bs_s || a.bs_s_ld % 4))) return 1;
    if (((uintptr_t)a.bs_gamma | (uintptr_t)a.bs_beta | (uintptr_t)a.bs_gf | (uintptr_t)a.bs_drop | (uintptr_t)a.bs_sums) % 16) return 1;
    return ws_pick_k<9, WS_BS>(a, st);
  }
  if (var == WS_CENTER) {
    if (mode != SX_CONV3 || !a.w2 || a.K % 32 || a.x_ld < 2 * a.K) return 1;
    return ws_pick_k<9, WS_CENTER>(a, st);
  }
  if (a.KPad == 512 && a.K == 512) {
    // 512 input channels do not fit the register-resident weight slice: two K halves, the second accumulating into the
    // first one's output (GroupNorm statistics, if any, come from the second, i.e. from the complete sums)
    ConvArgs h = a;
    h.K = h.KPad = 256; h.KRow = 512; h.gn_stats = nullptr;
    int rc = mode == SX_CONV3 ? ws_pick_k<9, WS_PLAIN>(h, st) : ws_pick_k<1, WS_PLAIN>(h, st);
    if (rc != 0) return rc;
    h.gn_stats = a.gn_stats;
    h.x = reinterpret_cast<const bf16*>(a.x) + 256;
    h.kc0 = 8;
    h.accumulate = 1;
    return mode == SX_CONV3 ? ws_pick_k<9, WS_PLAIN>(h, st) : ws_pick_k<1, WS_PLAIN>(h, st);
  }
  return mode == SX_CONV3 ? ws_pick_k<9, WS_PLAIN>(a, st) : ws_pick_k<1, WS_PLAIN>(a, st);
}

}  // namespace sx
