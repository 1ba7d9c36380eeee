// Whole-network orchestration of the FiLM-conditioned residual U-Net (models.py:98-213) on top of
// the kernels of this library: UNet.encode / UNet.decode and their backward passes, each ONE call
// from the host language.  Pure host code: it computes where every tensor lives inside the
// caller-provided workspace and enqueues kernels on the caller's stream.
//
// HBM layout (NHWC, activation dtype T): per encoder level l one "cat" buffer [B,Hl,Wl,2*Cl] whose
// upper channel half IS the skip tensor (written by the encoder's block epilogue) and whose lower
// half is written by the decoder's transposed convolution -- torch.cat (models.py:200) costs nothing.
// Every DoubleConv block keeps Y1 (raw conv1), A1, Y2 (raw conv2), S (shortcut), R (pre-FiLM sum),
// and its GroupNorm statistics for backward.
#include "sx_common.h"
#include "film.h"
#include "wgrad_common.h"
#include "norm_act.h"
#include "conv_common.h"
#include <vector>
#include <algorithm>

namespace sx {

static constexpr float GN_EPS = 1e-5f;
static constexpr int BIAS_SLOTS = 8;      // rows of a transposed convolution's bias-gradient scratch (channel_sum_slots)

struct BlockBuf {   // byte offsets into the workspace
  int64_t Y1, A1, Y2, S, R, O;      // O: block output (skip half of a cat buffer for encoders)
  int o_ld;
  int64_t st1, st2;                 // GN statistics [B][G][2] fp32
  int64_t gf, bf, pre, dpre;        // FiLM: [B][C] fp32 x2, [B][Hf] fp32 x2
  int64_t dO, dY, dY1, dA1, dR;     // gradients (T); dY = d(conv2 out); dY1 = d(conv1 out) and dR = d(pre-FiLM sum) are the
                                    // two channel halves of ONE [B,H,W,2C] buffer (fused conv1 + shortcut data gradient)
  int64_t sums1, sums2;             // [B][C][4] fp32
  int64_t dX; int dx_ld;            // gradient w.r.t. the block input
  int64_t pool_ix;                  // encoder levels: window position (0..3) of every pooled value, one byte each [B][H/2][W/2][C]
  int Cin, Cin_real, C, H, W, G;
  int64_t in; int in_ld;            // block input
  // packed weights (byte offsets into wpack)
  int64_t w1f, w1d, w2f, w2d, wsf, wsd;
  int pidx;                         // index of net.0.weight in the parameter table
  int film_pidx;                    // index of film net.0.weight, -1 if none
};

struct Plan {
  sx_unet_desc d;
  int esz;                    // sizeof(T)
  int in_pad, out_pad;
  int depth;
  int C[8], H[8], W[8];       // per level 0..depth
  std::vector<BlockBuf> blk;  // encoders 0..depth-1, bot (index depth), decoders (index depth+1+j)
  int64_t X0, cat[8], dcat[8], Pool[8], pred, dpred, dparams;
  int64_t stats_begin, stats_end, sums_begin, sums_end;
  int64_t bias_scr[8], head_bias_scr, bwd_zero_end;   // bias-gradient scratch and dparams follow the decoder sums
  int64_t wg_ws; int64_t wg_ws_bytes;
  // convT / head packed weights and parameter indices
  int64_t upf[8], upb[8]; int up_pidx[8];
  int64_t headf, headd, head_bias; int out_pidx;
  int64_t ws_bytes, wpack_bytes;
  int nparams;
};

static int build_plan(const sx_unet_desc* dp, Plan& p) {
  if (!dp) { set_error("null descriptor"); return SX_ERR_INVALID; }
  const sx_unet_desc& d = *dp;
  p.d = d;
  if (d.dtype != SX_F32 && d.dtype != SX_BF16) { set_error("unet: bad dtype %d", d.dtype); return SX_ERR_INVALID; }
  if (d.depth < 1 || d.depth > 6) { set_error("unet: depth %d outside 1..6", d.depth); return SX_ERR_UNSUPPORTED; }
  if (d.base < 8 || d.base % 8) { set_error("unet: base=%d must be a positive multiple of 8 for the HIP path", d.base); return SX_ERR_UNSUPPORTED; }
  if (d.groups_gn < 1 || d.in_ch < 1 || d.out_ch < 1 || d.B < 1) { set_error("unet: bad sizes"); return SX_ERR_INVALID; }
  if (d.use_film && (d.P < 1 || d.P > 64 || d.film_hidden < 1 || d.film_hidden > 1024)) { set_error("unet: FiLM sizes unsupported (P=%d hidden=%d)", d.P, d.film_hidden); return SX_ERR_UNSUPPORTED; }
  if ((d.H >> d.depth) < 1 || (d.W >> d.depth) < 1) { set_error("unet: grid %dx%d too small for depth %d", d.H, d.W, d.depth); return SX_ERR_UNSUPPORTED; }
  if (d.base << d.depth > 1024) { set_error("unet: bottleneck width %d > 1024", d.base << d.depth); return SX_ERR_UNSUPPORTED; }
  p.esz = (int)dtype_size(d.dtype);
  p.depth = d.depth;
  p.in_pad = round_up(d.in_ch, 8);
  p.out_pad = round_up(d.out_ch, 8);
  for (int l = 0; l <= d.depth; ++l) {
    p.C[l] = d.base << l;
    p.H[l] = l == 0 ? d.H : p.H[l - 1] / 2;
    p.W[l] = l == 0 ? d.W : p.W[l - 1] / 2;
    const int g = std::min(d.groups_gn, p.C[l]);
    if (p.C[l] % g) { set_error("unet: GroupNorm groups %d do not divide %d channels", g, p.C[l]); return SX_ERR_INVALID; }
  }
  const int B = d.B, D = d.depth;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { int64_t o = off; off += (int64_t)align_up((size_t)bytes, 256); return o; };
  auto act = [&](int l, int C) { return take((int64_t)B * p.H[l] * p.W[l] * C * p.esz); };

  p.X0 = act(0, p.in_pad);
  for (int l = 0; l < D; ++l) {
    p.cat[l] = act(l, 2 * p.C[l]);
    p.dcat[l] = act(l, 2 * p.C[l]);
    p.Pool[l] = act(l + 1, p.C[l]);
  }
  p.pred = act(0, p.out_pad);
  p.dpred = act(0, p.out_pad);

  p.blk.assign(2 * D + 1, BlockBuf());
  // parameter-table indices
  const int n_blocks = 2 * D + 1;
  const int idx_bot = 7 * D, idx_up0 = 7 * (D + 1), idx_dec0 = idx_up0 + 2 * D, idx_out = idx_dec0 + 7 * D;
  const int idx_film0 = idx_out + 2;
  p.out_pidx = idx_out;
  p.nparams = idx_film0 + (d.P > 0 ? 4 * n_blocks : 0);

  for (int k = 0; k < n_blocks; ++k) {
    BlockBuf& b = p.blk[k];
    int l;
    if (k < D) {            // encoder level k
      l = k;
      b.Cin_real = k == 0 ? d.in_ch : p.C[k - 1];
      b.Cin = k == 0 ? p.in_pad : p.C[k - 1];
      b.in = k == 0 ? p.X0 : p.Pool[k - 1];
      b.in_ld = b.Cin;
      b.pidx = 7 * k;
      b.film_pidx = d.P > 0 ? idx_film0 + 4 * k : -1;
    } else if (k == D) {    // bottleneck
      l = D;
      b.Cin_real = b.Cin = p.C[D - 1];
      b.in = p.Pool[D - 1];
      b.in_ld = b.Cin;
      b.pidx = idx_bot;
      b.film_pidx = d.P > 0 ? idx_film0 + 4 * D : -1;
    } else {                // decoder j at level D-1-j
      const int j = k - D - 1;
      l = D - 1 - j;
      b.Cin_real = b.Cin = 2 * p.C[l];
      b.in = p.cat[l];
      b.in_ld = 2 * p.C[l];
      b.pidx = idx_dec0 + 7 * j;
      b.film_pidx = d.P > 0 ? idx_film0 + 4 * (D + 1 + j) : -1;
    }
    b.C = p.C[l]; b.H = p.H[l]; b.W = p.W[l];
    b.G = std::min(d.groups_gn, b.C);
    b.Y1 = act(l, b.C); b.A1 = act(l, b.C); b.Y2 = act(l, b.C); b.S = act(l, b.C);
    if (k < D) { b.O = p.cat[l] + (int64_t)b.C * p.esz; b.o_ld = 2 * b.C; }
    else { b.O = act(l, b.C); b.o_ld = b.C; }
    b.dO = act(l, b.C); b.dY = act(l, b.C); b.dA1 = act(l, b.C);
    b.dY1 = act(l, 2 * b.C); b.dR = b.dY1 + (int64_t)b.C * p.esz;
    b.gf = take((int64_t)B * b.C * 4); b.bf = take((int64_t)B * b.C * 4);
    b.pre = take((int64_t)B * std::max(1, d.film_hidden) * 4); b.dpre = take((int64_t)B * std::max(1, d.film_hidden) * 4);
  }
  // gradient w.r.t. block inputs
  for (int k = 0; k < n_blocks; ++k) {
    BlockBuf& b = p.blk[k];
    if (k == 0) { b.dX = take((int64_t)B * b.H * b.W * p.in_pad * p.esz); b.dx_ld = p.in_pad; }   // d(input image), SX_BWD_INPUT_GRAD
    else if (k <= D) { b.dX = take((int64_t)B * b.H * b.W * b.Cin * p.esz); b.dx_ld = b.Cin; }   // = d(pool_{k-1})
    else { const int l = D - 1 - (k - D - 1); b.dX = p.dcat[l]; b.dx_ld = 2 * p.C[l]; }
    b.pool_ix = k < D ? take((int64_t)B * (b.H / 2 + 1) * (b.W / 2 + 1) * b.C) : 0;
  }
  // statistics and backward sums: contiguous so one memset clears them
  p.stats_begin = off;
  for (auto& b : p.blk) { b.st1 = take((int64_t)B * b.G * 2 * 4); b.st2 = take((int64_t)B * b.G * 2 * 4); }
  p.stats_end = off;
  p.sums_begin = off;
  for (auto& b : p.blk) { b.sums1 = take((int64_t)B * b.C * 4 * 4); b.sums2 = take((int64_t)B * b.C * 4 * 4); }
  p.sums_end = off;
  for (int j = 0; j < D; ++j) p.bias_scr[j] = take((int64_t)BIAS_SLOTS * p.C[D - 1 - j] * 4);      // [slot][C]
  p.head_bias_scr = take((int64_t)SX_HEAD_BIAS_SLOTS * p.out_pad * 4);      // [slot][out_ch]: channel_sum fills slot 0, the loss kernel all
  p.bwd_zero_end = off;
  p.dparams = take((int64_t)B * std::max(1, d.P) * 4);
  // wgrad slabs: every layer of a backward half keeps its own slab until the fused reduction at the end of the half
  // (WgQueue flushes early if it ever runs out, so this is a capacity, not a requirement)
  int64_t wgmax = 0;
  auto wneed = [&](int mode, int h, int w, int ca, int cb) {
    wgmax += (int64_t)align_up((size_t)sx_conv_wgrad_workspace_bytes(mode, B, h, w, ca, cb), 256);
  };
  for (auto& b : p.blk) {
    wneed(SX_CONV3, b.H, b.W, b.C, b.C);
    wneed(SX_CONV3, b.H, b.W, b.C, b.Cin);
    wneed(SX_CONV1, b.H, b.W, b.C, b.Cin);
    if (2 * b.C == 64) wneed(SX_CONV3, b.H, b.W, 2 * b.C, b.Cin);     // fused conv1 + shortcut launch (WgQueue::add_dual)
  }
  for (int j = 0; j < D; ++j) {
    const int l = D - 1 - j;
    wneed(SX_CONVT_FWD, p.H[l + 1], p.W[l + 1], p.C[l + 1], p.C[l]);
  }
  wneed(SX_CONV1, d.H, d.W, p.out_pad, p.C[0]);
  p.wg_ws_bytes = wgmax;
  p.wg_ws = take(wgmax);
  p.ws_bytes = off;

  // ---- packed weights
  int64_t woff = 0;
  auto wtake = [&](int mode, int rows, int k) {
    int64_t o = woff;
    woff += (int64_t)align_up((size_t)(sx_packed_weight_elems(mode, d.dtype, rows, k) * p.esz), 256);
    return o;
  };
  for (auto& b : p.blk) {
    b.w1f = wtake(SX_CONV3, b.C, b.Cin_real); b.w1d = wtake(SX_CONV3, b.Cin_real, b.C);
    b.w2f = wtake(SX_CONV3, b.C, b.C);        b.w2d = wtake(SX_CONV3, b.C, b.C);
    b.wsf = wtake(SX_CONV1, b.C, b.Cin_real); b.wsd = wtake(SX_CONV1, b.Cin_real, b.C);
  }
  for (int j = 0; j < D; ++j) {
    const int l = D - 1 - j;
    p.upf[j] = wtake(SX_CONVT_FWD, p.C[l], p.C[l + 1]);
    p.upb[j] = wtake(SX_CONVT_BWD, p.C[l + 1], p.C[l]);
    p.up_pidx[j] = idx_up0 + 2 * j;
  }
  p.headf = wtake(SX_CONV1, p.out_pad, p.C[0]);
  p.headd = wtake(SX_CONV1, p.C[0], p.out_pad);
  p.head_bias = woff; woff += 256;
  p.wpack_bytes = woff;
  return SX_OK;
}

// fp32 tier, arithmetic per architecture.  Two-plane split products carry ~4e-6 per convolution; through the whole network and
// its GroupNorm backward (differences of large sums) that came to a worst per-tensor gradient error of 8e-5 at cfg A
// (bottleneck 256 channels) and 4.1e-4 at BASELINE configs[4] (bottleneck 512, 256x128; smooth loss, so this is arithmetic,
// not a loss kink: tests/test_gpu_fullsize.py) against 3.3e-5 with exact fp32 MFMA -- all of it from the forward / data-gradient
// convolutions (tools/fp32_mix.py), none from the weight gradients.  The contract is 1e-4, so networks whose widest layer
// reaches 512 channels run their convolutions on THREE planes (conv_x3.hip, six MFMAs per block, fp32-grade products).
// SX_FP32_WIDE = 2 keeps them on two planes, 0 sends them to the exact fp32 MFMA kernels.
static int plan_wide_mode(const Plan& p) {
  if (p.d.dtype != SX_F32 || p.C[p.depth] < 512) return 0;
  const int w = env_int("SX_FP32_WIDE", 3);       // read per call (a getenv): tests switch it
  return w == 2 ? 0 : (w == 0 ? 1 : 3);
}

static inline char* at(void* base, int64_t off) { return (char*)base + off; }
static inline const char* at(const void* base, int64_t off) { return (const char*)base + off; }

// ---------------------------------------------------------------- FiLM helpers
static void film_fill(const Plan& p, int k, const float* const* params, float* const* grads, void* ws, FilmGenDesc& g) {
  const BlockBuf& b = p.blk[k];
  g.w1 = params[b.film_pidx]; g.b1 = params[b.film_pidx + 1]; g.w2 = params[b.film_pidx + 2]; g.b2 = params[b.film_pidx + 3];
  if (grads) { g.dw1 = grads[b.film_pidx]; g.db1 = grads[b.film_pidx + 1]; g.dw2 = grads[b.film_pidx + 2]; g.db2 = grads[b.film_pidx + 3]; }
  else g.dw1 = g.db1 = g.dw2 = g.db2 = nullptr;
  g.gf = (float*)at(ws, b.gf); g.bf = (float*)at(ws, b.bf); g.pre = (float*)at(ws, b.pre); g.dpre = (float*)at(ws, b.dpre);
  const float* sums = (const float*)at(ws, b.sums2);
  g.dgf = sums + 2; g.dbf = sums + 3; g.dg_stride = 4;
  g.C = b.C;
}

static int film_run_forward(const Plan& p, int k0, int k1, const float* const* params, void* ws, const float* film_params, hipStream_t st,
                            float* zero = nullptr, int zero_n = 0) {
  FilmArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.zero = zero; fa.zero_n = zero_n;
  fa.ngen = k1 - k0; fa.B = p.d.B; fa.P = p.d.P; fa.Hf = p.d.film_hidden; fa.params = film_params;
  for (int k = k0; k < k1; ++k) film_fill(p, k, params, nullptr, ws, fa.g[k - k0]);
  return film_forward(fa, st);
}

static int film_run_backward(const Plan& p, int k0, int k1, const float* const* params, float* const* grads, void* ws,
                             const float* film_params, int flags, hipStream_t st, std::vector<GnGradDesc>* gn = nullptr,
                             std::vector<WgReduceDesc>* rd = nullptr) {
  FilmArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.ngen = k1 - k0; fa.B = p.d.B; fa.P = p.d.P; fa.Hf = p.d.film_hidden; fa.params = film_params;
  fa.dparams = (flags & SX_BWD_NO_DPARAMS) ? nullptr : (float*)at(ws, p.dparams);   // training does not need d(loss)/d(params)
  fa.accumulate = (flags & SX_BWD_ACCUMULATE) ? 1 : 0;
  const bool wg = (flags & SX_BWD_WEIGHT_GRADS) != 0;
  for (int k = k0; k < k1; ++k) film_fill(p, k, params, wg ? grads : nullptr, ws, fa.g[k - k0]);
  // gn: this half's GroupNorm affine / bias gradient reductions ride on the FiLM launch when it takes them (training steps)
  int done = 0, rdone = 0;
  SX_TRY(film_backward(fa, wg, st, gn ? gn->data() : nullptr, gn ? (int)gn->size() : 0, &done,
                       rd ? rd->data() : nullptr, rd ? (int)rd->size() : 0, &rdone));
  if (gn && done) gn->clear();
  if (rd && rdone) rd->clear();      // the slab reductions of this half rode along too
  return SX_OK;
}

// ---------------------------------------------------------------- weight-gradient queue
// The weight gradients of a backward half are independent of its data-gradient chain: every layer's split-K problem is
// only DESCRIBED when the chain passes it (its operands stay untouched in the workspace: nothing is aliased), and at the end of
// the half ONE launch runs all of them (wgrad_ws_launch: block ranges per problem, longest first) and ONE launch reduces all
// slabs.  23 weight-gradient launches per step of 7-45 us each -- ~8 us of ramp, first touch and store tail apiece -- become 2.
struct WgQueue {
  char* base; int64_t cap, used; int dtype, B, acc;
  hipStream_t st;
  std::vector<WgWsLaunch> launches;   // persistent-tier problems not launched yet
  std::vector<WgX3Launch> x3;         // fp32-tier problems not launched yet
  std::vector<WgReduceDesc> pending;  // slabs not reduced yet
  std::vector<GnGradDesc> gn;         // GroupNorm affine gradients, reduced over the batch in one launch at the end
  WgQueue(const Plan& p, void* ws, int acc_, hipStream_t st_)
      : base((char*)ws + p.wg_ws), cap(p.wg_ws_bytes), used(0), dtype(p.d.dtype), B(p.d.B), acc(acc_), st(st_) {}
  // launch and reduce everything pending, then the small per-channel reductions
  int finish() {
    SX_TRY(flush());
    return finish_gn();
  }
  int finish_gn() {
    if (!gn.empty()) SX_TRY(gn_param_grad_multi(gn.data(), (int)gn.size(), B, acc, st));
    gn.clear();
    return SX_OK;
  }
  void add_gn(const float* sums, float* dgamma, float* dbeta, int C) {
    GnGradDesc d;
    d.sums = sums; d.dgamma = dgamma; d.dbeta = dbeta; d.C = C; d.blk0 = 0; d.nb = 0; d.stride = 4;
    gn.push_back(d);
  }
  // bias gradient: channel sums accumulated in workspace scratch (zeroed with the backward sums), copied / added here
  void add_bias(const float* scratch, float* db, int C, int slots = 1) {
    GnGradDesc d;
    d.sums = scratch; d.dgamma = db; d.dbeta = nullptr; d.C = C; d.blk0 = 0; d.nb = slots; d.stride = 1;
    gn.push_back(d);
  }
  // reduce = false: the split-K launch only; the slab reductions stay in `pending` for a later launch to take (finish_reduce)
  int flush(bool reduce = true) {
    if (!launches.empty()) {
      // longest BLOCKS first (estimated block time: arithmetic at 4 TFLOP/s per CU + operand bytes at 60 GB/s per CU, over
      // the problem's block count): the short ones fill the tail.  SX_WG_SORT_BLOCK=0: by whole-problem cost (rounds 1-2;
      // cfg 5 6.54 -> 6.50 ms, cfg A 1.704 -> 1.698 ms, same box)
      std::stable_sort(launches.begin(), launches.end(), [](const WgWsLaunch& x, const WgWsLaunch& y) {
        static const int per_block = env_int("SX_WG_SORT_BLOCK", 1);
        auto cost = [](const WgWsLaunch& l) {
          const double c = (double)l.a.B * l.a.H * l.a.W * (l.a.CaPad + l.a.CbPad) * (l.variant >> 8 & 15);
          if (!per_block) return c;
          const double px = (double)l.a.B * l.a.H * l.a.W, taps = l.variant >> 8 & 15;
          return (2.0 * px * taps * l.a.CaPad * l.a.CbPad / 4e12 + 2.0 * px * (l.a.CaPad + l.a.CbPad) / 6e10) / std::max(1, l.blocks);
        };
        return cost(x) > cost(y);
      });
      SX_TRY(wgrad_ws_launch(launches.data(), (int)launches.size(), st));
      launches.clear();
    }
    if (!x3.empty()) {
      SX_TRY(wgrad_x3_launch(x3.data(), (int)x3.size(), st));
      x3.clear();
    }
    static const int report = env_int("SX_WG_REPORT", 0);        // diagnostic: slab bytes of this half (tools/wgrad_slabs.sh)
    if (report) {
      double bytes = 0;
      for (const WgReduceDesc& d : pending) bytes += (double)d.nsplit * (d.a16_n ? d.tap_n * d.a16_n * 16 : d.ntaps * d.CaPad) * d.CbPad * 4;
      fprintf(stderr, "[sx] weight-gradient slabs of this half: %zu problems, %.1f MB written + read\n", pending.size(), bytes / 1e6);
    }
    if (!reduce) return SX_OK;
    return finish_reduce();
  }
  int finish_reduce() {
    if (!pending.empty()) SX_TRY(wgrad_reduce_multi(pending.data(), (int)pending.size(), st));
    pending.clear();
    used = 0;
    return SX_OK;
  }
  // conv1 (3x3) and the 1x1 shortcut of a 32-channel DoubleConv in ONE problem: P = the [dY1 | dR] buffer, 64 channels = one
  // full 128-byte line per pixel (each half on its own is half-line reads of the same lines), the shortcut rows multiply the
  // centre tap only; two descriptors reduce the two row ranges of the slab.  Returns 1 when the shape is not covered.
  int add_dual(int H, int W, int C, int Cb, int Cb_out, const void* P, const void* Q, int q_ld, float* dw3, float* dw1) {
    if (dtype != SX_BF16 || 2 * C != 64 || (Cb != 64 && Cb > 16)) return 1;
    const int64_t bytes = wgrad_ws_slab_bytes(SX_CONV3, B, H, W, 2 * C, Cb);
    if (bytes <= 0) return 1;
    const int64_t need = (int64_t)align_up((size_t)bytes, 256);
    if (need > cap) return 1;
    if (used + need > cap) SX_TRY(flush());
    WgReduceDesc d;
    memset(&d, 0, sizeof(d));
    WgWsLaunch l;
    const int rc = wgrad_ws_prepare(SX_CONV3, B, H, W, 2 * C, Cb, P, 2 * C, Q, q_ld, H, W, (float*)(base + used), &d, &l, 1);
    if (rc != 0) return rc;
    launches.push_back(l);
    d.Ca = C; d.Cb = Cb_out; d.accumulate = acc;
    WgReduceDesc d3 = d, d1 = d;
    d3.out = dw3; d3.a16_lo = 0; d3.a16_n = C / 16; d3.tap_lo = 0; d3.tap_n = 9;
    d1.out = dw1; d1.a16_lo = C / 16; d1.a16_n = C / 16; d1.tap_lo = 4; d1.tap_n = 1;
    pending.push_back(d3);
    pending.push_back(d1);
    used += need;
    return SX_OK;
  }
  int add(int mode, int H, int W, int Ca, int Cb, int Ca_out, int Cb_out, const void* P, int p_ld, const void* Q, int q_ld,
          int Hq, int Wq, float* dw) {
    const int64_t need = (int64_t)align_up((size_t)sx_conv_wgrad_workspace_bytes(mode, B, H, W, Ca, Cb), 256);
    if (used + need > cap) SX_TRY(flush());
    WgReduceDesc d;
    memset(&d, 0, sizeof(d));
    WgWsLaunch l;
    WgX3Launch lx;
    const int rc = wgrad_launch(dtype, mode, B, H, W, Ca, Cb, P, p_ld, Q, q_ld, Hq, Wq, base + used, cap - used, &d, st, &l, &lx);
    if (rc == WG_DEFERRED) launches.push_back(l);
    else if (rc == WG_DEFERRED_X3) x3.push_back(lx);
    else SX_TRY(rc);
    d.out = dw; d.Ca = Ca_out; d.Cb = Cb_out; d.accumulate = acc;
    pending.push_back(d);
    used += need;
    return SX_OK;
  }
};

// ---------------------------------------------------------------- one DoubleConv block
static int block_forward(const Plan& p, int k, const float* const* params, const void* wpack, void* ws, bool film,
                         const float* drop, hipStream_t st, void* pool = nullptr, int pool_ld = 0, bool* pooled = nullptr) {
  const BlockBuf& b = p.blk[k];
  const int dt = p.d.dtype, B = p.d.B, HW = b.H * b.W;
  const float* g1 = params[b.pidx + 1]; const float* be1 = params[b.pidx + 2];
  const float* g2 = params[b.pidx + 4]; const float* be2 = params[b.pidx + 5];
  // conv1 (+ GroupNorm statistics) and the 1x1 shortcut in one pass over the block input
  SX_TRY(sx_conv3_shortcut_forward(dt, B, b.H, b.W, b.Cin, b.C, at(ws, b.in), b.in_ld, at(wpack, b.w1f), at(wpack, b.wsf),
                                   at(ws, b.Y1), b.C, (float*)at(ws, b.st1), b.G, at(ws, b.S), b.C, st));
  SX_TRY(sx_gn_silu_forward(dt, B, HW, b.C, b.G, GN_EPS, at(ws, b.Y1), b.C, (const float*)at(ws, b.st1), g1, be1, at(ws, b.A1), b.C, st));
  SX_TRY(sx_conv_forward(dt, SX_CONV3, B, b.H, b.W, b.C, b.C, at(ws, b.A1), b.C, b.H, b.W, at(wpack, b.w2f), nullptr,
                         at(ws, b.Y2), b.C, b.H, b.W, 0, (float*)at(ws, b.st2), b.G, st));
  const bool keep_r = film || drop;
  if (pool && !(b.H & 1) && !(b.W & 1)) {      // encoder block: its max-pooled output in the same pass
    *pooled = true;
    // (the window positions are recorded for the backward pass, which then need not read the block output again)
    return sx_block_out_pool_idx_forward(dt, B, b.H, b.W, b.C, b.G, GN_EPS, at(ws, b.Y2), b.C, (const float*)at(ws, b.st2), g2, be2,
                                         at(ws, b.S), b.C, film ? (const float*)at(ws, b.gf) : nullptr,
                                         film ? (const float*)at(ws, b.bf) : nullptr, drop, at(ws, b.O), b.o_ld,
                                         nullptr, 0, pool, pool_ld, at(ws, b.pool_ix), b.C, st);
  }
  SX_TRY(sx_block_out_forward(dt, B, HW, b.C, b.G, GN_EPS, at(ws, b.Y2), b.C, (const float*)at(ws, b.st2), g2, be2, at(ws, b.S), b.C,
                              film ? (const float*)at(ws, b.gf) : nullptr, film ? (const float*)at(ws, b.bf) : nullptr, drop,
                              at(ws, b.O), b.o_ld, nullptr, 0, st));
  return SX_OK;
}

// d_out is read from b.dO.  Writes parameter gradients (if wg) and the input gradient into b.dX (if need_dx).
static int block_backward(const Plan& p, int k, const float* const* params, float* const* grads, const void* wpack, void* ws,
                          bool film, const float* drop, bool need_dx, int flags, WgQueue& wq, hipStream_t st,
                          float* ch_sum = nullptr, int ch_sum_n = 0, bool sums2_ready = false) {
  const BlockBuf& b = p.blk[k];
  const int dt = p.d.dtype, B = p.d.B, HW = b.H * b.W;
  const bool wg = (flags & SX_BWD_WEIGHT_GRADS) != 0;
  const int acc = (flags & SX_BWD_ACCUMULATE) ? 1 : 0;
  const float* g1 = params[b.pidx + 1]; const float* be1 = params[b.pidx + 2];
  const float* g2 = params[b.pidx + 4]; const float* be2 = params[b.pidx + 5];
  const bool keep_r = film || drop;
  // R = (SiLU(GN(y2)) + S) * drop is not stored by the forward pass: with FiLM or dropout the reduction rebuilds it from
  // the shortcut branch S (mode 2; one tensor write per block saved), without them R is the block output itself
  const void* Rsrc = keep_r ? at(ws, b.S) : at(ws, b.O);
  const int r_ld = keep_r ? b.C : b.o_ld;
  // through FiLM / dropout / residual / SiLU / GN2
  SX_TRY(sx_gn_silu_backward(dt, (keep_r ? 2 : 1) | (sums2_ready ? SX_GN_BWD_SUMS_READY : 0), B, HW, b.C, b.G, GN_EPS, at(ws, b.Y2), b.C, (const float*)at(ws, b.st2), g2, be2,
                             at(ws, b.dO), b.C, Rsrc, r_ld, film ? (const float*)at(ws, b.gf) : nullptr, drop,
                             (float*)at(ws, b.sums2), at(ws, b.dY), b.C, at(ws, b.dR), 2 * b.C,
                             nullptr, nullptr, acc, st));
  if (wg) wq.add_gn((const float*)at(ws, b.sums2), grads[b.pidx + 4], grads[b.pidx + 5], b.C);
  // dR = d(pre-FiLM sum) is always materialised next to dY1 (without FiLM / dropout it is a copy of dO): the joint
  // conv1 + shortcut data gradient reads both from one buffer
  const void* dR = at(ws, b.dR);
  const int dr_ld = 2 * b.C;
  if (wg) {
    SX_TRY(wq.add(SX_CONV3, b.H, b.W, b.C, b.C, b.C, b.C, at(ws, b.dY), b.C, at(ws, b.A1), b.C, b.H, b.W, grads[b.pidx + 3]));
  }
  // dA1 = W2^T (*) dY, with phase 1 of GroupNorm-1's backward (per-(sample, channel) sums) in the convolution's epilogue
  // (small levels: the sums are left to the one-launch backward below, `deferred`)
  int deferred = 0;
  SX_TRY(conv_forward_gn_bwd_sums(dt, SX_CONV3, B, b.H, b.W, b.C, b.C, at(ws, b.dY), b.C, at(wpack, b.w2d), at(ws, b.dA1), b.C,
                                  0, at(ws, b.Y1), b.C, nullptr, 0, (const float*)at(ws, b.st1), b.G, GN_EPS, g1, be1,
                                  nullptr, nullptr, (float*)at(ws, b.sums1), &deferred, st));
  // through SiLU / GN1, element-wise phase (its own buffer: the deferred weight-gradient launch still reads dY)
  SX_TRY(sx_gn_silu_backward(dt, deferred ? 0 : SX_GN_BWD_SUMS_READY, B, HW, b.C, b.G, GN_EPS, at(ws, b.Y1), b.C, (const float*)at(ws, b.st1), g1, be1,
                             at(ws, b.dA1), b.C, nullptr, 0, nullptr, nullptr, (float*)at(ws, b.sums1), at(ws, b.dY1), 2 * b.C,
                             nullptr, 0, nullptr, nullptr, acc, st));
  if (wg) wq.add_gn((const float*)at(ws, b.sums1), grads[b.pidx + 1], grads[b.pidx + 2], b.C);
  if (wg) {
    const int rc = wq.add_dual(b.H, b.W, b.C, b.Cin, b.Cin_real, at(ws, b.dY1), at(ws, b.in), b.in_ld, grads[b.pidx + 0], grads[b.pidx + 6]);
    if (rc < 0) return rc;
    if (rc == 1) {
      SX_TRY(wq.add(SX_CONV3, b.H, b.W, b.C, b.Cin, b.C, b.Cin_real, at(ws, b.dY1), 2 * b.C, at(ws, b.in), b.in_ld, b.H, b.W, grads[b.pidx + 0]));
      SX_TRY(wq.add(SX_CONV1, b.H, b.W, b.C, b.Cin, b.C, b.Cin_real, dR, dr_ld, at(ws, b.in), b.in_ld, b.H, b.W, grads[b.pidx + 6]));
    }
  }
  if (need_dx) {
    // dX = W1^T (*) dY1 + Ws^T dR in one pass over the [dY1 | dR] buffer
    SX_TRY(conv3_shortcut_dgrad_slots(dt, B, b.H, b.W, b.C, b.Cin, at(ws, b.dY1), 2 * b.C, at(wpack, b.w1d), at(wpack, b.wsd),
                                      at(ws, b.dX), b.dx_ld, ch_sum, ch_sum_n, BIAS_SLOTS, st));
  }
  return SX_OK;
}

static int check_tables(const Plan& p, const float* const* params, const void* wpack, void* ws) {
  if (!params || !wpack || !ws) { set_error("unet: null params / wpack / workspace"); return SX_ERR_INVALID; }
  for (int i = 0; i < p.nparams; ++i)
    if (!params[i]) { set_error("unet: parameter %d is null", i); return SX_ERR_INVALID; }
  if ((uintptr_t)ws % 256 || (uintptr_t)wpack % 256) { set_error("unet: workspace / wpack must be 256-byte aligned"); return SX_ERR_INVALID; }
  return SX_OK;
}

}  // namespace sx

using namespace sx;

extern "C" int sx_unet_num_params(const sx_unet_desc* d) {
  Plan p;
  if (build_plan(d, p) != SX_OK) return -1;
  return p.nparams;
}
extern "C" int64_t sx_unet_wpack_bytes(const sx_unet_desc* d) {
  Plan p;
  if (build_plan(d, p) != SX_OK) return -1;
  return p.wpack_bytes;
}
extern "C" int64_t sx_unet_workspace_bytes(const sx_unet_desc* d) {
  Plan p;
  if (build_plan(d, p) != SX_OK) return -1;
  return p.ws_bytes;
}

extern "C" int sx_unet_view(const sx_unet_desc* d, int which, int index, sx_view* out) {
  Plan p;
  SX_TRY(build_plan(d, p));
  SX_REQUIRE(out, "sx_unet_view: null output");
  const int D = p.depth;
  auto set = [&](int64_t off, int ld, int C, int l) { out->offset_bytes = off; out->ld = ld; out->C = C; out->H = p.H[l]; out->W = p.W[l]; };
  switch (which) {
    case SX_T_INPUT: set(p.X0, p.in_pad, d->in_ch, 0); break;
    case SX_T_SKIP:
      SX_REQUIRE(index >= 0 && index < D, "sx_unet_view: skip index %d", index);
      set(p.blk[index].O, 2 * p.C[index], p.C[index], index); break;
    case SX_T_BOTTLENECK: set(p.blk[D].O, p.C[D], p.C[D], D); break;
    case SX_T_PRED: set(p.pred, p.out_pad, d->out_ch, 0); break;
    case SX_T_DPRED: set(p.dpred, p.out_pad, d->out_ch, 0); break;
    case SX_T_DSKIP:
      SX_REQUIRE(index >= 0 && index < D, "sx_unet_view: dskip index %d", index);
      set(p.dcat[index] + (int64_t)p.C[index] * p.esz, 2 * p.C[index], p.C[index], index); break;
    case SX_T_DBOTTLENECK: set(p.blk[D].dO, p.C[D], p.C[D], D); break;
    case SX_T_DINPUT: set(p.blk[0].dX, p.in_pad, d->in_ch, 0); break;
    case SX_T_HEAD_BIAS_SUMS: out->offset_bytes = p.head_bias_scr; out->ld = d->out_ch; out->C = d->out_ch; out->H = 1; out->W = SX_HEAD_BIAS_SLOTS; break;
    case SX_T_FILM_DPARAMS: out->offset_bytes = p.dparams; out->ld = d->P; out->C = d->P; out->H = 1; out->W = 1; break;
    default: set_error("sx_unet_view: unknown tensor %d", which); return SX_ERR_INVALID;
  }
  return SX_OK;
}

// every packed weight of the network (forward / transposed pairs adjacent) + the head's zero-padded bias
static int pack_list(const Plan& p, const sx_unet_desc* d, const float* const* params, void* wpack, std::vector<PackDesc>& pk, PackPad& pad) {
  const int dt = d->dtype;
  auto add = [&](int mode, int tflip, const float* w, int Cout, int Cin, void* out) {
    PackDesc pd;
    SX_TRY(pack_desc_fill(dt, mode, tflip, w, Cout, Cin, out, &pd));
    pk.push_back(pd);
    return (int)SX_OK;
  };
  for (auto& b : p.blk) {
    SX_TRY(add(SX_CONV3, 0, params[b.pidx + 0], b.C, b.Cin_real, at(wpack, b.w1f)));
    SX_TRY(add(SX_CONV3, 1, params[b.pidx + 0], b.C, b.Cin_real, at(wpack, b.w1d)));
    SX_TRY(add(SX_CONV3, 0, params[b.pidx + 3], b.C, b.C, at(wpack, b.w2f)));
    SX_TRY(add(SX_CONV3, 1, params[b.pidx + 3], b.C, b.C, at(wpack, b.w2d)));
    SX_TRY(add(SX_CONV1, 0, params[b.pidx + 6], b.C, b.Cin_real, at(wpack, b.wsf)));
    SX_TRY(add(SX_CONV1, 1, params[b.pidx + 6], b.C, b.Cin_real, at(wpack, b.wsd)));
  }
  for (int j = 0; j < p.depth; ++j) {
    const int l = p.depth - 1 - j;
    // ConvTranspose2d weight is (Cin = C[l+1], Cout = C[l], 2, 2)
    SX_TRY(add(SX_CONVT_FWD, 0, params[p.up_pidx[j]], p.C[l], p.C[l + 1], at(wpack, p.upf[j])));
    SX_TRY(add(SX_CONVT_BWD, 0, params[p.up_pidx[j]], p.C[l], p.C[l + 1], at(wpack, p.upb[j])));
  }
  SX_TRY(add(SX_CONV1, 0, params[p.out_pidx], d->out_ch, p.C[0], at(wpack, p.headf)));
  SX_TRY(add(SX_CONV1, 1, params[p.out_pidx], d->out_ch, p.C[0], at(wpack, p.headd)));
  SX_REQUIRE(p.out_pad <= 64, "sx_unet_pack_weights: out_ch %d too large", d->out_ch);
  // the head's bias, zero padded, rides on the packing launch
  pad.src = params[p.out_pidx + 1]; pad.dst = (float*)at(wpack, p.head_bias); pad.n = d->out_ch; pad.npad = p.out_pad;
  return SX_OK;
}

extern "C" int sx_unet_pack_weights(const sx_unet_desc* d, const float* const* params, void* wpack, void* stream) {
  Plan p;
  SX_TRY(build_plan(d, p));
  Fp32WideScope fp32_scope(plan_wide_mode(p));
  SX_REQUIRE(params && wpack, "sx_unet_pack_weights: null pointer");
  std::vector<PackDesc> pk;
  PackPad pad;
  SX_TRY(pack_list(p, d, params, wpack, pk, pad));
  return pack_weights_multi(d->dtype, pk.data(), (int)pk.size(), (hipStream_t)stream, &pad);
}

// pack != 0: sx_unet_pack_weights first -- and, where the tables fit, in ONE launch with the input pack and the FiLM generators
static int encode_impl(const sx_unet_desc* d, const float* const* params, void* wpack, void* ws,
                       const float* x, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                       const float* film_params, void* stream, int pack) {
  Plan p;
  SX_TRY(build_plan(d, p));
  Fp32WideScope fp32_scope(plan_wide_mode(p));
  SX_TRY(check_tables(p, params, wpack, ws));
  SX_REQUIRE(x, "sx_unet_encode: null input");
  const bool film = d->use_film != 0;
  SX_REQUIRE(!film || film_params, "sx_unet_encode: FiLM requested without params");
  hipStream_t st = (hipStream_t)stream;
  const int dt = d->dtype, B = d->B, D = p.depth;
  // clear the encoder's GN statistics (blocks 0..D are laid out first).  use_film == 2 pairs every decode with the encode
  // before it: the decoder's statistics are cleared here too, and by the FiLM launch itself instead of a memset dispatch
  // ... and, laid out right behind them, the backward pass's sums and bias scratch (SX_BWD_SUMS_CLEARED: the backward of a
  // training step then has no memset dispatch either)
  const bool film_clears = film && d->use_film == 2;
  const int64_t zb0 = p.blk[0].st1, zb1 = d->use_film == 2 ? (film_clears ? p.bwd_zero_end : p.stats_end) : p.blk[D + 1].st1;
  if (!film_clears) {
    hipError_t e = hipMemsetAsync(at(ws, zb0), 0, (size_t)(zb1 - zb0), st);
    if (e != hipSuccess) return hip_fail(e, "memset(stats)");
  }
  // use_film == 2: encode and decode of this descriptor always see the same film_params, so every generator (the
  // decoder's too) is evaluated here in one launch and sx_unet_decode launches none
  const int ngen = film ? (d->use_film == 2 ? 2 * D + 1 : D + 1) : 0;
  bool head_done = false;
  if (pack) {
    std::vector<PackDesc> pk;
    PackPad pad;
    SX_TRY(pack_list(p, d, params, wpack, pk, pad));
    FilmArgs fa;
    memset(&fa, 0, sizeof(fa));
    if (ngen <= FILM_MAX_GEN) {
      fa.zero = film_clears ? (float*)at(ws, zb0) : nullptr; fa.zero_n = film_clears ? (int)((zb1 - zb0) / 4) : 0;
      fa.ngen = ngen; fa.B = B; fa.P = d->P; fa.Hf = d->film_hidden; fa.params = film_params;
      for (int k = 0; k < ngen; ++k) film_fill(p, k, params, nullptr, ws, fa.g[k]);
      const EncodeHeadExtra E = {fa, x, sb, sc, sh, sw, at(ws, p.X0), B, d->in_ch, d->H, d->W, p.in_pad};
      const int rc = encode_head_launch(dt, pk.data(), (int)pk.size(), &pad, E, st);
      if (rc < 0) return rc;
      head_done = rc == 0;
    }
    if (!head_done) SX_TRY(pack_weights_multi(dt, pk.data(), (int)pk.size(), st, &pad));
  }
  if (!head_done) {
    SX_TRY(sx_pack_input(dt, x, sb, sc, sh, sw, B, d->in_ch, d->H, d->W, at(ws, p.X0), p.in_pad, stream));
    if (film) SX_TRY(film_run_forward(p, 0, ngen, params, ws, film_params, st,
                                      film_clears ? (float*)at(ws, zb0) : nullptr, film_clears ? (int)((zb1 - zb0) / 4) : 0));
  }
  for (int k = 0; k <= D; ++k) {
    bool pooled = false;
    SX_TRY(block_forward(p, k, params, wpack, ws, film, nullptr, st, k < D ? at(ws, p.Pool[k]) : nullptr, k < D ? p.C[k] : 0, &pooled));
    if (k < D && !pooled)
      SX_TRY(sx_maxpool2_forward(dt, B, p.H[k], p.W[k], p.C[k], at(ws, p.blk[k].O), 2 * p.C[k], at(ws, p.Pool[k]), p.C[k], stream));
  }
  return SX_OK;
}

extern "C" int sx_unet_encode(const sx_unet_desc* d, const float* const* params, const void* wpack, void* ws,
                              const float* x, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                              const float* film_params, void* stream) {
  return encode_impl(d, params, const_cast<void*>(wpack), ws, x, sb, sc, sh, sw, film_params, stream, 0);
}

// sx_unet_pack_weights + sx_unet_encode in one call: the weight packs, the input pack and the FiLM generators share a launch
extern "C" int sx_unet_pack_encode(const sx_unet_desc* d, const float* const* params, void* wpack, void* ws,
                                   const float* x, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                                   const float* film_params, void* stream) {
  SX_REQUIRE(params && wpack, "sx_unet_pack_encode: null pointer");
  return encode_impl(d, params, wpack, ws, x, sb, sc, sh, sw, film_params, stream, 1);
}

extern "C" int sx_unet_decode(const sx_unet_desc* d, const float* const* params, const void* wpack, void* ws,
                              const float* film_params, const float* const* drop, void* stream) {
  Plan p;
  SX_TRY(build_plan(d, p));
  Fp32WideScope fp32_scope(plan_wide_mode(p));
  SX_TRY(check_tables(p, params, wpack, ws));
  const bool film = d->use_film != 0;
  SX_REQUIRE(!film || film_params, "sx_unet_decode: FiLM requested without params");
  hipStream_t st = (hipStream_t)stream;
  const int dt = d->dtype, B = d->B, D = p.depth;
  if (d->use_film != 2) {
    const int64_t b0 = p.blk[D + 1].st1;
    hipError_t e = hipMemsetAsync(at(ws, b0), 0, (size_t)(p.stats_end - b0), st);
    if (e != hipSuccess) return hip_fail(e, "memset(stats)");
  }
  if (film && d->use_film != 2) SX_TRY(film_run_forward(p, D + 1, 2 * D + 1, params, ws, film_params, st));
  for (int j = 0; j < D; ++j) {
    const int l = D - 1 - j, k = D + 1 + j;
    const BlockBuf& prev = p.blk[k - 1];   // bottleneck for j = 0, else decoder j-1
    // h = up(h): ConvTranspose2d into channels [0, C_l) of the cat buffer; zero the F.pad strip
    if (p.H[l] != 2 * p.H[l + 1] || p.W[l] != 2 * p.W[l + 1])
      SX_TRY(sx_zero_border(dt, B, p.H[l], p.W[l], 2 * p.H[l + 1], 2 * p.W[l + 1], p.C[l], at(ws, p.cat[l]), 2 * p.C[l], stream));
    SX_TRY(sx_conv_forward(dt, SX_CONVT_FWD, B, p.H[l + 1], p.W[l + 1], p.C[l + 1], p.C[l], at(ws, prev.O), prev.o_ld, p.H[l + 1], p.W[l + 1],
                           at(wpack, p.upf[j]), params[p.up_pidx[j] + 1], at(ws, p.cat[l]), 2 * p.C[l], p.H[l], p.W[l], 0, nullptr, 0, stream));
    SX_TRY(block_forward(p, k, params, wpack, ws, film, drop ? drop[j] : nullptr, st));
  }
  const BlockBuf& last = p.blk[2 * D];
  SX_TRY(sx_conv_forward(dt, SX_CONV1, B, d->H, d->W, p.C[0], p.out_pad, at(ws, last.O), last.o_ld, d->H, d->W, at(wpack, p.headf),
                         (const float*)at(wpack, p.head_bias), at(ws, p.pred), p.out_pad, d->H, d->W, 0, nullptr, 0, stream));
  return SX_OK;
}

extern "C" int sx_unet_decode_backward(const sx_unet_desc* d, const float* const* params, float* const* grads,
                                       const void* wpack, void* ws, const float* film_params,
                                       const float* const* drop, int flags, void* stream) {
  Plan p;
  SX_TRY(build_plan(d, p));
  Fp32WideScope fp32_scope(plan_wide_mode(p));
  SX_TRY(check_tables(p, params, wpack, ws));
  const bool wg = (flags & SX_BWD_WEIGHT_GRADS) != 0;
  const int acc = (flags & SX_BWD_ACCUMULATE) ? 1 : 0;
  SX_REQUIRE(!wg || grads, "sx_unet_decode_backward: weight gradients requested without a gradient table");
  const bool film = d->use_film != 0;
  hipStream_t st = (hipStream_t)stream;
  const int dt = d->dtype, B = d->B, D = p.depth;
  // SX_BWD_SUMS_CLEARED: the caller's promise that sx_unet_encode of THIS step (use_film == 2, FiLM on) cleared them
  const bool cleared = (flags & SX_BWD_SUMS_CLEARED) && d->use_film == 2 && film_params;
  {
    // decoder sums, bias scratch and (optionally) dparams are contiguous: one memset. SX_BWD_DEFER_FILM promises that
    // sx_unet_encode_backward follows on this workspace, so the encoder's sums (laid out just before) are cleared too
    const int64_t s0 = cleared ? p.bwd_zero_end : ((flags & SX_BWD_DEFER_FILM) ? p.blk[0].sums1 : p.blk[D + 1].sums1);
    const int64_t s1 = (flags & SX_BWD_ZERO_DPARAMS) ? p.dparams + (int64_t)align_up((size_t)B * std::max(1, d->P) * 4, 256) : p.bwd_zero_end;
    if (s1 > s0) {
      hipError_t e = hipMemsetAsync(at(ws, s0), 0, (size_t)(s1 - s0), st);
      if (e != hipSuccess) return hip_fail(e, "memset(sums)");
    }
  }
  WgQueue wq(p, ws, acc, st);
  // head
  const BlockBuf& last = p.blk[2 * D];
  if (wg) {
    SX_TRY(wq.add(SX_CONV1, d->H, d->W, p.out_pad, p.C[0], d->out_ch, p.C[0], at(ws, p.dpred), p.out_pad,
                  at(ws, last.O), last.o_ld, d->H, d->W, grads[p.out_pidx]));
    if (!(cleared && (flags & SX_BWD_HEAD_BIAS_READY)))
      SX_TRY(channel_sum_slots(dt, B, d->H, d->W, d->H, d->W, p.out_pad, d->out_ch, at(ws, p.dpred), p.out_pad, (float*)at(ws, p.head_bias_scr), SX_HEAD_BIAS_SLOTS, stream));
    wq.add_bias((const float*)at(ws, p.head_bias_scr), grads[p.out_pidx + 1], d->out_ch, SX_HEAD_BIAS_SLOTS);
  }
  SX_TRY(sx_conv_forward(dt, SX_CONV1, B, d->H, d->W, p.out_pad, p.C[0], at(ws, p.dpred), p.out_pad, d->H, d->W, at(wpack, p.headd),
                         nullptr, at(ws, last.dO), p.C[0], d->H, d->W, 0, nullptr, 0, stream));
  for (int j = D - 1; j >= 0; --j) {
    const int l = D - 1 - j, k = D + 1 + j;
    // the transposed conv's bias gradient = channel sums of the lower half of d(cat): they ride on the block's joint
    // data gradient unless an F.pad strip (odd sizes) has to be left out of the sum
    const int Hs = 2 * p.H[l + 1], Ws = 2 * p.W[l + 1];
    const bool ride = wg && Hs == p.H[l] && Ws == p.W[l];
    SX_TRY(block_backward(p, k, params, grads, wpack, ws, film, drop ? drop[j] : nullptr, true, flags, wq, st,
                          ride ? (float*)at(ws, p.bias_scr[j]) : nullptr, p.C[l]));
    const BlockBuf& prev = p.blk[k - 1];
    if (wg) {
      if (!ride)
        SX_TRY(channel_sum_slots(dt, B, p.H[l], p.W[l], Hs, Ws, p.C[l], p.C[l], at(ws, p.dcat[l]), 2 * p.C[l], (float*)at(ws, p.bias_scr[j]), BIAS_SLOTS, stream));
      wq.add_bias((const float*)at(ws, p.bias_scr[j]), grads[p.up_pidx[j] + 1], p.C[l], BIAS_SLOTS);
        SX_TRY(wq.add(SX_CONVT_FWD, p.H[l + 1], p.W[l + 1], p.C[l + 1], p.C[l], p.C[l + 1], p.C[l], at(ws, prev.O), prev.o_ld,
                    at(ws, p.dcat[l]), 2 * p.C[l], p.H[l], p.W[l], grads[p.up_pidx[j]]));
    }
    // gradient w.r.t. the transposed conv's input = d_out of the previous block
    SX_TRY(sx_conv_forward(dt, SX_CONVT_BWD, B, p.H[l + 1], p.W[l + 1], p.C[l], p.C[l + 1], at(ws, p.dcat[l]), 2 * p.C[l], p.H[l], p.W[l],
                           at(wpack, p.upb[j]), nullptr, at(ws, prev.dO), p.C[l + 1], p.H[l + 1], p.W[l + 1], 0, nullptr, 0, stream));
  }
  if (flags & SX_BWD_DEFER_FILM) wq.gn.clear();      // GroupNorm affine and bias gradients of this half: reduced by the encoder half's launch
  SX_TRY(wq.finish());
  // SX_BWD_DEFER_FILM: the decoder's FiLM generators are back-propagated by sx_unet_encode_backward together with the
  // encoder's (one launch pair per step instead of two; each pair is ~45 us of latency-bound work on a few CUs)
  if (film && !(flags & SX_BWD_DEFER_FILM)) SX_TRY(film_run_backward(p, D + 1, 2 * D + 1, params, grads, ws, film_params, flags, st));
  return SX_OK;
}

extern "C" int sx_unet_encode_backward(const sx_unet_desc* d, const float* const* params, float* const* grads,
                                       const void* wpack, void* ws, const float* film_params, int flags, void* stream) {
  Plan p;
  SX_TRY(build_plan(d, p));
  Fp32WideScope fp32_scope(plan_wide_mode(p));
  SX_TRY(check_tables(p, params, wpack, ws));
  const bool wg = (flags & SX_BWD_WEIGHT_GRADS) != 0;
  SX_REQUIRE(!wg || grads, "sx_unet_encode_backward: weight gradients requested without a gradient table");
  const bool film = d->use_film != 0;
  hipStream_t st = (hipStream_t)stream;
  const int dt = d->dtype, B = d->B, D = p.depth;
  // SX_BWD_ONLY_BOTTLENECK / SX_BWD_AFTER_BOTTLENECK split this call in two so that a data-parallel caller can start the
  // all-reduce of the bottleneck's convolution weights (40 % of all parameters) while the encoder levels are still being
  // back-propagated.  The first part ends with the bottleneck's slab reduction; its GroupNorm sums stay in the workspace
  // and are finished, with everything else, by the second part.
  const bool only_bot = (flags & SX_BWD_ONLY_BOTTLENECK) != 0, after_bot = (flags & SX_BWD_AFTER_BOTTLENECK) != 0;
  SX_REQUIRE(!(only_bot && after_bot), "sx_unet_encode_backward: ONLY_BOTTLENECK and AFTER_BOTTLENECK are two separate calls");
  if (!after_bot) {
    const int64_t s0 = p.blk[0].sums1, s1 = p.blk[D + 1].sums1;
    hipError_t e = hipSuccess;
    if (!(flags & SX_BWD_DEFER_FILM)) e = hipMemsetAsync(at(ws, s0), 0, (size_t)(s1 - s0), st);   // else: cleared by the decoder half
    if (e != hipSuccess) return hip_fail(e, "memset(sums)");
    if (flags & SX_BWD_ZERO_DPARAMS) {
      e = hipMemsetAsync(at(ws, p.dparams), 0, (size_t)B * std::max(1, d->P) * 4, st);
      if (e != hipSuccess) return hip_fail(e, "memset(dparams)");
    }
  }
  WgQueue wq(p, ws, (flags & SX_BWD_ACCUMULATE) ? 1 : 0, st);
  // bottleneck: d_out already in blk[D].dO ; skips' gradients in dcat[l] upper halves
  if (!after_bot) SX_TRY(block_backward(p, D, params, grads, wpack, ws, film, nullptr, true, flags, wq, st));
  if (only_bot) {
    wq.gn.clear();
    return wq.flush();
  }
  if (after_bot && wg) {
    const BlockBuf& b = p.blk[D];
    wq.add_gn((const float*)at(ws, b.sums2), grads[b.pidx + 4], grads[b.pidx + 5], b.C);
    wq.add_gn((const float*)at(ws, b.sums1), grads[b.pidx + 1], grads[b.pidx + 2], b.C);
  }
  for (int l = D - 1; l >= 0; --l) {
    const BlockBuf& b = p.blk[l];
    const BlockBuf& nxt = p.blk[l + 1];   // its dX is d(pool_l)
    if (!(p.H[l] & 1) && !(p.W[l] & 1))     // the forward pass pooled in its block-output launch and recorded the positions
      SX_TRY(sx_maxpool2_backward_add_idx(dt, B, p.H[l], p.W[l], p.C[l], at(ws, b.pool_ix), p.C[l], at(ws, nxt.dX), nxt.dx_ld,
                                          at(ws, p.dcat[l]) + (int64_t)p.C[l] * p.esz, 2 * p.C[l], at(ws, b.dO), p.C[l], stream));
    else
      SX_TRY(sx_maxpool2_backward_add(dt, B, p.H[l], p.W[l], p.C[l], at(ws, b.O), b.o_ld, at(ws, nxt.dX), nxt.dx_ld,
                                      at(ws, p.dcat[l]) + (int64_t)p.C[l] * p.esz, 2 * p.C[l], at(ws, b.dO), p.C[l], stream));
    SX_TRY(block_backward(p, l, params, grads, wpack, ws, film, nullptr, l > 0 || (flags & SX_BWD_INPUT_GRAD), flags, wq, st));
  }
  if (wg && (flags & SX_BWD_DEFER_FILM))
    for (int k = D + 1; k <= 2 * D; ++k) {       // decoder blocks: their per-sample sums are still in the workspace
      const BlockBuf& b = p.blk[k];
      wq.add_gn((const float*)at(ws, b.sums2), grads[b.pidx + 4], grads[b.pidx + 5], b.C);
      wq.add_gn((const float*)at(ws, b.sums1), grads[b.pidx + 1], grads[b.pidx + 2], b.C);
    }
  if (wg && (flags & SX_BWD_DEFER_FILM)) {       // ... and so are the head / transposed-conv bias sums
    wq.add_bias((const float*)at(ws, p.head_bias_scr), grads[p.out_pidx + 1], d->out_ch, SX_HEAD_BIAS_SLOTS);
    for (int j = 0; j < D; ++j) wq.add_bias((const float*)at(ws, p.bias_scr[j]), grads[p.up_pidx[j] + 1], p.C[D - 1 - j], BIAS_SLOTS);
  }
  const int ngen = (flags & SX_BWD_DEFER_FILM) ? 2 * D + 1 : D + 1;
  SX_TRY(wq.flush(!film));      // with FiLM: the slab reductions try to ride on the training tail's launch
  if (film) SX_TRY(film_run_backward(p, 0, ngen, params, grads, ws, film_params, flags, st, &wq.gn, &wq.pending));
  SX_TRY(wq.finish_reduce());   // whatever the FiLM launch did not take
  return wq.finish_gn();
}
