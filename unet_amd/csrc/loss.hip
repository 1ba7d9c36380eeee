// Mask-aware weighted loss, forward + gradient, on the device (losses.py:60-213), plus the masked
// metrics of the training loop (losses.py:226-253) and the normalizer maps (data.py:96-110,139-143,
// 157-163,187-223).
//
// The loss tensors are tiny next to the network's activations (out_ch = 8 channels against 32..256),
// so this is a short pipeline of simple bandwidth-bound kernels on planar fp32 copies rather than one
// monolith:
//   phase 1 (sx_loss_prepare): pooled masks / weight maps per scale and the raw denominators
//           sum(m), sum(m*w), sum(m_g) -> `den` (9 floats).  Under data parallelism the caller
//           all-reduces `den` between the phases: the reference's denominators are batch-global.
//   phase 2 (sx_loss_forward_backward): numerators, the scalar loss, and d(loss)/d(pred) written
//           back in the network's NHWC layout; also accumulates sum|p-t|*m for mae_norm.
// Reductions: per-thread partials -> wave shuffle -> one atomicAdd per block (fp32).
#include "sx_common.h"

namespace sx {

enum { PX_HUBER = 0, PX_L1 = 1, PX_MSE = 2 };

__device__ __forceinline__ float px_val(int kind, float d, float beta) {
  const float ad = fabsf(d);
  if (kind == PX_L1) return ad;
  if (kind == PX_MSE) return d * d;
  return ad < beta ? 0.5f * d * d / beta : ad - 0.5f * beta;
}
__device__ __forceinline__ float px_grad(int kind, float d, float beta) {
  if (kind == PX_L1) return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  if (kind == PX_MSE) return 2.f * d;
  return d < -beta ? -1.f : (d > beta ? 1.f : d / beta);
}

template <int NV>
__device__ __forceinline__ void block_atomic_add(float (&v)[NV], float* dst) {
  __shared__ float red[NV * 4];
  block_sum<NV>(v, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) atomicAdd(dst + i, v[i]);
  }
}

// The scalar numerators are accumulated into NSLOT copies (one 128-byte line each, chosen by block index) and
// summed by loss_final_kernel: 2048 blocks adding into ONE address serialise in the L2 (~13 ns per atomic, 90 us
// for the base term alone), 32 adders per line do not.
static constexpr int NSLOT = 64, SLOT_STRIDE = 32;
__device__ __forceinline__ float* slot_of(float* num) { return num + (blockIdx.x & (NSLOT - 1)) * SLOT_STRIDE; }

#define GRID_STRIDE(i, n) for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// NHWC T (ld) -> planar fp32 [B][C][H][W]
template <typename T>
__global__ void to_planar_kernel(const T* __restrict__ in, int ld, float* __restrict__ out, int B, int C, int H, int W) {
  const int64_t n = (int64_t)B * C * H * W;
  GRID_STRIDE(i, n) {
    int64_t p = i;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H); p /= H;
    const int c = (int)(p % C);
    const int b = (int)(p / C);
    out[i] = to_f32(in[(((int64_t)b * H + y) * W + x) * ld + c]);
  }
}

// planar fp32 -> NHWC T (ld, channels >= C zero-filled), scaled
template <typename T>
__global__ void from_planar_kernel(const float* __restrict__ in, T* __restrict__ out, int ld, int B, int C, int H, int W, float scale) {
  const int64_t n = (int64_t)B * H * W * ld;
  GRID_STRIDE(i, n) {
    const int c = (int)(i % ld);
    int64_t p = i / ld;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    out[i] = from_f32<T>(c < C ? scale * in[(((int64_t)b * C + c) * H + y) * W + x] : 0.f);
  }
}

// k x k average pooling (stride k, floor) of `planes` planes; optional threshold (> 0.5 -> 1 else 0)
__global__ void avgpool_planar_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t planes, int h, int w, int k, int thresh) {
  const int ho = h / k, wo = w / k;
  const int64_t n = planes * ho * wo;
  const float inv = 1.0f / (float)(k * k);
  GRID_STRIDE(i, n) {
    int64_t p = i;
    const int x = (int)(p % wo); p /= wo;
    const int y = (int)(p % ho);
    const int64_t pl = p / ho;
    const float* src = in + (pl * h + (int64_t)y * k) * w + (int64_t)x * k;
    float s = 0.f;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) s += src[dy * w + dx];
    s *= inv;
    out[i] = thresh ? (s > 0.5f ? 1.f : 0.f) : s;
  }
}

// den[0] += sum m ; den[1] += sum m*w   over (b, pixels)
__global__ __launch_bounds__(256) void den_kernel(const float* __restrict__ m, const float* __restrict__ w, int B, int hw, float* den, int slot_m, int slot_w) {
  float v[2] = {0.f, 0.f};
  // grid.y = sample, grid.x = pixel chunks: no division per element, few blocks (the two sums end in atomics)
  const float* mb = m + (int64_t)blockIdx.y * hw;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
    const float mm = mb[i];
    v[0] += mm;
    if (w) v[1] += mm * w[i];
  }
  __shared__ float red[8];
  block_sum<2>(v, red);
  if (threadIdx.x == 0) {
    if (slot_m >= 0) atomicAdd(den + slot_m, v[0]);
    if (slot_w >= 0 && w) atomicAdd(den + slot_w, v[1]);
  }
}

// Single-scale fast path of phase 1 (multiscale = 0, the production setting): ONE block per sample forms
// {sum m, sum m*w, sum m_g} of its sample and the pooled + thresholded gradient-term mask, a second one-block kernel adds the
// per-sample partials in a fixed order into den (deterministic; replaces a memset, two atomic reductions and a pooling launch).
__global__ __launch_bounds__(256) void prepare1_kernel(const float* __restrict__ m, const float* __restrict__ w, int H, int W, int k,
                                                       int want_mg, float* __restrict__ mg, float* __restrict__ part, int rows_per) {
  // grid = (row chunks, samples); rows_per is a multiple of k, so pooling windows never straddle two blocks
  const int b = blockIdx.y, y0 = blockIdx.x * rows_per, y1 = min(H, y0 + rows_per);
  const float* mb = m + (int64_t)b * H * W;
  float v[3] = {0.f, 0.f, 0.f};
  for (int i = y0 * W + threadIdx.x; i < y1 * W; i += blockDim.x) {
    const float mm = mb[i];
    v[0] += mm;
    if (w) v[1] += mm * w[i];
  }
  if (want_mg) {
    if (k > 1) {
      const int gh = H / k, gw = W / k;
      const int g0 = y0 / k, g1 = min(gh, (y1 + k - 1) / k);
      const float inv = 1.0f / (float)(k * k);
      for (int i = g0 * gw + threadIdx.x; i < g1 * gw; i += blockDim.x) {
        const int y = i / gw, x = i - y * gw;
        const float* src = mb + (int64_t)y * k * W + x * k;
        float sm = 0.f;
        for (int dy = 0; dy < k; ++dy)
          for (int dx = 0; dx < k; ++dx) sm += src[dy * W + dx];
        const float t = sm * inv > 0.5f ? 1.f : 0.f;
        mg[(int64_t)b * gh * gw + i] = t;
        v[2] += t;
      }
    } else {
      v[2] = v[0];          // grad_ds = 1: the gradient term uses the mask itself (no threshold)
    }
  }
  __shared__ float red[12];
  block_sum<3>(v, red);
  if (threadIdx.x == 0) {
    float* o = part + ((int64_t)b * gridDim.x + blockIdx.x) * 4;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
  }
}

// den[0..2] = sums of the n partial triples (fixed order: thread t adds entries t, t + 256, ...; then a tree); den[3..8] = 0
__global__ __launch_bounds__(256) void den_reduce_kernel(const float* __restrict__ part, int n, float* __restrict__ den, float* __restrict__ num) {
  for (int i = threadIdx.x; i < NSLOT * SLOT_STRIDE; i += blockDim.x) num[i] = 0.f;   // phase 2's numerator slots start at zero
  float v[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < n; i += blockDim.x) { v[0] += part[i * 4]; v[1] += part[i * 4 + 1]; v[2] += part[i * 4 + 2]; }
  __shared__ float red[12];
  block_sum<3>(v, red);
  if (threadIdx.x == 0) {
    den[0] = v[0]; den[1] = v[1]; den[2] = v[2];
    // this rank's own pixel count, kept where no all-reduce of `den` reaches it (the unused fourth float of partial 0):
    // phase 2 reports it as out[3], so a data-parallel caller need not copy den[0] aside first
    const_cast<float*>(part)[3] = v[0];
  }
  if (threadIdx.x >= 3 && threadIdx.x < 9) den[threadIdx.x] = 0.f;
}

struct BaseArgs {
  const float* p; const float* t; const float* m; const float* w; const float* cw;
  float* g;           // planar gradient at this scale (written, not accumulated)
  float* num;         // num[0] += sum px*m*cw ; num[1] += sum px*m*w*cw ; num[2] += sum |p-t|*m
  int B, C, hw;
  int kind; float beta;
  const float* den;   // den + 3*scale : {sum m, sum m*w, sum m_g}
  float coef, lam_w;  // scale coefficient (1, ms_weight, ms_weight/2) and edge weight
};

__device__ __forceinline__ float cw_sum(const float* cw, int C) {
  if (!cw) return (float)C;
  float s = 0.f;
  if (C <= 16) {            // independent guarded loads, one memory latency (a loop of dependent trips pays one per channel)
    float t[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) t[c] = c < C ? cw[c] : 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += t[c];
    return s;
  }
  for (int c = 0; c < C; ++c) s += cw[c];
  return s;
}

__global__ __launch_bounds__(256) void base_kernel(BaseArgs a) {
  float v[3] = {0.f, 0.f, 0.f};
  const float scw = cw_sum(a.cw, a.C);
  const float cb = a.coef / fmaxf(a.den[0] * scw, 1e-8f);
  const float ce = (a.w && a.lam_w != 0.f) ? a.coef * a.lam_w / fmaxf(a.den[1] * scw, 1e-8f) : 0.f;
  const int64_t n = (int64_t)a.B * a.C * a.hw;
  GRID_STRIDE(i, n) {
    const int px = (int)(i % a.hw);
    const int64_t bc = i / a.hw;
    const int c = (int)(bc % a.C);
    const int b = (int)(bc / a.C);
    const float mm = a.m[(int64_t)b * a.hw + px];
    const float ww = a.w ? a.w[px] : 0.f;
    const float cwc = a.cw ? a.cw[c] : 1.f;
    const float d = a.p[i] - a.t[i];
    const float l = px_val(a.kind, d, a.beta);
    v[0] += l * mm * cwc;
    v[1] += l * mm * ww * cwc;
    v[2] += fabsf(d) * mm;
    if (a.g) a.g[i] = px_grad(a.kind, d, a.beta) * mm * cwc * (cb + ce * ww);
  }
  __shared__ float red[12];
  block_sum<3>(v, red);
  if (threadIdx.x == 0) {
    float* num = slot_of(a.num);
    atomicAdd(num + 0, v[0]);
    atomicAdd(num + 1, v[1]);
    atomicAdd(num + 2, v[2]);
  }
}

__device__ __forceinline__ float at0(const float* pl, int h, int w, int y, int x) {
  return (y >= 0 && y < h && x >= 0 && x < w) ? pl[y * w + x] : 0.f;
}
// 3x3 Sobel cross-correlation, zero padding (losses.py:27-39,55-56)
__device__ __forceinline__ void sobel_at(const float* pl, int h, int w, int y, int x, float& gx, float& gy) {
  const float a00 = at0(pl, h, w, y - 1, x - 1), a01 = at0(pl, h, w, y - 1, x), a02 = at0(pl, h, w, y - 1, x + 1);
  const float a10 = at0(pl, h, w, y, x - 1), a12 = at0(pl, h, w, y, x + 1);
  const float a20 = at0(pl, h, w, y + 1, x - 1), a21 = at0(pl, h, w, y + 1, x), a22 = at0(pl, h, w, y + 1, x + 1);
  gx = (a02 - a00) + 2.f * (a12 - a10) + (a22 - a20);
  gy = (a20 - a00) + 2.f * (a21 - a01) + (a22 - a02);
}

struct SobelArgs {
  const float* p; const float* t; const float* m; const float* cw;   // pooled planes, thresholded mask [B][h][w]
  float* ux; float* uy;     // d(term)/d(pgx), d(term)/d(pgy) per pooled pixel
  float* num;               // += sum grad_px * m * cw
  int B, C, h, w;
  int magnitude, kind; float beta;
  const float* den;         // den + 3*scale
  float cgl;                // coef * lam_grad
};

__global__ __launch_bounds__(256) void sobel_fwd_kernel(SobelArgs a) {
  float v[1] = {0.f};
  const float cg = a.cgl / fmaxf(a.den[2] * cw_sum(a.cw, a.C), 1e-8f);
  const int hw = a.h * a.w;
  const int64_t n = (int64_t)a.B * a.C * hw;
  GRID_STRIDE(i, n) {
    const int px = (int)(i % hw);
    const int64_t bc = i / hw;
    const int c = (int)(bc % a.C);
    const int b = (int)(bc / a.C);
    const int y = px / a.w, x = px - y * a.w;
    const float* pp = a.p + bc * hw;
    const float* tp = a.t + bc * hw;
    float pgx, pgy, tgx, tgy;
    sobel_at(pp, a.h, a.w, y, x, pgx, pgy);
    sobel_at(tp, a.h, a.w, y, x, tgx, tgy);
    const float mm = a.m[(int64_t)b * hw + px];
    const float cwc = a.cw ? a.cw[c] : 1.f;
    float l, dgx, dgy;
    if (a.magnitude) {
      const float P = sqrtf(pgx * pgx + pgy * pgy + 1e-12f), Tm = sqrtf(tgx * tgx + tgy * tgy + 1e-12f);
      l = px_val(a.kind, P - Tm, a.beta);
      const float gr = px_grad(a.kind, P - Tm, a.beta);
      dgx = gr * pgx / P;
      dgy = gr * pgy / P;
    } else {
      const float dx = pgx - tgx, dy = pgy - tgy;
      l = px_val(a.kind, dx, a.beta) + px_val(a.kind, dy, a.beta);
      dgx = px_grad(a.kind, dx, a.beta);
      dgy = px_grad(a.kind, dy, a.beta);
    }
    v[0] += l * mm * cwc;
    if (a.ux) {
      const float s = mm * cwc * cg;
      a.ux[i] = dgx * s;
      a.uy[i] = dgy * s;
    }
  }
  __shared__ float red[4];
  block_sum<1>(v, red);
  if (threadIdx.x == 0) atomicAdd(slot_of(a.num), v[0]);
}

// transposed Sobel of (ux, uy) -> gradient at pooled resolution, spread over the ds x ds window of
// the finer plane and ADDED to g (the finer-scale gradient): g[p] += dd[q(p)] / ds^2
__global__ void sobel_bwd_kernel(const float* __restrict__ ux, const float* __restrict__ uy, float* __restrict__ g,
                                 int64_t planes, int h, int w, int ds, int hf, int wf) {
  const int64_t n = planes * h * w;
  const float inv = 1.0f / (float)(ds * ds);
  GRID_STRIDE(i, n) {
    int64_t p = i;
    const int x = (int)(p % w); p /= w;
    const int y = (int)(p % h);
    const int64_t pl = p / h;
    const float* X = ux + pl * h * w;
    const float* Y = uy + pl * h * w;
    // gx(q) = sum_t kx[t] in(q + t - 1)  =>  d in(r) = sum_t kx[t] ux(r - t + 1)
    float d = 0.f;
    d += -1.f * at0(X, h, w, y + 1, x + 1) + 1.f * at0(X, h, w, y + 1, x - 1);
    d += -2.f * at0(X, h, w, y, x + 1) + 2.f * at0(X, h, w, y, x - 1);
    d += -1.f * at0(X, h, w, y - 1, x + 1) + 1.f * at0(X, h, w, y - 1, x - 1);
    d += -1.f * at0(Y, h, w, y + 1, x + 1) - 2.f * at0(Y, h, w, y + 1, x) - 1.f * at0(Y, h, w, y + 1, x - 1);
    d += 1.f * at0(Y, h, w, y - 1, x + 1) + 2.f * at0(Y, h, w, y - 1, x) + 1.f * at0(Y, h, w, y - 1, x - 1);
    d *= inv;
    float* dst = g + (pl * hf + (int64_t)y * ds) * wf + (int64_t)x * ds;
    for (int dy = 0; dy < ds; ++dy)
      for (int dx = 0; dx < ds; ++dx) dst[dy * wf + dx] += d;
  }
}

// ---------------------------------------------------------------- single-scale fused path (multiscale = 0, grad_ds > 1)
// One thread per grad_ds x grad_ds pooling window and all C channels.
//   base_pool_kernel : base + edge terms (as base_kernel) straight from the NHWC prediction, the planar base gradient,
//                      and the pooled prediction / target planes of the gradient term  (replaces to_planar, base, 2 x avgpool)
//   grad_out_kernel  : transposed Sobel of (ux, uy) once per window, added to the base gradient of its pixels and written
//                      as the NHWC d_pred                                                (replaces sobel_bwd, from_planar)
static constexpr int FUSE_MAXC = 16;
struct FuseArgs {
  const void* pred; int ld;
  const float* t; const float* m; const float* w; const float* cw;
  float* g; float* pg; float* tg; float* num;
  const float* ux; const float* uy; void* dpred; int dld; float scale;
  int B, C, H, W, k;
  int kind; float beta, lam_w;
  const float* den;
  float* bias_sums;   // grad_out (fast instance) only, nullable: [SX_HEAD_BIAS_SLOTS][C] fp32, += per-channel sums of d_pred as stored
};

typedef __attribute__((ext_vector_type(2))) float f32x2_;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
__device__ __forceinline__ f32x2_ load2(const float* p) { return *reinterpret_cast<const f32x2_*>(p); }
__device__ __forceinline__ f32x2_ load2(const bf16* p) {
  const bf16x2_ v = *reinterpret_cast<const bf16x2_*>(p);
  return f32x2_{(float)v[0], (float)v[1]};
}
__device__ __forceinline__ void store2(float* p, f32x2_ v) { *reinterpret_cast<f32x2_*>(p) = v; }
__device__ __forceinline__ void store2(bf16* p, f32x2_ v) { *reinterpret_cast<bf16x2_*>(p) = bf16x2_{(bf16)v[0], (bf16)v[1]}; }
static constexpr int FUSE_CG = 4;        // threads per pooling window in the fast instances (8 / FUSE_CG channels each)

// 2 x 2 windows (grad_ds = 2, the production setting), C <= 8, even W: both pixels of a window row move as one 8-byte
// (fp32 planes) or two 16-byte (NHWC bf16) accesses.  FAST = false: any window size / channel count, scalar accesses.
// The fast instance gives a window to FUSE_CG threads, 8 / FUSE_CG channels each: one thread per window left the chip at two
// waves per SIMD with ~25 long-latency accesses each (base_pool 28.5 us for 55 MB; two threads 20.0, four 19.9 us; grad_out 18.3 / 14.9 / 13.7 us).
template <typename T, bool FAST>
__global__ __launch_bounds__(256) void base_pool_kernel(FuseArgs a) {
  constexpr int CG = FAST ? FUSE_CG : 1;
  constexpr int MC = FAST ? 8 / FUSE_CG : FUSE_MAXC;
  float v[3] = {0.f, 0.f, 0.f};
  const float scw = cw_sum(a.cw, a.C);
  const float cb = 1.0f / fmaxf(a.den[0] * scw, 1e-8f);
  const float ce = (a.w && a.lam_w != 0.f) ? a.lam_w / fmaxf(a.den[1] * scw, 1e-8f) : 0.f;
  const int k = FAST ? 2 : a.k;
  const int gh = a.H / k, gw = a.W / k, hw = a.H * a.W;
  const float inv = 1.0f / (float)(k * k);
  const T* pred = reinterpret_cast<const T*>(a.pred);
  const int64_t n = (int64_t)a.B * gh * gw;
  GRID_STRIDE(ii, n * CG) {
    const int cg = FAST ? (int)(ii / n) : 0;           // the channel group is the slow index: a wave stays inside one plane
    const int c0 = cg * MC;                            // this thread's first channel
    const int64_t i = ii - cg * n;
    float cwv[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) cwv[c] = (c0 + c < a.C) ? (a.cw ? a.cw[c0 + c] : 1.f) : 0.f;
    const int x = (int)(i % gw);
    const int y = (int)((i / gw) % gh);
    const int b = (int)(i / ((int64_t)gw * gh));
    float sp[MC], st[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) { sp[c] = 0.f; st[c] = 0.f; }
    if constexpr (FAST) {
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int px = (y * 2 + dy) * a.W + x * 2;
        const f32x2_ mm = *reinterpret_cast<const f32x2_*>(a.m + (int64_t)b * hw + px);
        f32x2_ ww = {0.f, 0.f};
        if (a.w) ww = *reinterpret_cast<const f32x2_*>(a.w + px);
        const T* pp = pred + ((int64_t)b * hw + px) * a.ld + c0;
        f32x4 p0, p1;
        if constexpr (MC == 4) { p0 = load4(pp); p1 = load4(pp + a.ld); }
        else { const f32x2_ q0 = load2(pp), q1 = load2(pp + a.ld); p0 = f32x4{q0[0], q0[1], 0.f, 0.f}; p1 = f32x4{q1[0], q1[1], 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < MC; ++c) {
          if (c0 + c < a.C) {
            const int64_t e = ((int64_t)b * a.C + c0 + c) * hw + px;
            const f32x2_ tt = *reinterpret_cast<const f32x2_*>(a.t + e);
            const float pr[2] = {p0[c], p1[c]};
            f32x2_ go;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
              const float d = pr[dx] - tt[dx];
              const float l = px_val(a.kind, d, a.beta);
              v[0] += l * mm[dx] * cwv[c];
              v[1] += l * mm[dx] * ww[dx] * cwv[c];
              v[2] += fabsf(d) * mm[dx];
              go[dx] = px_grad(a.kind, d, a.beta) * mm[dx] * cwv[c] * (cb + ce * ww[dx]);
              sp[c] += pr[dx];
              st[c] += tt[dx];
            }
            if (a.g) *reinterpret_cast<f32x2_*>(a.g + e) = go;
          }
        }
      }
    } else {
      for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) {
          const int px = (y * k + dy) * a.W + x * k + dx;
          const float mm = a.m[(int64_t)b * hw + px];
          const float ww = a.w ? a.w[px] : 0.f;
          const T* pp = pred + ((int64_t)b * hw + px) * a.ld;
#pragma unroll
          for (int c = 0; c < MC; ++c) {
            if (c < a.C) {
              const int64_t e = ((int64_t)b * a.C + c) * hw + px;
              const float pr = to_f32(pp[c]), tt = a.t[e];
              const float d = pr - tt;
              const float l = px_val(a.kind, d, a.beta);
              v[0] += l * mm * cwv[c];
              v[1] += l * mm * ww * cwv[c];
              v[2] += fabsf(d) * mm;
              if (a.g) a.g[e] = px_grad(a.kind, d, a.beta) * mm * cwv[c] * (cb + ce * ww);
              sp[c] += pr;
              st[c] += tt;
            }
          }
        }
    }
#pragma unroll
    for (int c = 0; c < MC; ++c)
      if (c0 + c < a.C) {
        const int64_t q = (((int64_t)b * a.C + c0 + c) * gh + y) * gw + x;
        a.pg[q] = sp[c] * inv;
        a.tg[q] = st[c] * inv;
      }
  }
  __shared__ float red[12];
  block_sum<3>(v, red);
  if (threadIdx.x == 0) {
    float* num = slot_of(a.num);
    atomicAdd(num + 0, v[0]);
    atomicAdd(num + 1, v[1]);
    atomicAdd(num + 2, v[2]);
  }
}

struct FinalArgs {
  const float* den; float* num; const float* cw; int C;
  int nscales; float lam_w, lam_grad, ms_weight; int has_w;
  float* out;   // out[0] = loss, out[1] = sum|p-t|*m (scale 0), out[2] = out[3] = sum m (scale 0)
  const float* local_px;   // nullable: out[3] = *local_px (sum m as phase 1 formed it here, before any all-reduce of den)
};
// One block: total the numerator slots (fixed order), write out[0..3], and leave the slots ZERO for the next call (the
// single-scale path has no memset launch: sx_loss_prepare's reduction kernel zeroes them the first time).
// COHERENT: the slots were filled by atomics of THIS launch (other XCDs' included): read them where atomics are performed
template <bool COHERENT = false>
__device__ __forceinline__ void loss_final_block(const FinalArgs& a) {
  __shared__ float tot[16];
  if constexpr (COHERENT) {
    // one atomic read per thread (a loop over the 64 slots would be 64 dependent round trips), then the same fixed-order sum
    __shared__ float got[NSLOT][16];
    for (int i = threadIdx.x; i < NSLOT * 16; i += blockDim.x)
      got[i >> 4][i & 15] = atomicAdd(a.num + (i >> 4) * SLOT_STRIDE + (i & 15), 0.0f);
    __syncthreads();
    if (threadIdx.x < 16) {
      float t = 0.f;
      for (int sl = 0; sl < NSLOT; ++sl) t += got[sl][threadIdx.x];
      tot[threadIdx.x] = t;
    }
  } else if (threadIdx.x < 16) {
    float t = 0.f;
    for (int sl = 0; sl < NSLOT; ++sl) t += a.num[sl * SLOT_STRIDE + threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NSLOT * SLOT_STRIDE; i += blockDim.x) a.num[i] = 0.f;
  if (threadIdx.x) return;
  const float scw = cw_sum(a.cw, a.C);
  float loss = 0.f;
  for (int s = 0; s < a.nscales; ++s) {
    const float coef = s == 0 ? 1.f : (s == 1 ? a.ms_weight : 0.5f * a.ms_weight);
    float L = tot[s * 4 + 0] / fmaxf(a.den[s * 3 + 0] * scw, 1e-8f);
    if (a.has_w && a.lam_w != 0.f) L += a.lam_w * (tot[s * 4 + 1] / fmaxf(a.den[s * 3 + 1] * scw, 1e-8f));
    if (a.lam_grad != 0.f) L += a.lam_grad * (tot[s * 4 + 3] / fmaxf(a.den[s * 3 + 2] * scw, 1e-8f));
    loss += coef * L;
  }
  a.out[0] = loss;
  a.out[1] = tot[2];
  a.out[2] = a.den[0];
  a.out[3] = a.local_px ? *a.local_px : a.den[0];   // single scale: the local count; else a DP caller overwrites it
}
__global__ void loss_final_kernel(FinalArgs a) {
  if (blockIdx.x == 0) loss_final_block(a);
}

template <typename T, bool FAST>
__global__ __launch_bounds__(256) void grad_out_kernel(FuseArgs a, FinalArgs fin) {
  // the numerators are complete (base_pool and sobel_fwd ran before this launch): block 0 finishes the loss value here
  // instead of a one-block launch of its own
  if (blockIdx.x == 0) loss_final_block(fin);
  constexpr int CG = FAST ? FUSE_CG : 1;        // fast instance: FUSE_CG threads per window (as base_pool_kernel)
  constexpr int MC = FAST ? 8 / FUSE_CG : FUSE_MAXC;
  const int k = FAST ? 2 : a.k;
  const int gh = a.H / k, gw = a.W / k, hw = a.H * a.W;
  const float inv = 1.0f / (float)(k * k);
  T* out = reinterpret_cast<T*>(a.dpred);
  const int64_t n = (int64_t)a.B * gh * gw;
  float bsum[MC];                 // head bias gradient: this thread's share (fast instance; the host checks n % 256 == 0 and one
  int bcg = 0;                    // item per thread at most, so a block works on one channel group)
#pragma unroll
  for (int c = 0; c < MC; ++c) bsum[c] = 0.f;
  GRID_STRIDE(ii, n * CG) {
    const int cg = FAST ? (int)(ii / n) : 0;
    const int c0 = cg * MC;
    const int64_t i = ii - cg * n;
    const int x = (int)(i % gw);
    const int y = (int)((i / gw) % gh);
    const int b = (int)(i / ((int64_t)gw * gh));
    float d[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
      d[c] = 0.f;
      if (c0 + c < a.C) {
        const float* X = a.ux + ((int64_t)b * a.C + c0 + c) * gh * gw;
        const float* Y = a.uy + ((int64_t)b * a.C + c0 + c) * gh * gw;
        float t = 0.f;
        t += -1.f * at0(X, gh, gw, y + 1, x + 1) + 1.f * at0(X, gh, gw, y + 1, x - 1);
        t += -2.f * at0(X, gh, gw, y, x + 1) + 2.f * at0(X, gh, gw, y, x - 1);
        t += -1.f * at0(X, gh, gw, y - 1, x + 1) + 1.f * at0(X, gh, gw, y - 1, x - 1);
        t += -1.f * at0(Y, gh, gw, y + 1, x + 1) - 2.f * at0(Y, gh, gw, y + 1, x) - 1.f * at0(Y, gh, gw, y + 1, x - 1);
        t += 1.f * at0(Y, gh, gw, y - 1, x + 1) + 2.f * at0(Y, gh, gw, y - 1, x) + 1.f * at0(Y, gh, gw, y - 1, x - 1);
        d[c] = t * inv;
      }
    }
    if constexpr (FAST) {       // dld == 8: this thread's four channels of a pixel in one store, the plane gradient read as 8-byte pairs
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int px = (y * 2 + dy) * a.W + x * 2;
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
        for (int c = 0; c < MC; ++c) {
          f32x2_ gg = {0.f, 0.f};
          if (c0 + c < a.C) gg = *reinterpret_cast<const f32x2_*>(a.g + ((int64_t)b * a.C + c0 + c) * hw + px);
          o0[c] = c0 + c < a.C ? a.scale * (d[c] + gg[0]) : 0.f;
          o1[c] = c0 + c < a.C ? a.scale * (d[c] + gg[1]) : 0.f;
        }
        T* dst = out + ((int64_t)b * hw + px) * 8 + c0;
        if constexpr (MC == 4) { store4(dst, o0); store4(dst + 8, o1); }
        else { store2(dst, f32x2_{o0[0], o0[1]}); store2(dst + 8, f32x2_{o1[0], o1[1]}); }
#pragma unroll
        for (int c = 0; c < MC; ++c) bsum[c] += (float)(T)o0[c] + (float)(T)o1[c];      // the values as stored (what channel_sum would read)
        bcg = cg;
      }
    } else {
      for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) {
          const int px = (y * k + dy) * a.W + x * k + dx;
          T* dst = out + ((int64_t)b * hw + px) * a.dld;
          for (int c0 = 0; c0 < a.dld; c0 += 8) {          // dld % 8 == 0 (checked on the host): 16-byte stores, zero fill
            f32x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int c = c0 + e;
              float val = 0.f;
              if (c < a.C) {
#pragma unroll
                for (int cc = 0; cc < MC; ++cc) if (cc == c) val = d[cc];
                val = a.scale * (val + a.g[((int64_t)b * a.C + c) * hw + px]);
              }
              o.v[e] = val;
            }
            store8(dst + c0, o);
          }
        }
    }
  }
  if constexpr (FAST) {
    if (a.bias_sums) {          // wave butterfly -> LDS -> one atomic per channel and block, spread over SX_HEAD_BIAS_SLOTS lines
      __shared__ float bred[4][MC];
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        float v = bsum[c];
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((threadIdx.x & 63) == 0) bred[threadIdx.x >> 6][c] = v;
      }
      __syncthreads();
      const int cgb = __shfl(bcg, 0, 64);      // uniform per block (see above); blocks past the last item add zeros to group 0
      if (threadIdx.x < MC && cgb * MC + (int)threadIdx.x < a.C)
        atomicAdd(a.bias_sums + (blockIdx.x % SX_HEAD_BIAS_SLOTS) * a.C + cgb * MC + threadIdx.x,
                  (bred[0][threadIdx.x] + bred[1][threadIdx.x]) + (bred[2][threadIdx.x] + bred[3][threadIdx.x]));
    }
  }
}

// ---------------------------------------------------------------- the three fused kernels above as ONE tile kernel (training)
// Fast configuration only (2 x 2 windows, C <= 8 in 8-wide NHWC rows, even H, W, gradient wanted).  A block owns a tile of
// TGH x TGW pooling windows of one sample and four channels and never leaves the chip between the phases: the pooled planes
// of the tile + a halo of 2 and the Sobel-term gradients of the tile + a halo of 1 live in LDS, the base-term gradient of
// the thread's own windows in registers.  HBM sees pred, target, mask once (+ the halo windows, served by L2) and d_pred:
// ~40 MB instead of the ~140 MB of base_pool + sobel_fwd + grad_out (planar gradient, pooled planes, ux / uy round trips).
static constexpr int LT_GH = 16, LT_GW = 32;                // windows per tile
static constexpr int LT_EH1 = LT_GH + 4, LT_EW1 = LT_GW + 4, LT_P1 = LT_EW1 + 1;   // pooled planes: halo 2 (pitch padded)
static constexpr int LT_EH2 = LT_GH + 2, LT_EW2 = LT_GW + 2, LT_P2 = LT_EW2 + 1;   // ux / uy: halo 1
struct TileArgs {
  FuseArgs f;
  const float* mg;          // pooled + thresholded mask [B][gh][gw] (sx_loss_prepare)
  int magnitude, gkind;     // gradient term: compare magnitudes / components; its pixel loss kind
  float cgl;                // lam_grad
  int tiles_x, tiles_y;
  FinalArgs fin;            // ticket != 0: the last block to arrive totals the loss value (no one-block launch behind this one)
  int ticket;
};
static constexpr int TICKET_SLOT = 16;      // an unused word of numerator slot 0 (zeroed with the slots)

// Block = LT_NT threads for LT_GH x LT_GW windows x 4 channels: a thread owns ONE window x 2 channels (the first version,
// 256 threads with two windows x 4 channels each, took 37 us: ~3000 VALU instructions per thread on two waves per SIMD).
static constexpr int LT_NT = 1024;    // (8 x 32 tiles on 512 threads: 25.4 against 24.0 us)
template <typename T>
__global__ __launch_bounds__(LT_NT) void loss_tile_kernel(TileArgs A) {
  constexpr int MC = 4, TC = 2;                 // channels per block / per thread
  constexpr int NWIN = LT_GH * LT_GW;
  static_assert(NWIN * (MC / TC) == LT_NT, "one (window, channel pair) per thread");
  __shared__ float sP[MC][LT_EH1][LT_P1], sT[MC][LT_EH1][LT_P1];
  __shared__ float sUX[MC][LT_EH2][LT_P2], sUY[MC][LT_EH2][LT_P2];
  const FuseArgs& a = A.f;
  const int tile = blockIdx.x, b = blockIdx.y, c0 = blockIdx.z * MC;
  const int ty0 = (tile / A.tiles_x) * LT_GH, tx0 = (tile % A.tiles_x) * LT_GW;
  const int gh = a.H >> 1, gw = a.W >> 1, hw = a.H * a.W;
  const float scw = cw_sum(a.cw, a.C);
  const float cb = 1.0f / fmaxf(a.den[0] * scw, 1e-8f);
  const float ce = (a.w && a.lam_w != 0.f) ? a.lam_w / fmaxf(a.den[1] * scw, 1e-8f) : 0.f;
  const float cg = A.cgl / fmaxf(a.den[2] * scw, 1e-8f);
  const T* pred = reinterpret_cast<const T*>(a.pred);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  // ---- phase 1a: this thread's own window, two channels: base + edge terms, the base gradient (registers), pooled values
  const int o = threadIdx.x % NWIN, cs = (threadIdx.x / NWIN) * TC;      // window inside the tile, first channel inside the block
  const int oy = o / LT_GW, ox = o - oy * LT_GW;
  const int wy = ty0 + oy, wx = tx0 + ox;
  const bool own_in = wy < gh && wx < gw;
  float cwv[TC];
#pragma unroll
  for (int c = 0; c < TC; ++c) cwv[c] = (c0 + cs + c < a.C) ? (a.cw ? a.cw[c0 + cs + c] : 1.f) : 0.f;
  f32x2_ go[2][TC];
  {
    float sp[TC] = {0.f, 0.f}, st[TC] = {0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int c = 0; c < TC; ++c) go[dy][c] = f32x2_{0.f, 0.f};
    if (own_in && c0 + cs < a.C) {
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int px = (wy * 2 + dy) * a.W + wx * 2;
        const f32x2_ mm = *reinterpret_cast<const f32x2_*>(a.m + (int64_t)b * hw + px);
        f32x2_ ww = {0.f, 0.f};
        if (a.w) ww = *reinterpret_cast<const f32x2_*>(a.w + px);
        const T* pp = pred + ((int64_t)b * hw + px) * 8 + c0 + cs;
        const f32x2_ q0 = load2(pp), q1 = load2(pp + 8);
#pragma unroll
        for (int c = 0; c < TC; ++c) {
          if (c0 + cs + c < a.C) {
            const f32x2_ tt = *reinterpret_cast<const f32x2_*>(a.t + ((int64_t)b * a.C + c0 + cs + c) * hw + px);
            const float pr[2] = {q0[c], q1[c]};
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
              const float d = pr[dx] - tt[dx];
              const float l = px_val(a.kind, d, a.beta);
              v[0] += l * mm[dx] * cwv[c];
              v[1] += l * mm[dx] * ww[dx] * cwv[c];
              v[2] += fabsf(d) * mm[dx];
              go[dy][c][dx] = px_grad(a.kind, d, a.beta) * mm[dx] * cwv[c] * (cb + ce * ww[dx]);
              sp[c] += pr[dx];
              st[c] += tt[dx];
            }
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < TC; ++c) { sP[cs + c][oy + 2][ox + 2] = sp[c] * 0.25f; sT[cs + c][oy + 2][ox + 2] = st[c] * 0.25f; }
  }
  // ---- phase 1b: the halo ring of 2 windows: pooled values only (zero outside the image = the Sobel's zero padding)
  {
    constexpr int NRING = LT_EH1 * LT_EW1 - NWIN;      // 208 ring windows x 2 channel pairs <= LT_NT: one trip
    static_assert(NRING * (MC / TC) <= LT_NT, "halo ring in one trip");
    const int hI = threadIdx.x % NRING, hcs = (threadIdx.x / NRING) * TC;
    if (threadIdx.x < NRING * (MC / TC)) {
      // ring index -> (ey, ex): two full rows on top, two at the bottom, then 2 + 2 columns of the LT_GH middle rows
      int ey, ex;
      if (hI < 2 * LT_EW1) { ey = hI / LT_EW1; ex = hI - ey * LT_EW1; }
      else if (hI < 4 * LT_EW1) { const int q = hI - 2 * LT_EW1; ey = LT_EH1 - 2 + q / LT_EW1; ex = q % LT_EW1; }
      else { const int q = hI - 4 * LT_EW1; ey = 2 + (q >> 2); const int k4 = q & 3; ex = k4 < 2 ? k4 : LT_EW1 - 4 + k4; }
      const int y = ty0 - 2 + ey, x = tx0 - 2 + ex;
      float sp[TC] = {0.f, 0.f}, st[TC] = {0.f, 0.f};
      if (y >= 0 && y < gh && x >= 0 && x < gw && c0 + hcs < a.C) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const int px = (y * 2 + dy) * a.W + x * 2;
          const T* pp = pred + ((int64_t)b * hw + px) * 8 + c0 + hcs;
          const f32x2_ q0 = load2(pp), q1 = load2(pp + 8);
#pragma unroll
          for (int c = 0; c < TC; ++c)
            if (c0 + hcs + c < a.C) {
              const f32x2_ tt = *reinterpret_cast<const f32x2_*>(a.t + ((int64_t)b * a.C + c0 + hcs + c) * hw + px);
              sp[c] += q0[c] + q1[c];
              st[c] += tt[0] + tt[1];
            }
        }
      }
#pragma unroll
      for (int c = 0; c < TC; ++c) { sP[hcs + c][ey][ex] = sp[c] * 0.25f; sT[hcs + c][ey][ex] = st[c] * 0.25f; }
    }
  }
  __syncthreads();
  // ---- phase 2: Sobel term on the tile + a ring of 1: its value (own windows only) and gradient w.r.t. (pgx, pgy)
  for (int e = threadIdx.x; e < LT_EH2 * LT_EW2 * MC; e += LT_NT) {
    const int c = e / (LT_EH2 * LT_EW2), pos = e - c * (LT_EH2 * LT_EW2);
    const int ey = pos / LT_EW2, ex = pos - ey * LT_EW2;
    const int y = ty0 - 1 + ey, x = tx0 - 1 + ex;
    float ux = 0.f, uy = 0.f;
    if (y >= 0 && y < gh && x >= 0 && x < gw && c0 + c < a.C) {
      const bool own = ey >= 1 && ey <= LT_GH && ex >= 1 && ex <= LT_GW;
      const float mm = A.mg[((int64_t)b * gh + y) * gw + x];
      const float cwc = a.cw ? a.cw[c0 + c] : 1.f;
      const float (*P)[LT_P1] = sP[c];
      const float (*Q)[LT_P1] = sT[c];
      const int r = ey + 1, q = ex + 1;          // centre in halo-2 coordinates
      const float pgx = (P[r - 1][q + 1] - P[r - 1][q - 1]) + 2.f * (P[r][q + 1] - P[r][q - 1]) + (P[r + 1][q + 1] - P[r + 1][q - 1]);
      const float pgy = (P[r + 1][q - 1] - P[r - 1][q - 1]) + 2.f * (P[r + 1][q] - P[r - 1][q]) + (P[r + 1][q + 1] - P[r - 1][q + 1]);
      const float tgx = (Q[r - 1][q + 1] - Q[r - 1][q - 1]) + 2.f * (Q[r][q + 1] - Q[r][q - 1]) + (Q[r + 1][q + 1] - Q[r + 1][q - 1]);
      const float tgy = (Q[r + 1][q - 1] - Q[r - 1][q - 1]) + 2.f * (Q[r + 1][q] - Q[r - 1][q]) + (Q[r + 1][q + 1] - Q[r - 1][q + 1]);
      float l, dgx, dgy;
      if (A.magnitude) {
        const float Pm = sqrtf(pgx * pgx + pgy * pgy + 1e-12f), Tm = sqrtf(tgx * tgx + tgy * tgy + 1e-12f);
        l = px_val(A.gkind, Pm - Tm, a.beta);
        const float gr = px_grad(A.gkind, Pm - Tm, a.beta);
        dgx = gr * pgx / Pm;
        dgy = gr * pgy / Pm;
      } else {
        const float dx = pgx - tgx, dy = pgy - tgy;
        l = px_val(A.gkind, dx, a.beta) + px_val(A.gkind, dy, a.beta);
        dgx = px_grad(A.gkind, dx, a.beta);
        dgy = px_grad(A.gkind, dy, a.beta);
      }
      if (own) v[3] += l * mm * cwc;
      const float sc = mm * cwc * cg;
      ux = dgx * sc;
      uy = dgy * sc;
    }
    sUX[c][ey][ex] = ux;
    sUY[c][ey][ex] = uy;
  }
  __syncthreads();
  // ---- phase 3: transposed Sobel of the own window + the base gradient -> d_pred (NHWC), head bias sums
  T* out = reinterpret_cast<T*>(a.dpred);
  float bsum[TC] = {0.f, 0.f};
  if (own_in) {
    float d[TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      const float (*X)[LT_P2] = sUX[cs + c];
      const float (*Y)[LT_P2] = sUY[cs + c];
      const int r = oy + 1, q = ox + 1;
      float t = 0.f;
      t += -1.f * X[r + 1][q + 1] + 1.f * X[r + 1][q - 1];
      t += -2.f * X[r][q + 1] + 2.f * X[r][q - 1];
      t += -1.f * X[r - 1][q + 1] + 1.f * X[r - 1][q - 1];
      t += -1.f * Y[r + 1][q + 1] - 2.f * Y[r + 1][q] - 1.f * Y[r + 1][q - 1];
      t += 1.f * Y[r - 1][q + 1] + 2.f * Y[r - 1][q] + 1.f * Y[r - 1][q - 1];
      d[c] = t * 0.25f;
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int px = (wy * 2 + dy) * a.W + wx * 2;
      f32x2_ o0, o1;
#pragma unroll
      for (int c = 0; c < TC; ++c) {
        o0[c] = c0 + cs + c < a.C ? a.scale * (d[c] + go[dy][c][0]) : 0.f;
        o1[c] = c0 + cs + c < a.C ? a.scale * (d[c] + go[dy][c][1]) : 0.f;
      }
      T* dst = out + ((int64_t)b * hw + px) * 8 + c0 + cs;
      store2(dst, o0);
      store2(dst + 8, o1);
#pragma unroll
      for (int c = 0; c < TC; ++c) bsum[c] += (float)(T)o0[c] + (float)(T)o1[c];
    }
  }
  const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  __shared__ float red[4 * (LT_NT / 64)];
  block_sum<4>(v, red);
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    float back = 0.f;
    if (c0 < a.C) {
      float* num = a.num + (lin & (NSLOT - 1)) * SLOT_STRIDE;
      // returning atomics: their old values are in hand only once the additions have been performed
      back = atomicAdd(num + 0, v[0]) + atomicAdd(num + 1, v[1]) + atomicAdd(num + 2, v[2]) + atomicAdd(num + 3, v[3]);
    }
    int last = 0;
    if (A.ticket) {
      unsigned one = 1u;
      asm volatile("" : "+v"(one) : "v"(back));      // the ticket is drawn after those returns (no fence: nothing else is published)
      const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
      last = atomicAdd(reinterpret_cast<unsigned*>(a.num + TICKET_SLOT), one) == nblk - 1;
    }
    s_last = last;
  }
  if (a.bias_sums && c0 < a.C) {
    // waves 0..7 hold channel pair 0, waves 8..15 pair 1 (NWIN = 512 threads each)
    __shared__ float bred[LT_NT / 64][TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      float t = bsum[c];
      for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
      if ((threadIdx.x & 63) == 0) bred[threadIdx.x >> 6][c] = t;
    }
    __syncthreads();
    if (threadIdx.x < MC && c0 + (int)threadIdx.x < a.C) {
      const int pair = threadIdx.x / TC, c = threadIdx.x % TC;
      float t = 0.f;
      for (int w = 0; w < NWIN / 64; ++w) t += bred[pair * (NWIN / 64) + w][c];
      atomicAdd(a.bias_sums + (lin % SX_HEAD_BIAS_SLOTS) * a.C + c0 + threadIdx.x, t);
    }
  }
  __syncthreads();
  if (s_last) loss_final_block<true>(A.fin);      // every block has arrived: total the slots, write out[0..3], zero slots and ticket
}

// fine[p] += coarse[q(p)] / 4   (avg_pool2d(2) backward), floor semantics: trailing odd row/col untouched
__global__ void unpool2_add_kernel(const float* __restrict__ coarse, float* __restrict__ fine, int64_t planes, int h, int w, int hf, int wf) {
  const int64_t n = planes * h * w;
  GRID_STRIDE(i, n) {
    int64_t p = i;
    const int x = (int)(p % w); p /= w;
    const int y = (int)(p % h);
    const int64_t pl = p / h;
    const float d = 0.25f * coarse[i];
    float* dst = fine + (pl * hf + 2 * (int64_t)y) * wf + 2 * x;
    dst[0] += d; dst[1] += d; dst[wf] += d; dst[wf + 1] += d;
  }
}


// ---------------------------------------------------------------- workspace layout (floats)
struct LossLayout {
  int ns;                     // scales (1 + multiscale)
  int h[3], w[3];             // scale dims
  int gds[3], gh[3], gw[3];   // grad-term pooling factor and dims per scale
  int64_t P[3], T[3], M[3], Wm[3], G[3];     // planar copies per scale (P[0] = unpacked pred; T[0] = caller's target)
  int64_t PG[3], TG[3], MG[3], UX[3], UY[3]; // grad-term buffers per scale
  int64_t NUM;                // 12 floats
  int64_t PART;               // [B][4] per-sample denominators (single-scale phase 1)
  int64_t total;
};

static LossLayout loss_layout(int B, int C, int H, int W, int multiscale, int grad_ds) {
  LossLayout L;
  memset(&L, 0, sizeof(L));
  L.ns = 1 + std::max(0, std::min(multiscale, 2));
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t o = off; off += (n + 63) / 64 * 64; return o; };
  L.NUM = take(NSLOT * SLOT_STRIDE);
  L.PART = take((int64_t)B * 4 * 16);       // up to 16 row chunks per sample
  int gd = grad_ds < 1 ? 1 : grad_ds;
  for (int s = 0; s < L.ns; ++s) {
    L.h[s] = s == 0 ? H : L.h[s - 1] / 2;
    L.w[s] = s == 0 ? W : L.w[s - 1] / 2;
    L.gds[s] = s == 0 ? gd : std::max(1, gd >> s);   // max(1, grad_ds // 2), max(1, grad_ds // 4)
    L.gh[s] = L.h[s] / L.gds[s];
    L.gw[s] = L.w[s] / L.gds[s];
    const int64_t n = (int64_t)B * C * L.h[s] * L.w[s];
    const int64_t ng = (int64_t)B * C * L.gh[s] * L.gw[s];
    L.P[s] = take(n);
    L.T[s] = s == 0 ? -1 : take(n);
    L.M[s] = s == 0 ? -1 : take((int64_t)B * L.h[s] * L.w[s]);
    L.Wm[s] = s == 0 ? -1 : take((int64_t)L.h[s] * L.w[s]);
    L.G[s] = take(n);
    L.PG[s] = take(ng); L.TG[s] = take(ng); L.MG[s] = take((int64_t)B * L.gh[s] * L.gw[s]);
    L.UX[s] = take(ng); L.UY[s] = take(ng);
  }
  L.total = off;
  return L;
}

static inline int gsz(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 2048)); }

static int px_kind(int k, const char* who) {
  if (k < 0 || k > 2) { set_error("%s: unknown pixel loss kind %d", who, k); return -1; }
  return k;
}

}  // namespace sx

using namespace sx;

extern "C" int64_t sx_loss_workspace_bytes(int B, int C, int H, int W, int multiscale, int grad_ds) {
  return loss_layout(B, C, H, W, multiscale, grad_ds).total * (int64_t)sizeof(float);
}

// Phase 1.  mask (B,1,H,W) fp32, w (H,W) fp32 or NULL.  Writes den[9] = per scale {sum m, sum m*w, sum m_g}.
extern "C" int sx_loss_prepare(int B, int C, int H, int W, const float* mask, const float* w, int multiscale,
                               int grad_ds, float lam_grad, void* workspace, float* den, void* stream) {
  SX_REQUIRE(mask && workspace && den, "sx_loss_prepare: null pointer");
  hipStream_t st = (hipStream_t)stream;
  LossLayout L = loss_layout(B, C, H, W, multiscale, grad_ds);
  float* ws = (float*)workspace;
  if (L.ns == 1) {
    const bool want_mg = lam_grad != 0.f;
    if (want_mg && (L.gh[0] < 1 || L.gw[0] < 1)) { set_error("sx_loss_prepare: grid too small for grad_ds"); return SX_ERR_UNSUPPORTED; }
    const int k = L.gds[0];
    int chunks = std::max(1, std::min(16, H / std::max(8, k)));
    const int rows_per = cdiv(cdiv(H, chunks), k) * k;
    chunks = cdiv(H, rows_per);
    hipLaunchKernelGGL(prepare1_kernel, dim3(chunks, B), dim3(256), 0, st, mask, w, H, W, k, want_mg ? 1 : 0, ws + L.MG[0], ws + L.PART, rows_per);
    hipLaunchKernelGGL(den_reduce_kernel, dim3(1), dim3(256), 0, st, ws + L.PART, B * chunks, den, ws + L.NUM);
    SX_LAUNCH_CHECK("loss_prepare");
    return SX_OK;
  }
  hipError_t e = hipMemsetAsync(den, 0, 9 * sizeof(float), st);
  if (e != hipSuccess) return hip_fail(e, "memset(den)");
  for (int s = 0; s < L.ns; ++s) {
    if (L.h[s] < 1 || L.w[s] < 1) { set_error("sx_loss_prepare: grid %dx%d too small for multiscale=%d", H, W, multiscale); return SX_ERR_UNSUPPORTED; }
    const float* m_s = s == 0 ? mask : ws + L.M[s];
    const float* w_s = !w ? nullptr : (s == 0 ? w : ws + L.Wm[s]);
    if (s > 0) {
      const float* m_prev = s == 1 ? mask : ws + L.M[s - 1];
      hipLaunchKernelGGL(avgpool_planar_kernel, dim3(gsz((int64_t)B * L.h[s] * L.w[s])), dim3(256), 0, st, m_prev, ws + L.M[s], (int64_t)B, L.h[s - 1], L.w[s - 1], 2, 0);
      if (w) {
        const float* w_prev = s == 1 ? w : ws + L.Wm[s - 1];
        hipLaunchKernelGGL(avgpool_planar_kernel, dim3(gsz((int64_t)L.h[s] * L.w[s])), dim3(256), 0, st, w_prev, ws + L.Wm[s], (int64_t)1, L.h[s - 1], L.w[s - 1], 2, 0);
      }
    }
    hipLaunchKernelGGL(den_kernel, dim3(std::max(1, std::min(4, L.h[s] * L.w[s] / 1024)), B), dim3(256), 0, st, m_s, w_s, B, L.h[s] * L.w[s], den, s * 3 + 0, s * 3 + 1);
    if (lam_grad != 0.f) {
      if (L.gh[s] < 1 || L.gw[s] < 1) { set_error("sx_loss_prepare: grid too small for grad_ds"); return SX_ERR_UNSUPPORTED; }
      const float* mg = m_s;
      if (L.gds[s] > 1) {
        hipLaunchKernelGGL(avgpool_planar_kernel, dim3(gsz((int64_t)B * L.gh[s] * L.gw[s])), dim3(256), 0, st, m_s, ws + L.MG[s], (int64_t)B, L.h[s], L.w[s], L.gds[s], 1);
        mg = ws + L.MG[s];
      }
      hipLaunchKernelGGL(den_kernel, dim3(std::max(1, std::min(4, L.gh[s] * L.gw[s] / 1024)), B), dim3(256), 0, st, mg, (const float*)nullptr, B, L.gh[s] * L.gw[s], den, s * 3 + 2, -1);
    }
  }
  SX_LAUNCH_CHECK("loss_prepare");
  return SX_OK;
}

// Phase 2.  pred: NHWC `dtype` tensor (ld); target: planar fp32 (B,C,H,W); den from phase 1 (possibly
// all-reduced); same workspace as phase 1.  d_pred (NHWC dtype, ld) may be NULL (evaluation).
// out (4 floats): out[0] = loss, out[1] = sum|pred-target|*mask, out[2] = out[3] = sum mask.  grad_scale multiplies d_pred.
extern "C" int sx_loss_forward_backward(int dtype, int B, int C, int H, int W, const void* pred, int pred_ld,
                                        const float* target, const float* mask, const float* w, const float* cw,
                                        int base_kind, float huber_beta, float lam_w, float lam_grad, int grad_magnitude,
                                        int grad_kind, int multiscale, float ms_weight, int grad_ds, const float* den,
                                        void* workspace, float* out, void* d_pred, int dpred_ld, float grad_scale,
                                        void* stream) {
  return sx_loss_forward_backward_bias(dtype, B, C, H, W, pred, pred_ld, target, mask, w, cw, base_kind, huber_beta, lam_w, lam_grad,
                                       grad_magnitude, grad_kind, multiscale, ms_weight, grad_ds, den, workspace, out, d_pred,
                                       dpred_ld, grad_scale, nullptr, nullptr, stream);
}

extern "C" int sx_loss_forward_backward_bias(int dtype, int B, int C, int H, int W, const void* pred, int pred_ld,
                                             const float* target, const float* mask, const float* w, const float* cw,
                                             int base_kind, float huber_beta, float lam_w, float lam_grad, int grad_magnitude,
                                             int grad_kind, int multiscale, float ms_weight, int grad_ds, const float* den,
                                             void* workspace, float* out, void* d_pred, int dpred_ld, float grad_scale,
                                             float* bias_sums, int* bias_done, void* stream) {
  if (bias_done) *bias_done = 0;
  SX_REQUIRE(pred && target && mask && den && workspace && out, "sx_loss_forward_backward: null pointer");
  if (px_kind(base_kind, "sx_loss_forward_backward") < 0 || px_kind(grad_kind, "sx_loss_forward_backward") < 0) return SX_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  LossLayout L = loss_layout(B, C, H, W, multiscale, grad_ds);
  float* ws = (float*)workspace;
  float* num = ws + L.NUM;
  const bool want_grad = d_pred != nullptr;
  if (L.ns == 1 && lam_grad != 0.f && L.gds[0] > 1 && H % L.gds[0] == 0 && W % L.gds[0] == 0 && C <= FUSE_MAXC &&
      (!want_grad || dpred_ld % 8 == 0)) {
    const int gd = L.gds[0], gh = L.gh[0], gw = L.gw[0];
    FuseArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.pred = pred; fa.ld = pred_ld; fa.t = target; fa.m = mask; fa.w = w; fa.cw = cw;
    fa.g = want_grad ? ws + L.G[0] : nullptr; fa.pg = ws + L.PG[0]; fa.tg = ws + L.TG[0]; fa.num = num;
    fa.ux = ws + L.UX[0]; fa.uy = ws + L.UY[0]; fa.dpred = d_pred; fa.dld = dpred_ld; fa.scale = grad_scale;
    fa.B = B; fa.C = C; fa.H = H; fa.W = W; fa.k = gd; fa.kind = base_kind; fa.beta = huber_beta; fa.lam_w = lam_w; fa.den = den;
    const int64_t nwin = (int64_t)B * gh * gw, ng = nwin * C;
    // fast instance: 2 x 2 windows, <= 8 channels in 8-wide NHWC rows, 8-byte aligned plane rows
    const bool fast = gd == 2 && C <= 8 && pred_ld == 8 && (!want_grad || dpred_ld == 8) && W % 2 == 0 &&
                      ((uintptr_t)target % 8) == 0 && ((uintptr_t)mask % 8) == 0 && (!w || ((uintptr_t)w % 8) == 0) &&
                      ((uintptr_t)pred % 16) == 0 && (!want_grad || ((uintptr_t)d_pred % 16) == 0);
    FinalArgs f;
    f.den = den; f.num = num; f.cw = cw; f.C = C; f.nscales = 1; f.lam_w = lam_w; f.lam_grad = lam_grad;
    f.ms_weight = ms_weight; f.has_w = w != nullptr; f.out = out; f.local_px = ws + L.PART + 3;
#define SX_FUSE_LAUNCH(KERN, ...)                                                                               \
    do {                                                                                                        \
      if (dtype == SX_BF16) { if (fast) hipLaunchKernelGGL((KERN<bf16, true>), dim3(gsz(FUSE_CG * nwin)), dim3(256), 0, st, __VA_ARGS__);   \
                              else hipLaunchKernelGGL((KERN<bf16, false>), dim3(gsz(nwin)), dim3(256), 0, st, __VA_ARGS__); }     \
      else { if (fast) hipLaunchKernelGGL((KERN<float, true>), dim3(gsz(FUSE_CG * nwin)), dim3(256), 0, st, __VA_ARGS__);                   \
             else hipLaunchKernelGGL((KERN<float, false>), dim3(gsz(nwin)), dim3(256), 0, st, __VA_ARGS__); }                     \
    } while (0)
    static const int tile_on = env_int("SX_LOSS_TILE", 1);      // A/B knob: 0 = the three-kernel form
    if (tile_on && fast && want_grad) {
      TileArgs ta;
      ta.f = fa; ta.f.g = nullptr;
      ta.mg = ws + L.MG[0]; ta.magnitude = grad_magnitude; ta.gkind = grad_kind; ta.cgl = lam_grad;
      ta.tiles_x = cdiv(gw, LT_GW); ta.tiles_y = cdiv(gh, LT_GH);
      static const int ticket = env_int("SX_LOSS_TICKET", 1);     // A/B knob: 0 = a one-block launch totals the loss value
      ta.fin = f; ta.ticket = ticket;
      if (bias_sums) {
        ta.f.bias_sums = bias_sums;
        if (bias_done) *bias_done = 1;
      }
      const dim3 grid(ta.tiles_x * ta.tiles_y, B, 2);
      if (dtype == SX_BF16) hipLaunchKernelGGL(loss_tile_kernel<bf16>, grid, dim3(LT_NT), 0, st, ta);
      else hipLaunchKernelGGL(loss_tile_kernel<float>, grid, dim3(LT_NT), 0, st, ta);
      if (!ticket) hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, f);
      SX_LAUNCH_CHECK("loss_forward_backward");
      return SX_OK;
    }
    SX_FUSE_LAUNCH(base_pool_kernel, fa);
    SobelArgs sa;
    sa.p = ws + L.PG[0]; sa.t = ws + L.TG[0]; sa.m = ws + L.MG[0]; sa.cw = cw;
    sa.ux = want_grad ? ws + L.UX[0] : nullptr; sa.uy = want_grad ? ws + L.UY[0] : nullptr;
    sa.num = num + 3; sa.B = B; sa.C = C; sa.h = gh; sa.w = gw;
    sa.magnitude = grad_magnitude; sa.kind = grad_kind; sa.beta = huber_beta;
    sa.den = den; sa.cgl = lam_grad;
    hipLaunchKernelGGL(sobel_fwd_kernel, dim3(gsz(ng)), dim3(256), 0, st, sa);
    if (want_grad) {
      // head bias gradient on the way out (fast instance, block-uniform channel group, one item per thread at most)
      if (bias_sums && fast && nwin % 256 == 0 && FUSE_CG * nwin <= (int64_t)2048 * 256) {
        fa.bias_sums = bias_sums;
        if (bias_done) *bias_done = 1;
      }
      SX_FUSE_LAUNCH(grad_out_kernel, fa, f);          // block 0 also finishes the loss value
    } else {
      hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, f);
    }
#undef SX_FUSE_LAUNCH
    SX_LAUNCH_CHECK("loss_forward_backward");
    return SX_OK;
  }
  {
    hipError_t e = hipMemsetAsync(num, 0, NSLOT * SLOT_STRIDE * sizeof(float), st);
    if (e != hipSuccess) return hip_fail(e, "memset(num)");
    const int64_t n = (int64_t)B * C * H * W;
    if (dtype == SX_BF16) hipLaunchKernelGGL(to_planar_kernel<bf16>, dim3(gsz(n)), dim3(256), 0, st, (const bf16*)pred, pred_ld, ws + L.P[0], B, C, H, W);
    else hipLaunchKernelGGL(to_planar_kernel<float>, dim3(gsz(n)), dim3(256), 0, st, (const float*)pred, pred_ld, ws + L.P[0], B, C, H, W);
  }
  for (int s = 0; s < L.ns; ++s) {
    const int hs = L.h[s], wsz = L.w[s];
    const int64_t n = (int64_t)B * C * hs * wsz;
    const float coef = s == 0 ? 1.f : (s == 1 ? ms_weight : 0.5f * ms_weight);
    const float* t_s = s == 0 ? target : ws + L.T[s];
    const float* m_s = s == 0 ? mask : ws + L.M[s];
    const float* w_s = !w ? nullptr : (s == 0 ? w : ws + L.Wm[s]);
    if (s > 0) {
      hipLaunchKernelGGL(avgpool_planar_kernel, dim3(gsz(n)), dim3(256), 0, st, ws + L.P[s - 1], ws + L.P[s], (int64_t)B * C, L.h[s - 1], L.w[s - 1], 2, 0);
      const float* t_prev = s == 1 ? target : ws + L.T[s - 1];
      hipLaunchKernelGGL(avgpool_planar_kernel, dim3(gsz(n)), dim3(256), 0, st, t_prev, ws + L.T[s], (int64_t)B * C, L.h[s - 1], L.w[s - 1], 2, 0);
    }
    BaseArgs ba;
    ba.p = ws + L.P[s]; ba.t = t_s; ba.m = m_s; ba.w = w_s; ba.cw = cw;
    ba.g = want_grad ? ws + L.G[s] : nullptr;
    ba.num = num + s * 4; ba.B = B; ba.C = C; ba.hw = hs * wsz; ba.kind = base_kind; ba.beta = huber_beta;
    ba.den = den + s * 3; ba.coef = coef; ba.lam_w = lam_w;
    hipLaunchKernelGGL(base_kernel, dim3(gsz(n)), dim3(256), 0, st, ba);
    if (lam_grad != 0.f) {
      const int gd = L.gds[s], gh = L.gh[s], gw = L.gw[s];
      const int64_t ng = (int64_t)B * C * gh * gw;
      const float* pg = ws + L.P[s];
      const float* tg = t_s;
      const float* mg = m_s;
      if (gd > 1) {
        hipLaunchKernelGGL(avgpool_planar_kernel, dim3(gsz(ng)), dim3(256), 0, st, ws + L.P[s], ws + L.PG[s], (int64_t)B * C, hs, wsz, gd, 0);
        hipLaunchKernelGGL(avgpool_planar_kernel, dim3(gsz(ng)), dim3(256), 0, st, t_s, ws + L.TG[s], (int64_t)B * C, hs, wsz, gd, 0);
        pg = ws + L.PG[s]; tg = ws + L.TG[s]; mg = ws + L.MG[s];   // MG built (thresholded) in phase 1
      }
      SobelArgs sa;
      sa.p = pg; sa.t = tg; sa.m = mg; sa.cw = cw;
      sa.ux = want_grad ? ws + L.UX[s] : nullptr; sa.uy = want_grad ? ws + L.UY[s] : nullptr;
      sa.num = num + s * 4 + 3; sa.B = B; sa.C = C; sa.h = gh; sa.w = gw;
      sa.magnitude = grad_magnitude; sa.kind = grad_kind; sa.beta = huber_beta;
      sa.den = den + s * 3; sa.cgl = coef * lam_grad;
      hipLaunchKernelGGL(sobel_fwd_kernel, dim3(gsz(ng)), dim3(256), 0, st, sa);
      if (want_grad)
        hipLaunchKernelGGL(sobel_bwd_kernel, dim3(gsz(ng)), dim3(256), 0, st, ws + L.UX[s], ws + L.UY[s], ws + L.G[s], (int64_t)B * C, gh, gw, gd, hs, wsz);
    }
  }
  SX_LAUNCH_CHECK("loss scales");
  if (want_grad) {
    for (int s = L.ns - 1; s > 0; --s) {
      const int64_t n = (int64_t)B * C * L.h[s] * L.w[s];
      hipLaunchKernelGGL(unpool2_add_kernel, dim3(gsz(n)), dim3(256), 0, st, ws + L.G[s], ws + L.G[s - 1], (int64_t)B * C, L.h[s], L.w[s], L.h[s - 1], L.w[s - 1]);
    }
    const int64_t n = (int64_t)B * H * W * dpred_ld;
    if (dtype == SX_BF16) hipLaunchKernelGGL(from_planar_kernel<bf16>, dim3(gsz(n)), dim3(256), 0, st, ws + L.G[0], (bf16*)d_pred, dpred_ld, B, C, H, W, grad_scale);
    else hipLaunchKernelGGL(from_planar_kernel<float>, dim3(gsz(n)), dim3(256), 0, st, ws + L.G[0], (float*)d_pred, dpred_ld, B, C, H, W, grad_scale);
  }
  FinalArgs f;
  f.den = den; f.num = num; f.cw = cw; f.C = C; f.nscales = L.ns; f.lam_w = lam_w; f.lam_grad = lam_grad;
  f.ms_weight = ms_weight; f.has_w = w != nullptr; f.out = out; f.local_px = L.ns == 1 ? ws + L.PART + 3 : nullptr;
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, f);
  SX_LAUNCH_CHECK("loss_forward_backward");
  return SX_OK;
}
