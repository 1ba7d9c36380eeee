// Bandwidth-bound kernels between the convolutions of a DoubleConv block (models.py:61-80) and the
// FiLM modulation that follows it (models.py:152-156): GroupNorm apply + SiLU, the block output
//   O = ((SiLU(GN(Y2)) + S) * drop) * gf + bf,
// their backward passes (two-phase: per-(sample,channel) sums, then the elementwise gradient), and
// the tiny parameter-gradient reductions.  GroupNorm statistics arrive as fp32 [B][G][2] (sum,
// sum of squares) accumulated by the producing convolution's epilogue.
//
// Mapping: grid = (pixel chunks, B); a block works on ONE sample, so the per-channel scale/shift
// (and FiLM / dropout factors) are computed once into LDS.  Each thread owns a fixed 8-channel
// slot and walks pixels: 16-byte bf16 / 2x16-byte fp32 accesses, fully coalesced in NHWC.
#include "sx_common.h"
#include "norm_act.h"
#include <algorithm>

namespace sx {

static constexpr int MAXC = 1024;

struct ActArgs {
  // tensors (T, NHWC)
  const void* Y; int y_ld;        // raw conv output (pre-GN)
  const void* S; int s_ld;        // shortcut branch (block_out fwd)
  const void* R; int r_ld;        // saved (A2 + S) * drop  (bwd: FiLM gradient)
  const void* dO; int do_ld;      // incoming gradient
  void* out0; int o0_ld;          // fwd: A or O     ; bwd: dY
  void* out1; int o1_ld;          // fwd: R (opt)    ; bwd: dR (opt)
  // per-sample / per-channel fp32
  const float* stats;             // [B][G][2]
  const float* gamma; const float* beta;   // [C]
  const float* gf; const float* bf;        // [B][C] FiLM (1+gamma), beta   (nullable)
  const float* drop;                       // [B][C] dropout factors        (nullable)
  float* sums;                             // [B][C][4] backward sums
  int B, HW, C, G;
  float eps;
  int pix_per_block;
  int W; void* pool; int p_ld;    // fused 2x2 max-pool of out0 (act_fwd_pool_kernel)
  unsigned char* pidx; int pi_ld; // optional: position (0..3, first maximum) of each pooled value inside its window
};

// per-channel GN coefficients of sample b into LDS: n_hat = y * rs[c] - mr[c]; z = n_hat * ga[c] + be[c]
__device__ __forceinline__ void gn_coeffs(const ActArgs& a, int b, float* rs, float* mr) {
  const int cpg = a.C / a.G;
  const float inv_n = 1.0f / ((float)cpg * (float)a.HW);
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    const float* st = a.stats + ((int64_t)b * a.G + c / cpg) * 2;
    const float mean = st[0] * inv_n;
    const float var = fmaxf(st[1] * inv_n - mean * mean, 0.f);
    const float r = rsqrtf(var + a.eps);
    rs[c] = r;
    mr[c] = mean * r;
  }
}

// ---------------------------------------------------------------- forward
// MODE 0: out0 = silu(gn(Y))                      (first GN of the block)
// MODE 1: R = (silu(gn(Y)) + S) * drop ; out0 = R * gf + bf ; out1 = R (if non-null)
template <typename T, int MODE, int CMAX>
__global__ __launch_bounds__(256) void act_fwd_kernel(ActArgs a) {
  __shared__ float sc[CMAX], sh[CMAX];            // CMAX = 256 instances keep LDS small enough for 8 waves per SIMD
  __shared__ float fg[MODE ? CMAX : 1], fb[MODE ? CMAX : 1], dr[MODE ? CMAX : 1];
  const int b = blockIdx.y;
  gn_coeffs(a, b, sc, sh);
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    const float r = sc[c], m = sh[c], ga = a.gamma[c];
    sc[c] = r * ga;
    sh[c] = a.beta[c] - m * ga;
    if (MODE) {
      fg[c] = a.gf ? a.gf[(int64_t)b * a.C + c] : 1.f;
      fb[c] = a.bf ? a.bf[(int64_t)b * a.C + c] : 0.f;
      dr[c] = a.drop ? a.drop[(int64_t)b * a.C + c] : 1.f;
    }
  }
  __syncthreads();
  const int slots = a.C >> 3;
  // 32-bit item arithmetic inside the block (pix_per_block * slots is far below 2^31); 64-bit only for the pixel base
  const int p0 = blockIdx.x * a.pix_per_block;
  const int p1 = min(a.HW, p0 + a.pix_per_block);
  const unsigned n_items = (unsigned)(p1 - p0) * (unsigned)slots;
  const int64_t pbase = (int64_t)b * a.HW + p0;
  const T* Y = reinterpret_cast<const T*>(a.Y);
  const T* S = reinterpret_cast<const T*>(a.S);
  T* O0 = reinterpret_cast<T*>(a.out0);
  T* O1 = reinterpret_cast<T*>(a.out1);
  for (unsigned it = threadIdx.x; it < n_items; it += blockDim.x) {
    const unsigned pq = it / (unsigned)slots;
    const int64_t pix = pbase + pq;
    const int c0 = (int)(it - pq * (unsigned)slots) << 3;
    f32x8 y = load8(Y + pix * a.y_ld + c0);
    f32x8 o;
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o.v[e] = siluf_(y.v[e] * sc[c0 + e] + sh[c0 + e]);
      store8(O0 + pix * a.o0_ld + c0, o);
    } else {
      f32x8 s = load8(S + pix * a.s_ld + c0), r;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        r.v[e] = (siluf_(y.v[e] * sc[c0 + e] + sh[c0 + e]) + s.v[e]) * dr[c0 + e];
        o.v[e] = r.v[e] * fg[c0 + e] + fb[c0 + e];
      }
      store8(O0 + pix * a.o0_ld + c0, o);
      if (O1) store8(O1 + pix * a.o1_ld + c0, r);
    }
  }
}

// MODE 1 fused with nn.MaxPool2d(2) of its output (encoder blocks; H and W even): a thread owns one 2x2 window x 8
// channels, writes the four outputs (and R) and their maximum.  max(round(o)) == round(max(o)), so the pooled tensor is
// bit-identical to pooling the stored bf16 output.  pix_per_block counts windows here.
template <typename T, int CMAX>
__global__ __launch_bounds__(256) void act_fwd_pool_kernel(ActArgs a) {
  __shared__ float sc[CMAX], sh[CMAX], fg[CMAX], fb[CMAX], dr[CMAX];
  const int b = blockIdx.y;
  gn_coeffs(a, b, sc, sh);
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    const float r = sc[c], m = sh[c], ga = a.gamma[c];
    sc[c] = r * ga;
    sh[c] = a.beta[c] - m * ga;
    fg[c] = a.gf ? a.gf[(int64_t)b * a.C + c] : 1.f;
    fb[c] = a.bf ? a.bf[(int64_t)b * a.C + c] : 0.f;
    dr[c] = a.drop ? a.drop[(int64_t)b * a.C + c] : 1.f;
  }
  __syncthreads();
  const int slots = a.C >> 3, Wo = a.W >> 1, nwin = a.HW >> 2;
  const int w0 = blockIdx.x * a.pix_per_block;
  const int w1 = min(nwin, w0 + a.pix_per_block);
  const unsigned n_items = (unsigned)(w1 - w0) * (unsigned)slots;
  const T* Y = reinterpret_cast<const T*>(a.Y);
  const T* S = reinterpret_cast<const T*>(a.S);
  T* O0 = reinterpret_cast<T*>(a.out0);
  T* O1 = reinterpret_cast<T*>(a.out1);
  T* P = reinterpret_cast<T*>(a.pool);
  for (unsigned it = threadIdx.x; it < n_items; it += blockDim.x) {
    const unsigned wq = it / (unsigned)slots;
    const int c0 = (int)(it - wq * (unsigned)slots) << 3;
    const unsigned win = (unsigned)w0 + wq, wy = win / (unsigned)Wo, wx = win - wy * (unsigned)Wo;
    const int64_t pix0 = (int64_t)b * a.HW + (int64_t)(2 * wy) * a.W + 2 * wx;
    f32x8 y[4], sv[4], mx;
    float bestv[8];
    unsigned long long bestk = 0;      // eight 2-bit-in-a-byte positions
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t pix = pix0 + (k >> 1) * a.W + (k & 1);
      y[k] = load8(Y + pix * a.y_ld + c0);
      sv[k] = load8(S + pix * a.s_ld + c0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t pix = pix0 + (k >> 1) * a.W + (k & 1);
      f32x8 o, r;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        r.v[e] = (siluf_(y[k].v[e] * sc[c0 + e] + sh[c0 + e]) + sv[k].v[e]) * dr[c0 + e];
        o.v[e] = r.v[e] * fg[c0 + e] + fb[c0 + e];
        mx.v[e] = k ? fmaxf(mx.v[e], o.v[e]) : o.v[e];
        // the position is decided on the value as STORED (what the backward pass of a separate max-pool would compare),
        // first maximum wins, a NaN beats any number: ATen's rule (sx_maxpool2_backward_add applies the same one)
        const float q = (float)(T)o.v[e];
        if (k == 0) bestv[e] = q;
        else if (q > bestv[e] || (q != q && bestv[e] == bestv[e])) {
          bestv[e] = q;
          bestk = (bestk & ~(0xffull << (8 * e))) | ((unsigned long long)k << (8 * e));
        }
      }
      store8(O0 + pix * a.o0_ld + c0, o);
      if (O1) store8(O1 + pix * a.o1_ld + c0, r);
    }
    store8(P + ((int64_t)b * nwin + win) * a.p_ld + c0, mx);
    if (a.pidx) *reinterpret_cast<unsigned long long*>(a.pidx + ((int64_t)b * nwin + win) * a.pi_ld + c0) = bestk;
  }
}

// ---------------------------------------------------------------- backward, phase 1
// sums[b][c] += { sum dz*n_hat, sum dz, sum dO*R, sum dO }   (last two only for MODE 1)
//   dz = dO * gf * drop * silu'(z)
// A thread owns a 4-channel slot here (not 8): sixteen accumulators and sixteen loop-invariant coefficients keep
// the kernel near 64 VGPRs, and this read-only pass needs the waves (measured: 3 -> 4 waves per SIMD was -17 %).
// The per-channel factor gf*drop is applied to the finished sums, not per pixel.
template <typename T, int MODE, int CMAX>
__global__ __launch_bounds__(256) void act_bwd_reduce_kernel(ActArgs a) {
  constexpr int CH = 4;
  __shared__ float rs[CMAX], mr[CMAX], ga[CMAX], be[CMAX], fac[CMAX], drp[MODE == 2 ? CMAX : 1];
  __shared__ float red[MODE ? 4 : 2][256 * CH];
  const int b = blockIdx.y;
  gn_coeffs(a, b, rs, mr);
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    ga[c] = a.gamma[c];
    be[c] = a.beta[c];
    float f = 1.f;
    if (MODE) {
      const float dr = a.drop ? a.drop[(int64_t)b * a.C + c] : 1.f;
      if (a.gf) f *= a.gf[(int64_t)b * a.C + c];
      f *= dr;
      if (MODE == 2) drp[c] = dr;
    }
    fac[c] = f;
  }
  __syncthreads();
  const int NT = blockDim.x;
  const int slots = a.C / CH;
  const int lanes = NT / slots;                  // pixel lanes per block (slots <= MAXC / 4 = 256 <= NT)
  const int slot = threadIdx.x % slots, pl = threadIdx.x / slots;
  const int c0 = slot * CH;
  const int p0 = blockIdx.x * a.pix_per_block;
  const int p1 = min(a.HW, p0 + a.pix_per_block);
  const T* Y = reinterpret_cast<const T*>(a.Y);
  const T* R = reinterpret_cast<const T*>(a.R);
  const T* dO = reinterpret_cast<const T*>(a.dO);
  constexpr int NS = MODE ? 4 : 2;
  float acc[NS][CH];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[k][e] = 0.f;
  if (pl < lanes) {
    for (int p = p0 + pl; p < p1; p += lanes) {
      const int64_t pix = (int64_t)b * a.HW + p;
      f32x4 y = load4(Y + pix * a.y_ld + c0);
      f32x4 g = load4(dO + pix * a.do_ld + c0);
      f32x4 r;
      if (MODE) r = load4(R + pix * a.r_ld + c0);
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const float nh = y[e] * rs[c0 + e] - mr[c0 + e];
        const float z = nh * ga[c0 + e] + be[c0 + e];
        const float sg = sigmoidf_(z);
        const float dz = g[e] * (sg * (1.0f + z * (1.0f - sg)));     // g * silu'(z)
        acc[0][e] += dz * nh;
        acc[1][e] += dz;
        if (MODE) {
          // mode 2: the forward pass did not store R; rebuild it from the shortcut branch (z * sg = silu(z))
          const float rr = MODE == 2 ? (z * sg + r[e]) * drp[c0 + e] : r[e];
          acc[2][e] += g[e] * rr;
          acc[3][e] += g[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < CH; ++e) { acc[0][e] *= fac[c0 + e]; acc[1][e] *= fac[c0 + e]; }
  }
  // reduce over the pixel lanes that share a slot: red[k] = [lane][C] floats.  All NS sums go through LDS together (three
  // barriers in all; one sum at a time was four barriers per sum, and this tail is what a small launch consists of)
  const bool fold = a.C <= NT && NT % a.C == 0;   // then element j*NT + t belongs to channel t % C for every j
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int e = 0; e < CH; ++e) red[k][threadIdx.x * CH + e] = acc[k][e];
  __syncthreads();
  if (fold) {
    float s[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      s[k] = 0.f;
#pragma unroll
      for (int j = 0; j < CH; ++j) s[k] += red[k][j * NT + threadIdx.x];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NS; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    const int per = NT / a.C;                     // partials per channel
    for (int o = threadIdx.x; o < NS * a.C; o += NT) {
      const int k = o / a.C, c = o - k * a.C;
      float t = 0.f;
      for (int q = 0; q < per; ++q) t += red[k][c + q * a.C];
      atomicAdd(a.sums + ((int64_t)b * a.C + c) * 4 + k, t);
    }
  } else {
    for (int o = threadIdx.x; o < NS * a.C; o += NT) {
      const int k = o / a.C, c = o - k * a.C;
      float t = 0.f;
      for (int l = 0; l < lanes; ++l) t += red[k][l * a.C + c];
      atomicAdd(a.sums + ((int64_t)b * a.C + c) * 4 + k, t);
    }
  }
}

// ---------------------------------------------------------------- backward, both phases in ONE launch (small levels)
// A GroupNorm group of one sample is self-contained: the element-wise phase needs the group's two means, the per-channel sums
// stay inside the group.  Where HW x C/G elements fit the registers of one block (<= 8192: the 32x16 and 16x8 levels of
// cfg A), block (b, g) reads Y, dO (and S) ONCE, forms the sums (wave butterfly + one LDS exchange), stores them -- plain
// stores, this block owns (b, channels of g): no atomics, no zeroing needed -- and applies the gradient from registers.
// Replaces a reduce + apply pair (2 launches of 7-14 us on the dispatch floor, Y and dO read twice).
// Grid (B, G): block id & 7 = b & 7, so the groups of a sample share an XCD's L2 (they read the same 128-byte lines).
// A thread owns NSLOT 8-channel pieces with ONE piece index (256 % pieces-per-pixel == 0): its coefficients are registers.
template <typename T, int MODE, int NSLOT>
__global__ __launch_bounds__(256) void act_bwd_group_kernel(ActArgs a) {
  constexpr int NS = MODE ? 4 : 2;
  __shared__ float red[4][NS][64];       // [wave][sum][channel of the group]
  __shared__ float tot[NS][64];
  const int b = blockIdx.x, grp = blockIdx.y;
  const int cpg = a.C / a.G, ppp = cpg >> 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int piece = tid % ppp;
  const int c0 = grp * cpg + piece * 8;
  const float inv_n = 1.0f / ((float)cpg * (float)a.HW);
  const float* stp = a.stats + ((int64_t)b * a.G + grp) * 2;
  const float mean = stp[0] * inv_n;
  const float rs = rsqrtf(fmaxf(stp[1] * inv_n - mean * mean, 0.f) + a.eps), mr = mean * rs;
  float ga[8], be[8], fac[8], drp[MODE == 2 ? 8 : 1];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ga[e] = a.gamma[c0 + e];
    be[e] = a.beta[c0 + e];
    float f = 1.f;
    if (MODE) {
      const float dr = a.drop ? a.drop[(int64_t)b * a.C + c0 + e] : 1.f;
      if (a.gf) f *= a.gf[(int64_t)b * a.C + c0 + e];
      f *= dr;
      if (MODE == 2) drp[e] = dr;
    }
    fac[e] = f;
  }
  const int n_slots = a.HW * ppp;
  const T* Y = reinterpret_cast<const T*>(a.Y);
  const T* R = reinterpret_cast<const T*>(a.R);
  const T* dO = reinterpret_cast<const T*>(a.dO);
  f32x8 y[NSLOT], g[NSLOT], r[MODE ? NSLOT : 1];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const int q = tid + k * 256;
    if (q < n_slots) {
      const int64_t pix = (int64_t)b * a.HW + q / ppp;
      y[k] = load8(Y + pix * a.y_ld + c0);
      g[k] = load8(dO + pix * a.do_ld + c0);
      if (MODE) r[k] = load8(R + pix * a.r_ld + c0);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { y[k].v[e] = 0.f; g[k].v[e] = 0.f; if (MODE) r[k].v[e] = 0.f; }
    }
  }
  float acc[NS][8];
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
  // phase 1; afterwards g holds da = dO * gf * drop and (r in mode 1 / 2, g's partner otherwise) holds dz = da * silu'(z)
  f32x8 dzs[MODE ? 1 : NSLOT];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float nh = y[k].v[e] * rs - mr;
      const float z = nh * ga[e] + be[e];
      const float sg = sigmoidf_(z);
      const float gq = g[k].v[e];                 // zero outside the sample: contributes nothing
      const float dz = gq * (sg * (1.0f + z * (1.0f - sg)));
      acc[0][e] += dz * nh;
      acc[1][e] += dz;
      if (MODE) {
        const float rr = MODE == 2 ? (z * sg + r[k].v[e]) * drp[e] : r[k].v[e];
        acc[2][e] += gq * rr;
        acc[3][e] += gq;
        r[k].v[e] = dz * fac[e];
        g[k].v[e] = gq * fac[e];
      } else {
        dzs[k].v[e] = dz;
      }
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) { acc[0][e] *= fac[e]; acc[1][e] *= fac[e]; }
  // lanes with the same piece index: butterfly over the lane bits above log2(ppp)
  // (mask loop outside: the NS x 8 exchanges of one step are independent and pipeline; the other nesting ran 32 serial chains)
  for (int m = 32; m >= ppp; m >>= 1) {
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[i][e] += __shfl_xor(acc[i][e], m, 64);
  }
  if (lane < ppp) {
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave][i][piece * 8 + e] = acc[i][e];
  }
  __syncthreads();
  for (int o = tid; o < NS * cpg; o += 256) {
    const int i = o / cpg, c = o - i * cpg;
    const float t = (red[0][i][c] + red[1][i][c]) + (red[2][i][c] + red[3][i][c]);
    // gamma-weighted for the group means below (one parallel load per thread here; a global load inside the loop over the
    // group's channels would cost one memory latency per trip)
    tot[i][c] = i < 2 ? t * a.gamma[grp * cpg + c] : t;
    a.sums[((int64_t)b * a.C + grp * cpg + c) * 4 + i] = t;
  }
  __syncthreads();
  float m1 = 0.f, m2 = 0.f;
  for (int c = 0; c < cpg; ++c) {
    m1 += tot[1][c];
    m2 += tot[0][c];
  }
  m1 *= inv_n; m2 *= inv_n;
  T* dY = reinterpret_cast<T*>(a.out0);
  T* dR = reinterpret_cast<T*>(a.out1);
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const int q = tid + k * 256;
    if (q < n_slots) {
      const int64_t pix = (int64_t)b * a.HW + q / ppp;
      f32x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float nh = y[k].v[e] * rs - mr;
        const float dz = MODE ? r[k].v[e] : dzs[k].v[e];
        o.v[e] = rs * (ga[e] * dz - m1 - nh * m2);
      }
      store8(dY + pix * a.o0_ld + c0, o);
      if (MODE && dR) store8(dR + pix * a.o1_ld + c0, g[k]);
    }
  }
}

// ---------------------------------------------------------------- backward, phase 2
// dY = rstd * (gamma*dz - mean_g(gamma*dz) - n_hat * mean_g(gamma*dz*n_hat)) ; dR = dO*gf*drop (optional)
template <typename T, int MODE, int CMAX>
__global__ __launch_bounds__(256) void act_bwd_apply_kernel(ActArgs a) {
  __shared__ float rs[CMAX], mr[CMAX], ga[CMAX], be[CMAX], fac[CMAX], m1[CMAX], m2[CMAX], q1[CMAX], q2[CMAX];
  const int b = blockIdx.y;
  gn_coeffs(a, b, rs, mr);
  const int cpg = a.C / a.G;
  const float inv_n = 1.0f / ((float)cpg * (float)a.HW);
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    ga[c] = a.gamma[c];
    be[c] = a.beta[c];
    float f = 1.f;
    if (MODE) {
      if (a.gf) f *= a.gf[(int64_t)b * a.C + c];
      if (a.drop) f *= a.drop[(int64_t)b * a.C + c];
    }
    fac[c] = f;
    // gamma-weighted channel sums; their per-group totals are formed from LDS below (one L2 round trip per block
    // instead of one per channel of the group)
    const float4 sv = *reinterpret_cast<const float4*>(a.sums + ((int64_t)b * a.C + c) * 4);
    q1[c] = ga[c] * sv.y;
    q2[c] = ga[c] * sv.x;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    const int gbase = (c / cpg) * cpg;
    float t1 = 0.f, t2 = 0.f;
#pragma unroll 4
    for (int cc = gbase; cc < gbase + cpg; ++cc) { t1 += q1[cc]; t2 += q2[cc]; }
    m1[c] = t1 * inv_n;
    m2[c] = t2 * inv_n;
  }
  __syncthreads();
  const int slots = a.C >> 3;
  const int p0 = blockIdx.x * a.pix_per_block;
  const int p1 = min(a.HW, p0 + a.pix_per_block);
  const unsigned n_items = (unsigned)(p1 - p0) * (unsigned)slots;
  const int64_t pbase = (int64_t)b * a.HW + p0;
  const T* Y = reinterpret_cast<const T*>(a.Y);
  const T* dO = reinterpret_cast<const T*>(a.dO);
  T* dY = reinterpret_cast<T*>(a.out0);
  T* dR = reinterpret_cast<T*>(a.out1);
  for (unsigned it = threadIdx.x; it < n_items; it += blockDim.x) {
    const unsigned pq = it / (unsigned)slots;
    const int64_t pix = pbase + pq;
    const int c0 = (int)(it - pq * (unsigned)slots) << 3;
    f32x8 y = load8(Y + pix * a.y_ld + c0);
    f32x8 g = load8(dO + pix * a.do_ld + c0);
    f32x8 o, r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      const float nh = y.v[e] * rs[c] - mr[c];
      const float z = nh * ga[c] + be[c];
      const float da = g.v[e] * fac[c];
      const float dz = da * dsiluf_(z);
      r.v[e] = da;
      o.v[e] = rs[c] * (ga[c] * dz - m1[c] - nh * m2[c]);
    }
    store8(dY + pix * a.o0_ld + c0, o);
    if (MODE && dR) store8(dR + pix * a.o1_ld + c0, r);
  }
}

// dgamma[c] (+)= sum_b sums[b][c][0] ; dbeta[c] (+)= sum_b sums[b][c][1]   for every GroupNorm of a descriptor table
// block = 64 channels x 4 batch lanes (fixed-order LDS combine); one launch covers all layers of a backward half
__global__ __launch_bounds__(256) void gn_param_grad_kernel(const GnGradArgs) {
  gn_param_grad_block((GnGradArgsK)__builtin_amdgcn_kernarg_segment_ptr(), (int)blockIdx.x);
}

int gn_param_grad_multi(const GnGradDesc* descs, int n, int B, int accumulate, hipStream_t st) {
  for (int i0 = 0; i0 < n; i0 += GN_GRAD_MAX) {
    GnGradArgs A;
    A.n = std::min(GN_GRAD_MAX, n - i0);
    A.B = B; A.accumulate = accumulate;
    for (int i = 0; i < A.n; ++i) A.d[i] = descs[i0 + i];
    const int blk = gn_param_grad_layout(A);
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3(blk), dim3(256), 0, st, A);
    SX_LAUNCH_CHECK("gn_param_grad");
  }
  return SX_OK;
}

static int act_check(const ActArgs& a, const char* who) {
  if (a.C % 8 || a.C > MAXC || a.G <= 0 || a.C % a.G) {
    set_error("%s: unsupported channel count C=%d groups=%d (need C %% 8 == 0, C <= %d)", who, a.C, a.G, MAXC);
    return SX_ERR_UNSUPPORTED;
  }
  if (a.B <= 0 || a.HW <= 0) { set_error("%s: empty problem", who); return SX_ERR_INVALID; }
  return SX_OK;
}

static dim3 act_grid(ActArgs& a, int target = 2048) {
  // ~target blocks in total with at least 64 pixels per block; small problems (few hundred blocks at that size) go down
  // to 16 pixels per block so that the chip is covered: their time is a latency chain, not bandwidth
  const int big = std::min(cdiv(a.HW, 64), cdiv(target, a.B));
  const int small = std::min(cdiv(a.HW, 16), cdiv(512, a.B));
  int chunks = std::max(1, std::max(big, small));
  a.pix_per_block = cdiv(a.HW, chunks);
  chunks = cdiv(a.HW, a.pix_per_block);
  return dim3(chunks, a.B);
}

// slots per thread of act_bwd_group_kernel for this shape, 0 = not covered (the two-launch path runs)
static int act_group_slots(const ActArgs& a) {
  static const int on = env_int("SX_GN_GROUP_FUSED", 1);       // A/B knob
  if (!on) return 0;
  const int cpg = a.C / a.G;
  if (cpg % 8 || cpg > 64) return 0;
  const int ppp = cpg / 8;
  if (ppp & (ppp - 1)) return 0;
  const int64_t n_slots = (int64_t)a.HW * ppp;
  if (n_slots > 1024) return 0;
  return n_slots <= 256 ? 1 : (n_slots <= 512 ? 2 : 4);
}

template <typename T, int MODE>
static void act_group_launch(int nslot, dim3 grid, hipStream_t st, const ActArgs& a) {
  if (nslot == 1) hipLaunchKernelGGL((act_bwd_group_kernel<T, MODE, 1>), grid, dim3(256), 0, st, a);
  else if (nslot == 2) hipLaunchKernelGGL((act_bwd_group_kernel<T, MODE, 2>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((act_bwd_group_kernel<T, MODE, 4>), grid, dim3(256), 0, st, a);
}

bool gn_bwd_group_fused_ok(int B, int HW, int C, int groups) {
  ActArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.HW = HW; a.C = C; a.G = groups;
  return groups > 0 && C % groups == 0 && C % 8 == 0 && act_group_slots(a) > 0;
}

}  // namespace sx

using namespace sx;

#define SX_DISPATCH_T(dtype, KERN, MODE, grid, st, args)                                                   \
  do {                                                                                                     \
    const bool small_c = (args).C <= 256;                                                                  \
    if ((dtype) == SX_BF16) {                                                                              \
      if (small_c) hipLaunchKernelGGL((KERN<bf16, MODE, 256>), grid, dim3(256), 0, st, args);              \
      else hipLaunchKernelGGL((KERN<bf16, MODE, MAXC>), grid, dim3(256), 0, st, args);                     \
    } else {                                                                                               \
      if (small_c) hipLaunchKernelGGL((KERN<float, MODE, 256>), grid, dim3(256), 0, st, args);             \
      else hipLaunchKernelGGL((KERN<float, MODE, MAXC>), grid, dim3(256), 0, st, args);                    \
    }                                                                                                      \
  } while (0)

extern "C" int sx_gn_silu_forward(int dtype, int B, int HW, int C, int groups, float eps, const void* y, int y_ld,
                                  const float* stats, const float* gamma, const float* beta, void* out, int out_ld,
                                  void* stream) {
  SX_REQUIRE(y && stats && gamma && beta && out, "sx_gn_silu_forward: null pointer");
  ActArgs a;
  memset(&a, 0, sizeof(a));
  a.Y = y; a.y_ld = y_ld; a.stats = stats; a.gamma = gamma; a.beta = beta; a.out0 = out; a.o0_ld = out_ld;
  a.B = B; a.HW = HW; a.C = C; a.G = groups; a.eps = eps;
  SX_TRY(act_check(a, "sx_gn_silu_forward"));
  dim3 grid = act_grid(a);
  SX_DISPATCH_T(dtype, act_fwd_kernel, 0, grid, (hipStream_t)stream, a);
  SX_LAUNCH_CHECK("gn_silu_forward");
  return SX_OK;
}

extern "C" int sx_block_out_forward(int dtype, int B, int HW, int C, int groups, float eps, const void* y2, int y_ld,
                                    const float* stats, const float* gamma, const float* beta, const void* s, int s_ld,
                                    const float* film_g, const float* film_b, const float* drop, void* out, int out_ld,
                                    void* r_out, int r_ld, void* stream) {
  SX_REQUIRE(y2 && stats && gamma && beta && s && out, "sx_block_out_forward: null pointer");
  ActArgs a;
  memset(&a, 0, sizeof(a));
  a.Y = y2; a.y_ld = y_ld; a.S = s; a.s_ld = s_ld; a.stats = stats; a.gamma = gamma; a.beta = beta;
  a.gf = film_g; a.bf = film_b; a.drop = drop; a.out0 = out; a.o0_ld = out_ld; a.out1 = r_out; a.o1_ld = r_ld;
  a.B = B; a.HW = HW; a.C = C; a.G = groups; a.eps = eps;
  SX_TRY(act_check(a, "sx_block_out_forward"));
  dim3 grid = act_grid(a);
  SX_DISPATCH_T(dtype, act_fwd_kernel, 1, grid, (hipStream_t)stream, a);
  SX_LAUNCH_CHECK("block_out_forward");
  return SX_OK;
}

// sx_block_out_forward followed by nn.MaxPool2d(2) of `out`, in one pass (H, W even)
extern "C" int sx_block_out_pool_forward(int dtype, int B, int H, int W, int C, int groups, float eps, const void* y2, int y_ld,
                                         const float* stats, const float* gamma, const float* beta, const void* s, int s_ld,
                                         const float* film_g, const float* film_b, const float* drop, void* out, int out_ld,
                                         void* r_out, int r_ld, void* pool, int pool_ld, void* stream) {
  return sx_block_out_pool_idx_forward(dtype, B, H, W, C, groups, eps, y2, y_ld, stats, gamma, beta, s, s_ld, film_g, film_b, drop,
                                       out, out_ld, r_out, r_ld, pool, pool_ld, nullptr, 0, stream);
}

extern "C" int sx_block_out_pool_idx_forward(int dtype, int B, int H, int W, int C, int groups, float eps, const void* y2, int y_ld,
                                             const float* stats, const float* gamma, const float* beta, const void* s, int s_ld,
                                             const float* film_g, const float* film_b, const float* drop, void* out, int out_ld,
                                             void* r_out, int r_ld, void* pool, int pool_ld, void* pool_idx, int idx_ld,
                                             void* stream) {
  SX_REQUIRE(y2 && stats && gamma && beta && s && out && pool, "sx_block_out_pool_forward: null pointer");
  if (H < 2 || W < 2 || (H & 1) || (W & 1) || pool_ld % 8) {
    set_error("sx_block_out_pool_forward: needs even H, W and pool_ld %% 8 == 0 (H=%d W=%d)", H, W);
    return SX_ERR_UNSUPPORTED;
  }
  if (pool_idx && (idx_ld % 8 || idx_ld < C || ((uintptr_t)pool_idx % 8))) {
    set_error("sx_block_out_pool_idx_forward: pool_idx needs idx_ld %% 8 == 0, idx_ld >= C and an 8-byte aligned pointer");
    return SX_ERR_UNSUPPORTED;
  }
  ActArgs a;
  memset(&a, 0, sizeof(a));
  a.Y = y2; a.y_ld = y_ld; a.S = s; a.s_ld = s_ld; a.stats = stats; a.gamma = gamma; a.beta = beta;
  a.gf = film_g; a.bf = film_b; a.drop = drop; a.out0 = out; a.o0_ld = out_ld; a.out1 = r_out; a.o1_ld = r_ld;
  a.B = B; a.HW = H * W; a.C = C; a.G = groups; a.eps = eps; a.W = W; a.pool = pool; a.p_ld = pool_ld;
  a.pidx = (unsigned char*)pool_idx; a.pi_ld = idx_ld;
  SX_TRY(act_check(a, "sx_block_out_pool_forward"));
  // chunks of windows: the grid rule sees a quarter of the pixels, each item being four of them
  ActArgs g = a;
  g.HW = a.HW >> 2;
  dim3 grid = act_grid(g);
  a.pix_per_block = g.pix_per_block;
  hipStream_t st = (hipStream_t)stream;
  const bool small_c = C <= 256;
  if (dtype == SX_BF16) {
    if (small_c) hipLaunchKernelGGL((act_fwd_pool_kernel<bf16, 256>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((act_fwd_pool_kernel<bf16, MAXC>), grid, dim3(256), 0, st, a);
  } else {
    if (small_c) hipLaunchKernelGGL((act_fwd_pool_kernel<float, 256>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((act_fwd_pool_kernel<float, MAXC>), grid, dim3(256), 0, st, a);
  }
  SX_LAUNCH_CHECK("block_out_pool_forward");
  return SX_OK;
}

// phase 1 alone (the fallback of producers that could not fuse the sums into their epilogue); mode 0 or 2
namespace sx {
int gn_bwd_reduce_only(int dtype, int mode, int B, int HW, int C, int groups, float eps, const void* y, int y_ld, const float* stats,
                       const float* gamma, const float* beta, const void* d_out, int do_ld, const void* s, int s_ld,
                       const float* film_g, const float* drop, float* sums, hipStream_t st) {
  ActArgs a;
  memset(&a, 0, sizeof(a));
  a.Y = y; a.y_ld = y_ld; a.stats = stats; a.gamma = gamma; a.beta = beta; a.dO = d_out; a.do_ld = do_ld;
  a.R = s; a.r_ld = s_ld; a.gf = film_g; a.drop = drop; a.sums = sums;
  a.B = B; a.HW = HW; a.C = C; a.G = groups; a.eps = eps;
  SX_TRY(act_check(a, "gn_bwd_reduce"));
  dim3 grid = mode ? act_grid(a, 1024) : act_grid(a);
  if (mode == 2) { SX_DISPATCH_T(dtype, act_bwd_reduce_kernel, 2, grid, st, a); }
  else { SX_DISPATCH_T(dtype, act_bwd_reduce_kernel, 0, grid, st, a); }
  SX_LAUNCH_CHECK("gn_bwd_reduce");
  return SX_OK;
}
}  // namespace sx

// mode 0: gradient through A = silu(gn(Y));  mode 1: through the block output (FiLM, dropout, residual).
// sums must be zeroed by the caller ([B][C][4] fp32).
extern "C" int sx_gn_silu_backward(int dtype, int mode, int B, int HW, int C, int groups, float eps, const void* y,
                                   int y_ld, const float* stats, const float* gamma, const float* beta,
                                   const void* d_out, int do_ld, const void* r, int r_ld, const float* film_g,
                                   const float* drop, float* sums, void* dy, int dy_ld, void* d_r, int dr_ld,
                                   float* dgamma, float* dbeta, int accumulate, void* stream) {
  SX_REQUIRE(y && stats && gamma && beta && d_out && sums && dy, "sx_gn_silu_backward: null pointer");
  const bool sums_ready = (mode & SX_GN_BWD_SUMS_READY) != 0;      // the producer of d_out already accumulated `sums`
  mode &= ~SX_GN_BWD_SUMS_READY;
  SX_REQUIRE(mode == 0 || ((mode == 1 || mode == 2) && (r || sums_ready)), "sx_gn_silu_backward: mode 1 needs the saved block sum R, mode 2 the shortcut branch S");
  ActArgs a;
  memset(&a, 0, sizeof(a));
  a.Y = y; a.y_ld = y_ld; a.stats = stats; a.gamma = gamma; a.beta = beta; a.dO = d_out; a.do_ld = do_ld;
  a.R = r; a.r_ld = r_ld; a.gf = film_g; a.drop = drop; a.sums = sums; a.out0 = dy; a.o0_ld = dy_ld;
  a.out1 = d_r; a.o1_ld = dr_ld;
  a.B = B; a.HW = HW; a.C = C; a.G = groups; a.eps = eps;
  SX_TRY(act_check(a, "sx_gn_silu_backward"));
  dim3 grid = act_grid(a);
  hipStream_t st = (hipStream_t)stream;
  if (const int nslot = sums_ready ? 0 : act_group_slots(a)) {
    // small levels: both phases in one launch, one block per (sample, group)
    const dim3 gg(B, groups);
    if (dtype == SX_BF16) {
      if (mode == 2) act_group_launch<bf16, 2>(nslot, gg, st, a);
      else if (mode) act_group_launch<bf16, 1>(nslot, gg, st, a);
      else act_group_launch<bf16, 0>(nslot, gg, st, a);
    } else {
      if (mode == 2) act_group_launch<float, 2>(nslot, gg, st, a);
      else if (mode) act_group_launch<float, 1>(nslot, gg, st, a);
      else act_group_launch<float, 0>(nslot, gg, st, a);
    }
    SX_LAUNCH_CHECK("gn_bwd_group");
    if (dgamma && dbeta) {
      GnGradDesc d;
      d.sums = sums; d.dgamma = dgamma; d.dbeta = dbeta; d.C = C; d.blk0 = 0; d.nb = 0; d.stride = 4;
      SX_TRY(gn_param_grad_multi(&d, 1, B, accumulate, st));
    }
    return SX_OK;
  }
  // the block-output pass (four sums, three tensors) pays more per block for its combine: half as many blocks
  dim3 grid_r = mode ? act_grid(a, 1024) : grid;
  if (!sums_ready) {
    if (mode == 2) { SX_DISPATCH_T(dtype, act_bwd_reduce_kernel, 2, grid_r, st, a); }
    else if (mode) { SX_DISPATCH_T(dtype, act_bwd_reduce_kernel, 1, grid_r, st, a); }
    else { SX_DISPATCH_T(dtype, act_bwd_reduce_kernel, 0, grid_r, st, a); }
    SX_LAUNCH_CHECK("gn_bwd_reduce");
  }
  grid = act_grid(a);
  if (mode) { SX_DISPATCH_T(dtype, act_bwd_apply_kernel, 1, grid, st, a); }
  else { SX_DISPATCH_T(dtype, act_bwd_apply_kernel, 0, grid, st, a); }
  SX_LAUNCH_CHECK("gn_bwd_apply");
  if (dgamma && dbeta) {
    GnGradDesc d;
    d.sums = sums; d.dgamma = dgamma; d.dbeta = dbeta; d.C = C; d.blk0 = 0; d.nb = 0; d.stride = 4;
    SX_TRY(gn_param_grad_multi(&d, 1, B, accumulate, st));
  }
  return SX_OK;
}
