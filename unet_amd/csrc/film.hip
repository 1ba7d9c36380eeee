// FiLM generators (models.py:38-58): [gamma|beta] = W2 * SiLU(W1 p + b1) + b2, returned as
// (1 + gamma, beta) per (sample, channel).  All generators of the network run in ONE launch
// (grid = samples x generators).  Tiny GEMVs: <0.01 % of the FLOPs, so plain FMA code.
#include "sx_common.h"
#include "film.h"
#include "wgrad_common.h"
#include <algorithm>
#include <cstddef>

namespace sx {

// The launch tables are read straight from the kernel-argument segment (scalar loads at a uniform dynamic offset) into
// registers: indexing the by-value parameter with blockIdx, or passing it on by reference, can make the compiler copy the
// whole 2-3 KB struct to scratch, per lane (first seen in wgrad_ws_multi_kernel).
struct FilmCtx { int B, P, Hf, accumulate; const float* params; float* dparams; float* zero; int zero_n; };
typedef const FilmArgs __attribute__((address_space(4))) * FilmArgsK;
__device__ __forceinline__ void film_desc_load(FilmGenDesc& d, FilmArgsK K, int gen) {
  d.w1 = K->g[gen].w1; d.b1 = K->g[gen].b1; d.w2 = K->g[gen].w2; d.b2 = K->g[gen].b2;
  d.dw1 = K->g[gen].dw1; d.db1 = K->g[gen].db1; d.dw2 = K->g[gen].dw2; d.db2 = K->g[gen].db2;
  d.gf = K->g[gen].gf; d.bf = K->g[gen].bf; d.pre = K->g[gen].pre;
  d.dgf = K->g[gen].dgf; d.dbf = K->g[gen].dbf; d.dg_stride = K->g[gen].dg_stride;
  d.dpre = K->g[gen].dpre; d.C = K->g[gen].C;
}
#define SX_FILM_FETCH(a, d, gen)                                              \
  FilmArgsK K_ = (FilmArgsK)__builtin_amdgcn_kernarg_segment_ptr();           \
  FilmGenDesc d;                                                              \
  film_desc_load(d, K_, (int)(gen));                                          \
  FilmCtx a;                                                                  \
  a.B = K_->B; a.P = K_->P; a.Hf = K_->Hf; a.accumulate = K_->accumulate;     \
  a.params = K_->params; a.dparams = K_->dparams; a.zero = K_->zero; a.zero_n = K_->zero_n

__global__ __launch_bounds__(256) void film_fwd_kernel(const FilmArgs) {
  __shared__ float hid[FILM_MAX_HID];
  __shared__ float pv[64];
  const int b = blockIdx.x;
  SX_FILM_FETCH(a, d, blockIdx.y);
  if (a.zero) {
    const int nthr = gridDim.x * gridDim.y * blockDim.x;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < a.zero_n; i += nthr) a.zero[i] = 0.f;
  }
  film_fwd_block(d.w1, d.b1, d.w2, d.b2, d.gf, d.bf, d.pre, d.C, b, a.P, a.Hf, a.params, hid, pv);
}

// phase A (grid = samples/8 x ngen): dpre[b][j] = (sum_k dout[b][k] W2[k][j]) * silu'(pre[b][j]);
// dparams[b][p] += sum_j dpre[b][j] W1[j][p].  A block handles 4 samples so every W2 element it loads serves 4 FMAs;
// threads = (hidden unit j) x (k mod FKP), combined through LDS.
static constexpr int FB = 4;
static constexpr int FKP = 8;        // threads that share one hidden unit (k is split FKP ways): 128 x FKP threads
__global__ __launch_bounds__(1024) void film_bwd_a_kernel(const FilmArgs) {
  __shared__ float dout[FB][2 * 1024];
  __shared__ float part[FKP][FB][128];
  __shared__ float dp[FB][128];
  SX_FILM_FETCH(a, d, blockIdx.y);
  const int b0 = blockIdx.x * FB, nb = min(FB, a.B - b0);
  const int K = 2 * d.C;
  for (int i = threadIdx.x; i < FB * K; i += blockDim.x) {
    const int bb = i / K, k = i - bb * K;
    const int c = k < d.C ? k : k - d.C;
    const float* src = k < d.C ? d.dgf : d.dbf;
    dout[bb][k] = bb < nb ? src[((int64_t)(b0 + bb) * d.C + c) * d.dg_stride] : 0.f;
  }
  __syncthreads();
  const int jl = threadIdx.x & 127, kp = threadIdx.x >> 7;
  for (int j0 = 0; j0 < a.Hf; j0 += 128) {
    const int j = j0 + jl;
    float s[FB];
#pragma unroll
    for (int bb = 0; bb < FB; ++bb) s[bb] = 0.f;
    if (j < a.Hf)
      for (int k = kp; k < K; k += 16 * FKP) {    // 16 independent loads in flight per thread before the first FMA
        float w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = (k + FKP * u < K) ? d.w2[(int64_t)(k + FKP * u) * a.Hf + j] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int bb = 0; bb < FB; ++bb) s[bb] += dout[bb][min(k + FKP * u, K - 1)] * w[u];
      }
#pragma unroll
    for (int bb = 0; bb < FB; ++bb) part[kp][bb][jl] = s[bb];
    __syncthreads();
    for (int i = threadIdx.x; i < FB * 128; i += blockDim.x) {
      const int bb = i >> 7, jj = i & 127;
      float v = 0.f;
      if (bb < nb && j0 + jj < a.Hf) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < FKP; ++q) t += part[q][bb][jj];
        v = t * dsiluf_(d.pre[(int64_t)(b0 + bb) * a.Hf + j0 + jj]);
        d.dpre[(int64_t)(b0 + bb) * a.Hf + j0 + jj] = v;
      }
      dp[bb][jj] = v;
    }
    __syncthreads();
    if (a.dparams)
      for (int i = threadIdx.x; i < nb * a.P; i += blockDim.x) {
        const int bb = i / a.P, p = i - bb * a.P;
        float v = 0.f;
        for (int jj = 0; jj < min(128, a.Hf - j0); ++jj) v += dp[bb][jj] * d.w1[(j0 + jj) * a.P + p];
        atomicAdd(a.dparams + (int64_t)(b0 + bb) * a.P + p, v);
      }
    __syncthreads();
  }
}

// phase B (grid = row chunks x ngen): weight / bias gradients, summed over the batch in a fixed order.
// Rows 0 .. 2C-1 are rows of dW2 (+ db2), rows 2C .. 2C+Hf-1 rows of dW1 (+ db1).  A block takes FR rows and
// stages the batch-side operand once in LDS: hid[b][j] = silu(pre[b][j]) for the dW2 rows, params[b][p] for dW1.
static constexpr int FR = 16, FBT = 32;      // rows per block, batch tile

// dW1 / db1 rows j0 .. j0 + nrows - 1 of one generator with dpre rebuilt in place (256 threads):
//   dpre[b][j] = silu'(pre[b][j]) * sum_c (dgf[b][c] W2[c][j] + dbf[b][c] W2[C + c][j])
// Per 64-sample tile and 64-channel chunk the block first stages the (dgf, dbf) pairs (one 16-byte load per (sample, channel)
// when they are the last two of a four-float record, as in the network plan) and the 2 x 64 W2 rows of its 16 hidden units in
// LDS -- every load of a chunk issued before the first LDS store: one memory latency per chunk.  Walking the channels with
// the loads inside the loop cost ~900 cycles per channel (115 us for the launch), eight channels per trip 45 us, a
// load-then-store staging loop 46 us, this form 33 us; holding the next chunk in registers under the multiply was slower
// again (42 us).  Then thread (sample, hidden-unit quad) accumulates from LDS, and thread (row, column mod 16) sums
// dpre[b][row] * params[b][p] over the tile in a fixed order.
static constexpr int FWC = 64;       // channels per chunk
struct FilmW1Smem {
  f32x2 dout[64][FWC + 1];           // [sample][channel] (dgf, dbf)
  f32x4 w2[2][FWC][4];               // [gamma / beta half][channel][hidden-unit quad]
  float dpre[64][FR + 1];
  float par[64][64 + 1];
};
__device__ __forceinline__ void film_w1_fused(const FilmCtx& a, const FilmGenDesc& d, int j0, int nrows, FilmW1Smem& S) {
  const int tb = threadIdx.x >> 2, q = threadIdx.x & 3;
  const int rr = threadIdx.x & 15, pc = threadIdx.x >> 4;        // phase 2: row, column (+ 16 i)
  float accw[4] = {0.f, 0.f, 0.f, 0.f}, accb = 0.f;
  const bool rec4 = d.dg_stride == 4 && d.dbf == d.dgf + 1 && (((uintptr_t)(d.dgf - 2)) % 16) == 0;
  const bool vec = (a.Hf % 4 == 0) && j0 + FR <= a.Hf && (((uintptr_t)d.w2 % 16) == 0);
  for (int bt = 0; bt < a.B; bt += 64) {
    const int nb = min(64, a.B - bt);
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < d.C; c0 += FWC) {
      const int nc = min(FWC, d.C - c0);
      __syncthreads();                     // the previous chunk / tile is done with the images
      constexpr int ND = 64 * FWC / 256, NW = 2 * FWC * 4 / 256;
      f32x4 rd[ND], rw[NW];
      f32x2 rs[ND];
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        const int i = threadIdx.x + u * 256, b2 = i / FWC, cc = i - b2 * FWC;
        rd[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        rs[u] = f32x2{0.f, 0.f};
        if (b2 < nb && cc < nc) {
          const int64_t e = ((int64_t)(bt + b2) * d.C + c0 + cc) * d.dg_stride;
          if (rec4) rd[u] = *reinterpret_cast<const f32x4*>(d.dgf - 2 + e);
          else rs[u] = f32x2{d.dgf[e], d.dbf[e]};
        }
      }
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const int i = threadIdx.x + u * 256, h = i / (FWC * 4), cc = (i >> 2) % FWC, qq = i & 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (cc < nc) {
          const float* w = d.w2 + (int64_t)(h * d.C + c0 + cc) * a.Hf + j0 + 4 * qq;
          if (vec) v = *reinterpret_cast<const f32x4*>(w);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = j0 + 4 * qq + e < a.Hf ? w[e] : 0.f;
          }
        }
        rw[u] = v;
      }
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        const int i = threadIdx.x + u * 256, b2 = i / FWC, cc = i - b2 * FWC;
        S.dout[b2][cc] = rec4 ? f32x2{rd[u][2], rd[u][3]} : rs[u];
      }
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const int i = threadIdx.x + u * 256;
        S.w2[i / (FWC * 4)][(i >> 2) % FWC][i & 3] = rw[u];
      }
      __syncthreads();
#pragma unroll 8
      for (int cc = 0; cc < FWC; ++cc) {
        const f32x2 gb = S.dout[tb][cc];
        t += gb[0] * S.w2[0][cc][q] + gb[1] * S.w2[1][cc][q];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * q + e;
      float v = 0.f;
      if (tb < nb && r < nrows) v = t[e] * dsiluf_(d.pre[(int64_t)(bt + tb) * a.Hf + j0 + r]);
      S.dpre[tb][r] = v;            // (not read by the chunk loop: no barrier needed before this store)
    }
    for (int i = threadIdx.x; i < 64 * a.P; i += blockDim.x) {
      const int b2 = i / a.P, p = i - b2 * a.P;
      S.par[b2][p] = b2 < nb ? a.params[(int64_t)(bt + b2) * a.P + p] : 0.f;
    }
    __syncthreads();
    for (int b2 = 0; b2 < nb; ++b2) {
      const float l = S.dpre[b2][rr];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (pc + 16 * i < a.P) accw[i] += l * S.par[b2][pc + 16 * i];
      accb += l;
    }
    // (the next tile's first barrier comes before anything rewrites dpre / par)
  }
  if (rr < nrows) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = pc + 16 * i;
      if (p < a.P) {
        float* dst = d.dw1 + (int64_t)(j0 + rr) * a.P + p;
        *dst = a.accumulate ? *dst + accw[i] : accw[i];
      }
    }
    if (pc == 0) {
      float* dst = d.db1 + j0 + rr;
      *dst = a.accumulate ? *dst + accb : accb;
    }
  }
}

// FUSED: the dW1 rows rebuild dpre for their own 16 hidden units (film_w1_fused) instead of reading phase A's output: a
// training step needs no phase A at all then.
struct FilmBSmem { float hid[FBT][128 + 1]; float lhs[FBT][FR]; };
template <bool FUSED>
__device__ __forceinline__ void film_bwd_b_body() {
  // the roles of a block exclude each other: one buffer
  __shared__ __align__(16) char smem_raw[FUSED ? (sizeof(FilmW1Smem) > sizeof(FilmBSmem) ? sizeof(FilmW1Smem) : sizeof(FilmBSmem)) : sizeof(FilmBSmem)];
  FilmBSmem& SB = *reinterpret_cast<FilmBSmem*>(smem_raw);
  float (&hid)[FBT][128 + 1] = SB.hid;
  float (&lhs)[FBT][FR] = SB.lhs;
  SX_FILM_FETCH(a, d, blockIdx.y);
  const int K2 = 2 * d.C;
  const int r0 = blockIdx.x * FR;
  if (r0 >= K2 + a.Hf) return;
  const bool w2rows = r0 < K2;                 // K2 and Hf chunks never share a block (K2 % FR == 0)
  if constexpr (FUSED) {
    if (!w2rows) { film_w1_fused(a, d, r0 - K2, min(FR, a.Hf - (r0 - K2)), *reinterpret_cast<FilmW1Smem*>(smem_raw)); return; }
  }
  const int ncols = w2rows ? a.Hf : a.P;       // columns of the gradient rows this block owns
  const int nrows = min(FR, (w2rows ? K2 : K2 + a.Hf) - r0);
  for (int c0 = 0; c0 < ncols; c0 += 128) {
    const int cw = min(128, ncols - c0);
    // thread -> (column, row group): 128 columns x 2 row halves
    const int cl = threadIdx.x & 127, rh = threadIdx.x >> 7;
    float acc[FR / 2], bias[FR / 2];
#pragma unroll
    for (int i = 0; i < FR / 2; ++i) { acc[i] = 0.f; bias[i] = 0.f; }
    for (int bt = 0; bt < a.B; bt += FBT) {
      const int nb = min(FBT, a.B - bt);
      __syncthreads();
      for (int i = threadIdx.x; i < FBT * 128; i += blockDim.x) {
        const int bb = i >> 7, cc = i & 127;
        float v = 0.f;
        if (bb < nb && cc < cw)
          v = w2rows ? siluf_(d.pre[(int64_t)(bt + bb) * a.Hf + c0 + cc]) : a.params[(int64_t)(bt + bb) * a.P + c0 + cc];
        hid[bb][cc] = v;
      }
      for (int i = threadIdx.x; i < FBT * FR; i += blockDim.x) {
        const int bb = i / FR, rr = i - bb * FR;
        float v = 0.f;
        if (bb < nb && rr < nrows) {
          const int row = r0 + rr;
          if (w2rows) {
            const int c = row < d.C ? row : row - d.C;
            v = (row < d.C ? d.dgf : d.dbf)[((int64_t)(bt + bb) * d.C + c) * d.dg_stride];
          } else {
            v = d.dpre[(int64_t)(bt + bb) * a.Hf + (row - K2)];
          }
        }
        lhs[bb][rr] = v;
      }
      __syncthreads();
      for (int bb = 0; bb < nb; ++bb) {
        const float h = hid[bb][cl];
#pragma unroll
        for (int i = 0; i < FR / 2; ++i) {
          const float l = lhs[bb][rh * (FR / 2) + i];
          acc[i] += l * h;
          bias[i] += l;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < FR / 2; ++i) {
      const int rr = rh * (FR / 2) + i;
      if (rr >= nrows) continue;
      const int row = r0 + rr;
      if (cl < cw) {
        float* dst = w2rows ? d.dw2 + (int64_t)row * a.Hf + c0 + cl : d.dw1 + (int64_t)(row - K2) * a.P + c0 + cl;
        *dst = a.accumulate ? *dst + acc[i] : acc[i];
      }
      if (c0 == 0 && cl == 0) {
        float* dst = w2rows ? d.db2 + row : d.db1 + (row - K2);
        *dst = a.accumulate ? *dst + bias[i] : bias[i];
      }
    }
  }
}

__global__ __launch_bounds__(256) void film_bwd_b_kernel(const FilmArgs) { film_bwd_b_body<false>(); }

// Training tail: rows y < ngen are the generators (dW2 rows and fused dW1 rows), the rows behind them run the GroupNorm affine /
// bias gradient reductions of the same backward half.
// ... and behind those, the slab reduction of the same half's weight gradients (round 3): it is independent of the generators,
// and as a launch of its own (16 us) it ran while this one's 40 latency-bound blocks (33 us) had the chip to themselves.
// Table sizes are what fits the 4 KB kernel-argument segment next to FilmArgs; larger halves take the separate launches.
static constexpr int TAIL_GN = 20, TAIL_RED = 15;
struct TailGn { GnGradDesc d[TAIL_GN]; int n, B, accumulate; };
struct FilmTailArgs { FilmArgs f; TailGn gn; WgReduceDesc rd[TAIL_RED]; int gn_blocks, n_rd, rd_blocks; };
static_assert(offsetof(FilmTailArgs, f) == 0, "the FiLM table leads the kernel arguments");
static_assert(sizeof(FilmTailArgs) <= 4096, "kernel-argument segment");
__global__ __launch_bounds__(256) void film_bwd_train_kernel(const FilmTailArgs) {
  typedef const FilmTailArgs __attribute__((address_space(4))) * TailK;      // FilmArgs sits at offset 0: the body finds it there
  TailK A = (TailK)__builtin_amdgcn_kernarg_segment_ptr();
  const int ngen = A->f.ngen;
  if ((int)blockIdx.y >= ngen) {
    int block = ((int)blockIdx.y - ngen) * (int)gridDim.x + (int)blockIdx.x;
    if (block < A->gn_blocks) { gn_param_grad_block(&A->gn, block); return; }
    block -= A->gn_blocks;
    if (block < A->rd_blocks) {
      int di = 0;
      const int n = A->n_rd;
      while (di + 1 < n && block >= A->rd[di + 1].blk0) ++di;
      WgReduceDesc D;
      D.slab = A->rd[di].slab; D.out = A->rd[di].out; D.nsplit = A->rd[di].nsplit; D.ntaps = A->rd[di].ntaps;
      D.Ca = A->rd[di].Ca; D.Cb = A->rd[di].Cb; D.CaPad = A->rd[di].CaPad; D.CbPad = A->rd[di].CbPad;
      D.accumulate = A->rd[di].accumulate; D.tiled = A->rd[di].tiled; D.a16_lo = A->rd[di].a16_lo; D.a16_n = A->rd[di].a16_n;
      D.tap_lo = A->rd[di].tap_lo; D.tap_n = A->rd[di].tap_n; D.blk0 = A->rd[di].blk0; D.nblk = A->rd[di].nblk;
      wgrad_reduce_block(D, block - D.blk0);
    }
    return;
  }
  film_bwd_b_body<true>();
}

int film_forward(const FilmArgs& a, hipStream_t st) {
  if (a.ngen <= 0) return SX_OK;
  if (a.ngen > FILM_MAX_GEN || a.Hf > FILM_MAX_HID || a.P > 64) {
    set_error("FiLM: unsupported sizes (generators=%d hidden=%d P=%d)", a.ngen, a.Hf, a.P);
    return SX_ERR_UNSUPPORTED;
  }
  for (int i = 0; i < a.ngen; ++i)
    if (a.g[i].C > 1024) { set_error("FiLM: C=%d > 1024", a.g[i].C); return SX_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL(film_fwd_kernel, dim3(a.B, a.ngen), dim3(256), 0, st, a);
  SX_LAUNCH_CHECK("film_fwd");
  return SX_OK;
}

int film_backward(const FilmArgs& a, bool weight_grads, hipStream_t st, const GnGradDesc* gn, int n_gn, int* gn_done,
                  const WgReduceDesc* rd, int n_rd, int* rd_done) {
  if (gn_done) *gn_done = 0;
  if (rd_done) *rd_done = 0;
  if (a.ngen <= 0) return SX_OK;
  static const int no_tail = env_int("SX_FILM_NO_TAIL", 0);        // A/B knob: the three-launch form
  static const int tail_reduce = env_int("SX_TAIL_REDUCE", 1);     // A/B knob: slab reduction inside the tail launch
  if (weight_grads && !a.dparams && n_gn <= TAIL_GN && !no_tail) {
    FilmTailArgs T;
    T.f = a;
    memset(&T.gn, 0, sizeof(T.gn));
    T.gn.n = n_gn; T.gn.B = a.B; T.gn.accumulate = a.accumulate;
    for (int i = 0; i < n_gn; ++i) T.gn.d[i] = gn[i];
    T.gn_blocks = n_gn > 0 ? gn_param_grad_layout(T.gn) : 0;
    T.n_rd = 0; T.rd_blocks = 0;
    if (rd && rd_done && n_rd > 0 && n_rd <= TAIL_RED && tail_reduce) {
      int blk = 0;
      for (int i = 0; i < n_rd; ++i) {
        T.rd[i] = rd[i];
        T.rd[i].blk0 = blk;
        T.rd[i].nblk = wgrad_reduce_blocks(T.rd[i]);
        blk += T.rd[i].nblk;
      }
      T.n_rd = n_rd; T.rd_blocks = blk;
      *rd_done = 1;
    }
    int cmax = 0;
    for (int i = 0; i < a.ngen; ++i) cmax = std::max(cmax, a.g[i].C);
    const int gx = cdiv(2 * cmax + a.Hf, FR);
    hipLaunchKernelGGL(film_bwd_train_kernel, dim3(gx, a.ngen + cdiv(T.gn_blocks + T.rd_blocks, gx)), dim3(256), 0, st, T);
    SX_LAUNCH_CHECK("film_bwd_train");
    if (gn_done) *gn_done = 1;
    return SX_OK;
  }
  hipLaunchKernelGGL(film_bwd_a_kernel, dim3(cdiv(a.B, FB), a.ngen), dim3(128 * FKP), 0, st, a);
  SX_LAUNCH_CHECK("film_bwd_a");
  if (weight_grads) {
    int cmax = 0;
    for (int i = 0; i < a.ngen; ++i) cmax = std::max(cmax, a.g[i].C);
    // 2C is a multiple of FR (C % 8 == 0), so a block never mixes dW2 rows with dW1 rows
    hipLaunchKernelGGL(film_bwd_b_kernel, dim3(cdiv(2 * cmax + a.Hf, FR), a.ngen), dim3(256), 0, st, a);
    SX_LAUNCH_CHECK("film_bwd_b");
  }
  return SX_OK;
}

}  // namespace sx
