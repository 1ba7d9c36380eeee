// Small bandwidth-bound helpers around the convolutions: input packing (NCHW fp32 -> NHWC T),
// MaxPool2d(2) forward (models.py:106,177,181) and its backward fused with the skip-connection
// gradient add, per-channel sums (bias gradients of ConvTranspose2d / the 1x1 head,
// models.py:127,132), global average pool (models.py:10-12) and its broadcast inverse (:32-35).
#include "sx_common.h"

namespace sx {

// x: (B, Cin, H, W) fp32 with arbitrary element strides -> out NHWC T with ld channels (zero padded)
template <typename T>
__global__ void pack_input_kernel(const float* __restrict__ x, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                                  T* __restrict__ out, int B, int C, int H, int W, int ld) {
  // one thread per (pixel, 8-channel group): a 16-byte (bf16) store instead of eight 2-byte ones; ld % 8 == 0
  const int groups = ld >> 3;
  const int64_t total = (int64_t)B * H * W * groups;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % groups) << 3;
    int64_t p = i / groups;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int b = (int)(p / H);
    const float* src = x + b * sb + h * sh + w * sw;
    f32x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v.v[e] = c0 + e < C ? src[(c0 + e) * sc] : 0.f;
    store8(out + (i << 3), v);
  }
}

// NHWC T (ld) -> (B, C, H, W) fp32 contiguous
template <typename T>
__global__ void unpack_output_kernel(const T* __restrict__ in, int ld, float* __restrict__ out, int B, int C, int H, int W) {
  const int64_t total = (int64_t)B * C * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H); p /= H;
    const int c = (int)(p % C);
    const int b = (int)(p / C);
    out[i] = to_f32(in[(((int64_t)b * H + h) * W + w) * ld + c]);
  }
}

// (slot, x, y, b) of a flat work item; 32-bit divisions when the item count allows (a 64-bit division is ~100 VALU
// instructions, and these kernels run one item per thread: the index math sits in front of every load)
__device__ __forceinline__ void split_item(int64_t i, bool small, int slots, int Wn, int Hn, int& c0, int& ix, int& iy, int& b) {
  if (small) {
    const unsigned u = (unsigned)i, q = u / (unsigned)slots, r = q / (unsigned)Wn, bb = r / (unsigned)Hn;
    c0 = (int)(u - q * (unsigned)slots) << 3;
    ix = (int)(q - r * (unsigned)Wn);
    iy = (int)(r - bb * (unsigned)Hn);
    b = (int)bb;
  } else {
    c0 = (int)(i % slots) << 3;
    int64_t p = i / slots;
    ix = (int)(p % Wn); p /= Wn;
    iy = (int)(p % Hn);
    b = (int)(p / Hn);
  }
}

template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld, int B, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, slots = C >> 3;
  const int64_t total = (int64_t)B * Ho * Wo * slots;
  const bool small = total < ((int64_t)1 << 32);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c0, ox, oy, b;
    split_item(i, small, slots, Wo, Ho, c0, ox, oy, b);
    const T* src = x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * x_ld + c0;
    f32x8 v00 = load8(src), v01 = load8(src + x_ld), v10 = load8(src + (int64_t)W * x_ld), v11 = load8(src + (int64_t)(W + 1) * x_ld);
    f32x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = fmaxf(fmaxf(v00.v[e], v01.v[e]), fmaxf(v10.v[e], v11.v[e]));
    store8(y + (((int64_t)b * Ho + oy) * Wo + ox) * y_ld + c0, o);
  }
}

// dX[b,y,x,c] = (dskip ? dskip[b,y,x,c] : 0) + (x is the first maximum of its 2x2 window ? dP[b,y/2,x/2,c] : 0)
// One thread handles a whole 2x2 window (8 channels); odd trailing rows/cols get only the skip term.
template <typename T>
__global__ void maxpool_bwd_add_kernel(const T* __restrict__ x, int x_ld, const T* __restrict__ dp, int dp_ld,
                                       const T* __restrict__ dskip, int ds_ld, T* __restrict__ dx, int dx_ld,
                                       int B, int H, int W, int C) {
  const int Hw = (H + 1) >> 1, Ww = (W + 1) >> 1, Ho = H >> 1, Wo = W >> 1, slots = C >> 3;
  const int64_t total = (int64_t)B * Hw * Ww * slots;
  const bool small = total < ((int64_t)1 << 32);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c0, wx, wy, b;
    split_item(i, small, slots, Ww, Hw, c0, wx, wy, b);
    const bool pooled = wy < Ho && wx < Wo;
    f32x8 v[4], g;
    int best[8];
    if (pooled) {
      g = load8(dp + (((int64_t)b * Ho + wy) * Wo + wx) * dp_ld + c0);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        v[k] = load8(x + (((int64_t)b * H + 2 * wy + (k >> 1)) * W + 2 * wx + (k & 1)) * x_ld + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int bi = 0; float bv = v[0].v[e];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (v[k].v[e] > bv || (v[k].v[e] != v[k].v[e] && bv == bv)) { bv = v[k].v[e]; bi = k; }
        best[e] = bi;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = 2 * wy + (k >> 1), xx = 2 * wx + (k & 1);
      if (yy >= H || xx >= W) continue;
      const int64_t pix = ((int64_t)b * H + yy) * W + xx;
      f32x8 o;
      if (dskip) o = load8(dskip + pix * ds_ld + c0);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = 0.f;
      }
      if (pooled) {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (best[e] == k) o.v[e] += g.v[e];
      }
      store8(dx + pix * dx_ld + c0, o);
    }
  }
}

// The same with the window positions recorded by the forward pass (sx_block_out_pool_idx_forward: one byte per pooled
// element): the block output does not have to be read again to find them.  H and W even.
template <typename T>
__global__ void maxpool_bwd_add_idx_kernel(const unsigned char* __restrict__ idx, int idx_ld, const T* __restrict__ dp, int dp_ld,
                                           const T* __restrict__ dskip, int ds_ld, T* __restrict__ dx, int dx_ld,
                                           int B, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, slots = C >> 3;
  const int64_t total = (int64_t)B * Ho * Wo * slots;
  const bool small = total < ((int64_t)1 << 32);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c0, wx, wy, b;
    split_item(i, small, slots, Wo, Ho, c0, wx, wy, b);
    const int64_t win = ((int64_t)b * Ho + wy) * Wo + wx;
    const f32x8 g = load8(dp + win * dp_ld + c0);
    const unsigned long long ks = *reinterpret_cast<const unsigned long long*>(idx + win * idx_ld + c0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t pix = ((int64_t)b * H + 2 * wy + (k >> 1)) * W + 2 * wx + (k & 1);
      f32x8 o;
      if (dskip) o = load8(dskip + pix * ds_ld + c0);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) if ((int)((ks >> (8 * e)) & 3) == k) o.v[e] += g.v[e];
      store8(dx + pix * dx_ld + c0, o);
    }
  }
}

// out[c] (+)= sum over b, y < Hs, x < Ws of t[b,y,x,c]   (t is B x H x W, ld).  C % 8 == 0, C <= 2048:
// each thread owns an 8-channel slot and walks pixels (16-byte loads), lanes combine through LDS.
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ t, int ld, int B, int H, int W, int Hs, int Ws,
                                                          int C, int C_out, float* __restrict__ out, int pix_per_block, int nslots) {
  __shared__ float red[256 * 8];
  const int64_t npix = (int64_t)B * Hs * Ws;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
  const int64_t p1 = min(npix, p0 + pix_per_block);
  const int slots = C >> 3, lanes = 256 / slots;
  const int slot = threadIdx.x % slots, pl = threadIdx.x / slots;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (pl < lanes)
    for (int64_t p = p0 + pl; p < p1; p += lanes) {
      int64_t q = p;
      const int xx = (int)(q % Ws); q /= Ws;
      const int yy = (int)(q % Hs);
      const int b = (int)(q / Hs);
      const f32x8 v = load8(t + (((int64_t)b * H + yy) * W + xx) * ld + (slot << 3));
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v.v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < C_out; c += blockDim.x) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[(l * slots + (c >> 3)) * 8 + (c & 7)];
    atomicAdd(out + (blockIdx.x % nslots) * C_out + c, s);      // out: [nslots][C_out], block b adds into row b % nslots
  }
}

// z[b][c] = mean over pixels (models.py:10-12)
template <typename T>
__global__ void global_avg_pool_kernel(const T* __restrict__ x, int ld, int HW, int C, float* __restrict__ z) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += to_f32(x[((int64_t)b * HW + p) * ld + c]);
    z[(int64_t)b * C + c] = s / (float)HW;
  }
}

__global__ void fill_zero_kernel(float* p, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

static inline int grid_for(int64_t n, int per_block = 256, int cap = 4096) {
  return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(n, per_block), cap));
}

}  // namespace sx

using namespace sx;

extern "C" int sx_pack_input(int dtype, const float* x, int64_t sb, int64_t sc, int64_t sh, int64_t sw, int B, int C,
                             int H, int W, void* out, int out_ld, void* stream) {
  SX_REQUIRE(x && out && out_ld >= C, "sx_pack_input: bad arguments");
  if (out_ld % 8) { set_error("sx_pack_input: out_ld must be a multiple of 8 (got %d)", out_ld); return SX_ERR_UNSUPPORTED; }
  const int64_t total = (int64_t)B * H * W * (out_ld / 8);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16) hipLaunchKernelGGL(pack_input_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, st, x, sb, sc, sh, sw, (bf16*)out, B, C, H, W, out_ld);
  else hipLaunchKernelGGL(pack_input_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, x, sb, sc, sh, sw, (float*)out, B, C, H, W, out_ld);
  SX_LAUNCH_CHECK("pack_input");
  return SX_OK;
}

extern "C" int sx_unpack_output(int dtype, const void* in, int in_ld, int B, int C, int H, int W, float* out, void* stream) {
  SX_REQUIRE(in && out && in_ld >= C, "sx_unpack_output: bad arguments");
  const int64_t total = (int64_t)B * C * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16) hipLaunchKernelGGL(unpack_output_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16*)in, in_ld, out, B, C, H, W);
  else hipLaunchKernelGGL(unpack_output_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)in, in_ld, out, B, C, H, W);
  SX_LAUNCH_CHECK("unpack_output");
  return SX_OK;
}

extern "C" int sx_maxpool2_forward(int dtype, int B, int H, int W, int C, const void* x, int x_ld, void* y, int y_ld, void* stream) {
  SX_REQUIRE(x && y, "sx_maxpool2_forward: null pointer");
  if (C % 8 || x_ld % 8 || y_ld % 8 || H < 2 || W < 2) { set_error("sx_maxpool2_forward: need C %% 8 == 0 and H, W >= 2 (C=%d H=%d W=%d)", C, H, W); return SX_ERR_UNSUPPORTED; }
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16*)x, x_ld, (bf16*)y, y_ld, B, H, W, C);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, x_ld, (float*)y, y_ld, B, H, W, C);
  SX_LAUNCH_CHECK("maxpool_fwd");
  return SX_OK;
}

extern "C" int sx_maxpool2_backward_add(int dtype, int B, int H, int W, int C, const void* x, int x_ld, const void* dp,
                                        int dp_ld, const void* dskip, int ds_ld, void* dx, int dx_ld, void* stream) {
  SX_REQUIRE(x && dp && dx, "sx_maxpool2_backward_add: null pointer");
  if (C % 8 || x_ld % 8 || dp_ld % 8 || dx_ld % 8 || (dskip && ds_ld % 8)) { set_error("sx_maxpool2_backward_add: need C %% 8 == 0"); return SX_ERR_UNSUPPORTED; }
  const int64_t total = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16) hipLaunchKernelGGL(maxpool_bwd_add_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16*)x, x_ld, (const bf16*)dp, dp_ld, (const bf16*)dskip, ds_ld, (bf16*)dx, dx_ld, B, H, W, C);
  else hipLaunchKernelGGL(maxpool_bwd_add_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, x_ld, (const float*)dp, dp_ld, (const float*)dskip, ds_ld, (float*)dx, dx_ld, B, H, W, C);
  SX_LAUNCH_CHECK("maxpool_bwd_add");
  return SX_OK;
}

extern "C" int sx_maxpool2_backward_add_idx(int dtype, int B, int H, int W, int C, const void* pool_idx, int idx_ld, const void* dp,
                                            int dp_ld, const void* dskip, int ds_ld, void* dx, int dx_ld, void* stream) {
  SX_REQUIRE(pool_idx && dp && dx, "sx_maxpool2_backward_add_idx: null pointer");
  SX_REQUIRE(dtype == SX_F32 || dtype == SX_BF16, "sx_maxpool2_backward_add_idx: bad dtype %d", dtype);
  SX_REQUIRE(idx_ld >= C && dp_ld >= C && dx_ld >= C && (!dskip || ds_ld >= C),
             "sx_maxpool2_backward_add_idx: a leading dimension is smaller than C=%d (idx %d, dp %d, dskip %d, dx %d)", C, idx_ld, dp_ld,
             ds_ld, dx_ld);
  if (C % 8 || idx_ld % 8 || dp_ld % 8 || dx_ld % 8 || (dskip && ds_ld % 8) || (H & 1) || (W & 1) || H < 2 || W < 2 ||
      ((uintptr_t)pool_idx % 8)) {
    set_error("sx_maxpool2_backward_add_idx: need C %% 8 == 0, even H and W, 8-byte aligned positions");
    return SX_ERR_UNSUPPORTED;
  }
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  const unsigned char* ix = (const unsigned char*)pool_idx;
  if (dtype == SX_BF16) hipLaunchKernelGGL(maxpool_bwd_add_idx_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, st, ix, idx_ld, (const bf16*)dp, dp_ld, (const bf16*)dskip, ds_ld, (bf16*)dx, dx_ld, B, H, W, C);
  else hipLaunchKernelGGL(maxpool_bwd_add_idx_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, ix, idx_ld, (const float*)dp, dp_ld, (const float*)dskip, ds_ld, (float*)dx, dx_ld, B, H, W, C);
  SX_LAUNCH_CHECK("maxpool_bwd_add_idx");
  return SX_OK;
}

namespace sx {
int channel_sum_slots(int dtype, int B, int H, int W, int Hs, int Ws, int C, int C_out, const void* t, int ld, float* out, int slots, void* stream);
}
// out must be zeroed by the caller unless accumulating on purpose.
extern "C" int sx_channel_sum(int dtype, int B, int H, int W, int Hs, int Ws, int C, int C_out, const void* t, int ld, float* out, void* stream) {
  return channel_sum_slots(dtype, B, H, W, Hs, Ws, C, C_out, t, ld, out, 1, stream);
}

namespace sx {
// out: [slots][C_out] partial sums (the caller adds the rows).  Every block ends in C_out same-address atomics, ~13 ns each in
// the L2's atomic unit; with one row that caps the grid at a block per CU (4 waves per CU: the level-0 fp32 sums ran at
// 2.3 TB/s), with `slots` rows the grid can be slots times larger for the same queue per address.
int channel_sum_slots(int dtype, int B, int H, int W, int Hs, int Ws, int C, int C_out, const void* t, int ld, float* out, int slots, void* stream) {
  SX_REQUIRE(t && out && Hs <= H && Ws <= W && C > 0 && C_out > 0 && C_out <= C && slots >= 1, "sx_channel_sum: bad arguments");
  if (C % 8 || ld % 8 || C > 2048) { set_error("sx_channel_sum: need C %% 8 == 0 and C <= 2048 (C=%d ld=%d)", C, ld); return SX_ERR_UNSUPPORTED; }
  const int64_t npix = (int64_t)B * Hs * Ws;
  // (>= 128 pixels per block with slot rows: the 32x16 level's 64 blocks of 512 pixels took 19.5 us for 16 MB)
  int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(npix, slots > 1 ? 128 : 512), 256 * std::min(slots, 4)));      // (2048 blocks: 18 us at level 0 against 15 with 1024)
  const int ppb = (int)cdiv64(npix, blocks);
  blocks = (int)cdiv64(npix, ppb);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16) hipLaunchKernelGGL(channel_sum_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)t, ld, B, H, W, Hs, Ws, C, C_out, out, ppb, slots);
  else hipLaunchKernelGGL(channel_sum_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)t, ld, B, H, W, Hs, Ws, C, C_out, out, ppb, slots);
  SX_LAUNCH_CHECK("channel_sum");
  return SX_OK;
}
}  // namespace sx

extern "C" int sx_global_avg_pool(int dtype, int B, int HW, int C, const void* x, int ld, float* z, void* stream) {
  SX_REQUIRE(x && z, "sx_global_avg_pool: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16) hipLaunchKernelGGL(global_avg_pool_kernel<bf16>, dim3(B), dim3(256), 0, st, (const bf16*)x, ld, HW, C, z);
  else hipLaunchKernelGGL(global_avg_pool_kernel<float>, dim3(B), dim3(256), 0, st, (const float*)x, ld, HW, C, z);
  SX_LAUNCH_CHECK("global_avg_pool");
  return SX_OK;
}

namespace sx {
// zero channels [0, C) of pixels with y >= Hin or x >= Win (the F.pad strip of models.py:196-197)
template <typename T>
__global__ void zero_border_kernel(T* __restrict__ t, int ld, int B, int H, int W, int Hin, int Win, int C) {
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    if (y >= Hin || x >= Win) t[(((int64_t)b * H + y) * W + x) * ld + c] = from_f32<T>(0.f);
  }
}
}  // namespace sx

extern "C" int sx_zero_border(int dtype, int B, int H, int W, int Hin, int Win, int C, void* t, int ld, void* stream) {
  SX_REQUIRE(t && Hin <= H && Win <= W, "sx_zero_border: bad arguments");
  if (Hin == H && Win == W) return SX_OK;
  const int64_t total = (int64_t)B * H * W * C;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SX_BF16) hipLaunchKernelGGL(sx::zero_border_kernel<sx::bf16>, dim3(sx::grid_for(total)), dim3(256), 0, st, (sx::bf16*)t, ld, B, H, W, Hin, Win, C);
  else hipLaunchKernelGGL(sx::zero_border_kernel<float>, dim3(sx::grid_for(total)), dim3(256), 0, st, (float*)t, ld, B, H, W, Hin, Win, C);
  SX_LAUNCH_CHECK("zero_border");
  return SX_OK;
}

// ------------------------------------------------------------------ batch assembly from a device-resident dataset
// dst[r][0 .. row_elems) = src[idx[r]][0 .. row_elems): the collate step of a DataLoader over tensors that never leave
// HBM (data.py:380-391 builds the same batch item by item on the host).  16-byte lanes when strides allow.
namespace sx {
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t src_stride, const int64_t* __restrict__ idx,
                                                          int64_t n_src, int64_t row_elems, float* __restrict__ dst, int64_t dst_stride, int vec) {
  const int r = blockIdx.y;
  const int64_t s = idx[r];
  if (s < 0 || s >= n_src) return;          // host validates; never read out of bounds
  const float* sp = src + s * src_stride;
  float* dp = dst + (int64_t)r * dst_stride;
  if (vec) {
    const int64_t n4 = row_elems >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
      reinterpret_cast<f32x4*>(dp)[i] = reinterpret_cast<const f32x4*>(sp)[i];
  } else {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < row_elems; i += (int64_t)gridDim.x * blockDim.x) dp[i] = sp[i];
  }
}
}  // namespace sx

extern "C" int sx_gather_rows(const float* src, int64_t src_stride, int64_t n_src, const int64_t* idx, int n_rows, int64_t row_elems,
                              float* dst, int64_t dst_stride, void* stream) {
  SX_REQUIRE(src && idx && dst, "sx_gather_rows: null pointer");
  SX_REQUIRE(n_rows >= 0 && n_rows <= 65535 && row_elems >= 0 && n_src > 0 && src_stride >= row_elems && dst_stride >= row_elems,
             "sx_gather_rows: bad sizes (rows=%d row_elems=%lld)", n_rows, (long long)row_elems);
  if (n_rows == 0 || row_elems == 0) return SX_OK;
  const int vec = (row_elems % 4 == 0 && src_stride % 4 == 0 && dst_stride % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0) ? 1 : 0;
  const int64_t work = vec ? row_elems / 4 : row_elems;
  const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(work, 256), 64));
  hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, n_rows), dim3(256), 0, (hipStream_t)stream, src, src_stride, idx, n_src, row_elems, dst, dst_stride, vec);
  SX_LAUNCH_CHECK("gather_rows");
  return SX_OK;
}
