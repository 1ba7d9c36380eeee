/*
 * solpex_hip.h -- C ABI of libsolpex_hip.so: hand-written gfx950 (MI355X) HIP kernels for the
 * SOLPEx conditional U-Net hot path (forward/backward, mask-aware loss, optimizer step).
 *
 * The reference (abdoudiaw/Unet, pure Python/PyTorch) has no FFI for this path: its boundary is the
 * Python module API solpex.models / solpex.losses / solpex.train / solpex.predict, which dispatches
 * to ATen ops.  Each entry point below therefore cites the reference *Python call site* whose ATen
 * work it replaces (paths relative to the reference root).  The host-side mirror of that Python API
 * lives in unet_amd/ and binds these symbols with ctypes.
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers unless a
 *    parameter is documented as a host array.  `stream` is a hipStream_t passed as void*.
 *  - Activations are NHWC ("channels last"): element (b, y, x, c) of a tensor with H x W pixels and
 *    leading dimension ld lives at ((b*H + y)*W + x)*ld + c.  ld >= C lets a tensor be a channel
 *    slice of a wider buffer (that is how skip concatenation is made free).
 *  - dtype: SX_F32 or SX_BF16 = storage type of activations and packed weights, and the MFMA input
 *    type.  Accumulation, statistics, parameters, gradients of parameters and optimizer state are
 *    always fp32.
 *  - Every function returns SX_OK (0) or a negative sx_status; sx_last_error() returns a
 *    thread-local message.  Work is enqueued on `stream`; no function synchronises the device,
 *    allocates or frees device memory, or keeps a pointer after it returns.
 */
#ifndef SOLPEX_HIP_H_
#define SOLPEX_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SX_VERSION 100 /* 0.1.0 */

typedef enum sx_status {
  SX_OK = 0,
  SX_ERR_INVALID = -1,     /* bad argument (Python wrapper raises ValueError)            */
  SX_ERR_UNSUPPORTED = -2, /* shape / option outside what the kernels cover (ValueError) */
  SX_ERR_HIP = -3          /* a HIP runtime call failed (RuntimeError)                   */
} sx_status;

typedef enum sx_dtype { SX_F32 = 0, SX_BF16 = 1 } sx_dtype;

/* convolution geometry handled by the implicit-GEMM kernels */
typedef enum sx_conv_mode {
  SX_CONV3 = 0,   /* 3x3, stride 1, zero pad 1            (nn.Conv2d(.,.,3,padding=1), models.py:66,69) */
  SX_CONV1 = 1,   /* 1x1                                   (shortcut, models.py:73-77)                   */
  SX_CONVT_FWD = 2, /* 2x2 stride-2 transposed conv, fprop  (nn.ConvTranspose2d(.,.,2,2), models.py:127)  */
  SX_CONVT_BWD = 3  /* its data gradient (gather form)                                                   */
} sx_conv_mode;

int sx_version(void);
const char* sx_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Weight packing.  Replaces nothing in the reference (ATen consumes (Cout,Cin,kh,kw) directly);
 * the MFMA kernels want K-contiguous, zero-padded tiles in the activation dtype:
 *   fp32: CoutPad = roundup(Cout,16), CinPad = roundup(Cin, 32); in the default split-bf16 arithmetic (below) every
 *         (tap, row, 32-k chunk) holds 32 bf16 hi values followed by the 32 lo values (hi = bf16(w), lo = bf16(w - hi)),
 *         in the exact mode packed[tap][CoutPad][CinPad] fp32 -- the same number of bytes;
 *   bf16: the same elements (CinPad = roundup(Cin, 32)) in 1 KB units of 16 rows x 32 k, unit order
 *         [row/16][k/32][tap], 16-byte group (row, 8 k) at lane ((k%32)/8)*16 + row%16 inside a unit: each unit is one
 *         contiguous wave load in MFMA A-operand order (conv_common.h: wpack_off_bf16).  The layout is private to the
 *         library: treat a packed tensor as opaque.
 * `transpose_flip` != 0 builds the data-gradient operand of a 3x3/1x1 conv instead
 * (roles of Cin/Cout swapped, taps reversed).
 * sx_packed_weight_elems() returns the element count of the packed tensor.
 * --------------------------------------------------------------------------------------------- */
int64_t sx_packed_weight_elems(int mode, int dtype, int c_rows, int c_k);

/* fp32 arithmetic of sx_conv_forward / sx_conv_wgrad (and everything built on them).  Tensors are fp32 in HBM either way.
 *   default (split-bf16): each fp32 operand v is split into bf16 hi + lo on its way into LDS and a product block is three
 *     v_mfma_f32_16x16x32_bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulation): <= 1.1e-5 relative error per product, 4e-6
 *     relative L2 on a whole convolution (TF32, cuDNN's default for the reference's fp32 convolutions, carries 4.9e-4),
 *     at 3/16 of the matrix-core time of exact fp32;
 *   exact: v_mfma_f32_16x16x4_f32 throughout (environment SX_FP32_EXACT=1 at start-up, or this call).
 * Returns the previous setting (1 = exact); on < 0 only queries.  Packed fp32 weights are specific to the mode they were packed in: repack
 * after a switch.  Process-wide; do not switch while calls are in flight on other threads.  No counterpart in the
 * reference (torch.backends.cudnn.allow_tf32 is the closest knob). */
int sx_set_fp32_exact(int on);
/* The same switch with the split form's plane count spelled out: 0 = exact fp32 MFMA, 2 = hi + lo planes (three bf16 MFMAs per
 * product block, ~4e-6 per convolution; the default), 3 = hi + mid + lo (six MFMAs, fp32-grade products).  Any other value only
 * queries.  Returns the previous setting.  Packed fp32 weights depend on it: repack after a switch.  Whatever the setting,
 * the whole-network entry points run networks whose widest layer reaches 512 channels on three planes (SX_FP32_WIDE). */
int sx_set_fp32_planes(int planes);
int sx_pack_conv_weight(int dtype, int mode, int transpose_flip, const float* w, int Cout, int Cin,
                        void* packed, void* stream);

/* ---------------------------------------------------------------------------------------------
 * sx_conv_forward: implicit-GEMM convolution on MFMA (bf16: v_mfma_f32_16x16x32_bf16,
 * fp32: v_mfma_f32_16x16x4_f32), LDS-staged input halo.  Replaces F.conv2d / conv_transpose2d
 * forward and the data-gradient half of convolution_backward for
 *   models.py:66,69 (3x3), :73-77 (1x1 shortcut), :127 (ConvTranspose2d 2x2/2).
 * y[b,oy,ox,co] (+)= sum_{tap,ci} x[b, iy, ix, ci] * packed[tap][co][ci]  (+ bias[co])
 *   H, W      : the iteration grid = output pixels (SX_CONVT_FWD: INPUT pixels, output is 2H x 2W)
 *   Hx, Wx    : pixel dims of the x buffer; Hy, Wy: pixel dims of the y buffer (>= what is written)
 *   gn_stats  : optional fp32 [B][groups][2]; the epilogue atomically adds sum / sum-of-squares of
 *               the fp32 results per (sample, group) (GroupNorm statistics, models.py:67,70)
 *   accumulate: != 0 adds into y instead of overwriting it.
 * --------------------------------------------------------------------------------------------- */
int sx_conv_forward(int dtype, int mode, int B, int H, int W, int Cin, int Cout,
                    const void* x, int x_ld, int Hx, int Wx, const void* packed, const float* bias,
                    void* y, int y_ld, int Hy, int Wy, int accumulate, float* gn_stats, int gn_groups,
                    void* stream);


/* DoubleConv fusions (solpex/models.py:60-96: conv1 and the 1x1 shortcut read the same block input).
 * sx_conv3_shortcut_forward: y = conv3x3(x, packed3) with GroupNorm statistics (as sx_conv_forward) and
 *   s = conv1x1(x, packed1) in one pass over x (bf16 fast-path shapes; otherwise two launches, same results).
 * sx_conv3_shortcut_dgrad: dx = conv3x3^T(dy) + conv1x1^T(dr) where dy = dyr[..., 0:C] and dr = dyr[..., C:2C] are the
 *   channel halves of one NHWC buffer (leading dimension dyr_ld >= 2C); packed3_t / packed1_t are the
 *   transpose_flip = 1 packs (rows = Cin of the forward convolution, K = C).  ch_sum (nullable): ch_sum[c] += sum
 *   over all pixels of dx[..., c] for c < ch_sum_n (fp32 atomics; caller zeroes) -- the bias gradient of the transposed
 *   convolution whose output those channels are (models.py:190,200). */
int sx_conv3_shortcut_forward(int dtype, int B, int H, int W, int Cin, int Cout, const void* x, int x_ld,
                              const void* packed3, const void* packed1, void* y, int y_ld, float* gn_stats,
                              int gn_groups, void* s, int s_ld, void* stream);
int sx_conv3_shortcut_dgrad(int dtype, int B, int H, int W, int C, int Cin, const void* dyr, int dyr_ld,
                            const void* packed3_t, const void* packed1_t, void* dx, int dx_ld, float* ch_sum,
                            int ch_sum_n, void* stream);

/* sx_conv_forward_gn_bwd_sums: a 3x3 / 1x1 convolution whose OUTPUT y is the gradient w.r.t. the output of
 * GroupNorm + SiLU (gn_mode 0: y = dA1 = conv2's data gradient, models.py:66-69) or w.r.t. a DoubleConv block output
 * (gn_mode 2: FiLM, dropout and the residual in front of the second GroupNorm, models.py:80,152-156), with phase 1 of
 * that GroupNorm's backward fused into the epilogue:
 *   sums[b][c] += { sum dz*n_hat, sum dz, sum y*R, sum y }  (as sx_gn_silu_backward; last two for gn_mode 2 only)
 * gn_y = the raw convolution output that GroupNorm normalised (same pixel grid as y), gn_s = the block's shortcut
 * branch (gn_mode 2), stats / gamma / beta / film_g / drop as in sx_gn_silu_backward.  Follow with
 * sx_gn_silu_backward(mode | SX_GN_BWD_SUMS_READY).  Shapes the fused kernel does not cover run the convolution and
 * the reduction as two launches, same results. */
int sx_conv_forward_gn_bwd_sums(int dtype, int mode, int B, int H, int W, int Cin, int Cout, const void* x, int x_ld,
                                const void* packed, void* y, int y_ld, int gn_mode, const void* gn_y, int gn_y_ld,
                                const void* gn_s, int gn_s_ld, const float* stats, int groups, float eps,
                                const float* gamma, const float* beta, const float* film_g, const float* drop,
                                float* sums, void* stream);

/* ---------------------------------------------------------------------------------------------
 * sx_conv_wgrad: weight gradient as a split-K MFMA GEMM over pixels (weight half of
 * aten::convolution_backward for models.py:66,69,73-77,127,132).
 *   dW[tap][a][b] = sum_pixels P[p][a] * Q[q(p,tap)][b]
 *   SX_CONV3 / SX_CONV1: P = dY (Ca = Cout), Q = X (Cb = Cin)        -> dw laid out (Cout,Cin,kh,kw)
 *   SX_CONVT_FWD       : P = X  (Ca = Cin),  Q = dY (Cb = Cout, 2Hx2W) -> dw laid out (Cin,Cout,2,2)
 * Ca/Cb are the (padded) channel counts of the buffers, Ca_out/Cb_out (<= Ca/Cb) the real ones
 * written to dw.  workspace: sx_conv_wgrad_workspace_bytes() bytes of fp32 slabs (summed in a fixed
 * order: deterministic).  accumulate != 0 adds into dw.
 * --------------------------------------------------------------------------------------------- */
int64_t sx_conv_wgrad_workspace_bytes(int mode, int B, int H, int W, int Ca, int Cb);
int sx_conv_wgrad(int dtype, int mode, int B, int H, int W, int Ca, int Cb, int Ca_out, int Cb_out,
                  const void* P, int p_ld, const void* Q, int q_ld, int Hq, int Wq,
                  float* dw, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* sx_conv_wgrad_static: sx_conv_wgrad without the workspace argument, for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has NO counterpart in the
 * reference.  The split-K slabs live in a static device array of SX_WGRAD_STATIC_BYTES that every GPU the library runs
 * on holds until the process ends; the call allocates and frees nothing.  Problems whose slabs
 * (sx_conv_wgrad_workspace_bytes) exceed that size are rejected with SX_ERR_UNSUPPORTED.  All calls on one GPU share
 * the array: calls that overlap on one GPU (different streams or threads) race.
 * Under HIP graph capture: the FIRST call per device takes the array's address (hipGetSymbolAddress) and the first call per
 * kernel instantiation sets a function attribute (and, in fp32, asks the occupancy calculator) -- host calls that must not
 * happen for the first time inside a capture.  Run one call per (shape, dtype) on each device outside capture first. */
#define SX_WGRAD_STATIC_BYTES (32u << 20)
int sx_conv_wgrad_static(int dtype, int mode, int B, int H, int W, int Ca, int Cb, int Ca_out, int Cb_out,
                         const void* P, int p_ld, const void* Q, int q_ld, int Hq, int Wq,
                         float* dw, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm + SiLU and the DoubleConv block output, forward and backward (models.py:64-71,80,
 * 152-156).  stats = fp32 [B][groups][2] from sx_conv_forward.  HW = pixels per sample.
 *   sx_gn_silu_forward : out = SiLU(GN(y))
 *   sx_block_out_forward: R = (SiLU(GN(y2)) + s) * drop ; out = R * film_g + film_b ; r_out = R
 *                         (film_g/film_b/drop: fp32 [B][C] or NULL; r_out may be NULL)
 *   sx_gn_silu_backward: mode 0 through SiLU(GN(y)); mode 1 through the block output (r = the saved R);
 *       mode 2 = mode 1 with r = the shortcut branch s of sx_block_out_forward: R is rebuilt as (SiLU(GN(y)) + s) * drop,
 *       so the forward pass need not store it (r_out = NULL).
 *       sums: zeroed fp32 [B][C][4] = {sum dz*n_hat, sum dz, sum d_out*R, sum d_out} per (b,c)
 *       (entries 2,3 are the FiLM gamma/beta gradients); dy = gradient w.r.t. y; d_r (mode 1,
 *       optional) = d_out*film_g*drop = gradient w.r.t. the shortcut branch; dgamma/dbeta optional.
 * --------------------------------------------------------------------------------------------- */
int sx_gn_silu_forward(int dtype, int B, int HW, int C, int groups, float eps, const void* y, int y_ld,
                       const float* stats, const float* gamma, const float* beta, void* out, int out_ld,
                       void* stream);
int sx_block_out_forward(int dtype, int B, int HW, int C, int groups, float eps, const void* y2, int y_ld,
                         const float* stats, const float* gamma, const float* beta, const void* s, int s_ld,
                         const float* film_g, const float* film_b, const float* drop, void* out, int out_ld,
                         void* r_out, int r_ld, void* stream);
/* sx_block_out_forward followed by nn.MaxPool2d(2) of `out` (models.py:177,181: every encoder block's output is pooled),
 * in one pass over the block; H and W must be even, `pool` is [B][H/2][W/2] with row pitch pool_ld. */
int sx_block_out_pool_forward(int dtype, int B, int H, int W, int C, int groups, float eps, const void* y2, int y_ld,
                              const float* stats, const float* gamma, const float* beta, const void* s, int s_ld,
                              const float* film_g, const float* film_b, const float* drop, void* out, int out_ld,
                              void* r_out, int r_ld, void* pool, int pool_ld, void* stream);
/* The same launch also recording, for every pooled value, its position inside the 2x2 window (0..3 = (dy << 1) | dx, one byte
 * per pooled element in pool_idx[B][H/2][W/2][idx_ld], first maximum wins, a NaN beats any number -- ATen's rule, decided on
 * the value as stored): sx_maxpool2_backward_add_idx routes the gradient with it and need not read the block output again
 * (a third of that launch's traffic).  pool_idx may be null (then this is sx_block_out_pool_forward); idx_ld % 8 == 0,
 * idx_ld >= C, pool_idx 8-byte aligned.  Replaces MaxPool2d's saved indices (`return_indices` inside autograd). */
int sx_block_out_pool_idx_forward(int dtype, int B, int H, int W, int C, int groups, float eps, const void* y2, int y_ld,
                                  const float* stats, const float* gamma, const float* beta, const void* s, int s_ld,
                                  const float* film_g, const float* film_b, const float* drop, void* out, int out_ld,
                                  void* r_out, int r_ld, void* pool, int pool_ld, void* pool_idx, int idx_ld, void* stream);
/* mode | SX_GN_BWD_SUMS_READY: `sums` entries were already accumulated by the producer of d_out
 * (sx_conv_forward_gn_bwd_sums, sx_maxpool2_backward_gn_sums): only the element-wise phase runs.
 * Without the flag, shapes whose (sample, group) unit holds at most 8192 elements (HW * C / groups; C / groups a multiple of
 * 8 and a power-of-two number of 8-channel pieces) run both phases in ONE launch: one block per (sample, group) reads y, d_out
 * (and r) once, WRITES its sums entries (plain stores, no atomics) and applies the gradient from registers; larger shapes run
 * the two-launch form, which ACCUMULATES into `sums` (caller zeroes).  Zeroing `sums` first is correct for both. */
#define SX_GN_BWD_SUMS_READY 4
int sx_gn_silu_backward(int dtype, int mode, int B, int HW, int C, int groups, float eps, const void* y,
                        int y_ld, const float* stats, const float* gamma, const float* beta,
                        const void* d_out, int do_ld, const void* r, int r_ld, const float* film_g,
                        const float* drop, float* sums, void* dy, int dy_ld, void* d_r, int dr_ld,
                        float* dgamma, float* dbeta, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Layout / pooling helpers.
 *   sx_pack_input      : (B,C,H,W) fp32 with element strides (sb,sc,sh,sw) -> NHWC dtype, ld>=C zero padded
 *   sx_unpack_output   : NHWC dtype -> (B,C,H,W) fp32 contiguous
 *   sx_maxpool2_forward: nn.MaxPool2d(2) (models.py:106,177,181)
 *   sx_maxpool2_backward_add: dx = dskip (optional) + max-pool backward of dp (first maximum wins, as ATen)
 *   sx_channel_sum     : out[c] += sum_{b, y<Hs, x<Ws} t[b,y,x,c] for c < C_out <= C (bias gradients; C % 8 == 0;
 *                        caller zeroes out)
 *   sx_global_avg_pool : bottleneck_to_z (models.py:10-12)
 *   sx_zero_border     : zero channels [0,C) of pixels outside Hin x Win (F.pad strip, models.py:196-197)
 * --------------------------------------------------------------------------------------------- */
int sx_pack_input(int dtype, const float* x, int64_t sb, int64_t sc, int64_t sh, int64_t sw, int B, int C,
                  int H, int W, void* out, int out_ld, void* stream);
int sx_unpack_output(int dtype, const void* in, int in_ld, int B, int C, int H, int W, float* out, void* stream);
int sx_maxpool2_forward(int dtype, int B, int H, int W, int C, const void* x, int x_ld, void* y, int y_ld, void* stream);
int sx_maxpool2_backward_add(int dtype, int B, int H, int W, int C, const void* x, int x_ld, const void* dp,
                             int dp_ld, const void* dskip, int ds_ld, void* dx, int dx_ld, void* stream);
/* sx_maxpool2_backward_add with the window positions recorded by sx_block_out_pool_idx_forward instead of the pooled tensor's
 * input (even H and W only): dx = dskip + scatter(dp). */
int sx_maxpool2_backward_add_idx(int dtype, int B, int H, int W, int C, const void* pool_idx, int idx_ld, const void* dp,
                                 int dp_ld, const void* dskip, int ds_ld, void* dx, int dx_ld, void* stream);
int sx_channel_sum(int dtype, int B, int H, int W, int Hs, int Ws, int C, int C_out, const void* t, int ld, float* out, void* stream);
int sx_global_avg_pool(int dtype, int B, int HW, int C, const void* x, int ld, float* z, void* stream);
int sx_zero_border(int dtype, int B, int H, int W, int Hin, int Win, int C, void* t, int ld, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Latent MLPs (latent.py:14-48 ParamToZ / ZToParam and their SmoothL1 trainers, latent.py:51-310), fp32.
 *   sx_gemm_f32 : C[m][n] (+)= epilogue( sum_k A(m,k) * B(n,k) + bias[n] ), A(m,k) = A[m*a_sm + k*a_sk],
 *                 B(n,k) = B[n*b_sn + k*b_sk].  One kernel for nn.Linear forward (A = x, B = W), its input gradient
 *                 (A = dy, B = W^T by strides) and its weight gradient (A = dy^T, B = x^T).  Epilogue, in this order:
 *                 * mul[m][n] (dropout mask), * SiLU'(dact[m][n]), then if act: pre[m][n] = v, v = SiLU(v).
 *                 rowsum (nullable): rowsum[m] (+)= sum_k A(m,k)  -- the bias gradient when A = dy^T.
 *   sx_layernorm_forward / _backward : nn.LayerNorm over the last dimension; stats = [rows][mean, rstd];
 *                 dgamma / dbeta accumulate (fp32 atomics; caller zeroes); dact (nullable) multiplies dx by SiLU'(dact).
 *   sx_smooth_l1 : nn.SmoothL1Loss(beta), mean reduction: *out += mean; dpred (+)= grad_weight * dloss/dpred.
 *   sx_mul       : y = x * m (nn.Dropout with a caller-drawn scaled mask).
 * --------------------------------------------------------------------------------------------- */
int sx_gemm_f32(int M, int N, int K, const float* A, int64_t a_sm, int64_t a_sk, const float* B, int64_t b_sn, int64_t b_sk,
                float* C, int64_t ldc, const float* bias, int act, float* pre, int64_t ld_pre, const float* mul,
                int64_t ld_mul, const float* dact, int64_t ld_dact, float* rowsum, int accumulate, void* stream);
int sx_layernorm_forward(int rows, int N, const float* x, const float* gamma, const float* beta, float eps, float* y,
                         float* stats, void* stream);
int sx_layernorm_backward(int rows, int N, const float* x, const float* gamma, const float* stats, const float* dy,
                          float* dx, float* dgamma, float* dbeta, const float* dact, void* stream);
int sx_smooth_l1(int64_t n, const float* pred, const float* target, float beta, float* out, float* dpred,
                 float grad_weight, int accumulate, void* stream);
int sx_mul(int64_t n, const float* x, const float* m, float* y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Device-resident dataset (replaces the per-item host path of SOLPSDataset.__getitem__ + default collate,
 * data.py:359-391, for tensors that already live in HBM).
 *   sx_gather_rows : dst[r][0..row_elems) = src[idx[r]][0..row_elems) for r < n_rows; idx is a DEVICE int64 array,
 *                    strides in elements; out-of-range indices are skipped (the host wrapper raises IndexError first).
 * --------------------------------------------------------------------------------------------- */
int sx_gather_rows(const float* src, int64_t src_stride, int64_t n_src, const int64_t* idx, int n_rows, int64_t row_elems,
                   float* dst, int64_t dst_stride, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Mask-aware loss (losses.py:60-213), two phases so that data-parallel callers can all-reduce the
 * batch-global denominators in between.  Pixel-loss kinds: 0 huber (smooth L1), 1 l1, 2 mse.
 *   sx_loss_prepare: mask (B,1,H,W) fp32, w (H,W) fp32 or NULL -> den[9] = per scale
 *                    {sum m, sum m*w, sum m_grad}; builds pooled masks in `workspace`.
 *   sx_loss_forward_backward: pred NHWC `dtype` (ld), target planar fp32 (B,C,H,W);
 *                    out[0] = loss, out[1] = sum|pred-target|*mask, out[2] = sum mask (= den[0]: global if the
 *                    caller all-reduced den between the two phases), out[3] = sum mask as sx_loss_prepare formed it on
 *                    THIS rank (single scale; with multiscale > 0 out[3] = out[2])
 *                    (mae_norm, losses.py:226-227, is out[1]/max(out[2],1e-8));
 *                    d_pred (NHWC dtype, ld, channels >= C zeroed) = grad_scale * dloss/dpred, or NULL.
 * --------------------------------------------------------------------------------------------- */
int64_t sx_loss_workspace_bytes(int B, int C, int H, int W, int multiscale, int grad_ds);
int sx_loss_prepare(int B, int C, int H, int W, const float* mask, const float* w, int multiscale,
                    int grad_ds, float lam_grad, void* workspace, float* den, void* stream);
int sx_loss_forward_backward(int dtype, int B, int C, int H, int W, const void* pred, int pred_ld,
                             const float* target, const float* mask, const float* w, const float* cw,
                             int base_kind, float huber_beta, float lam_w, float lam_grad, int grad_magnitude,
                             int grad_kind, int multiscale, float ms_weight, int grad_ds, const float* den,
                             void* workspace, float* out, void* d_pred, int dpred_ld, float grad_scale,
                             void* stream);
/* The same call, also forming the output head's bias gradient on the way out (models.py:203, `out.bias`): bias_sums
 * [SX_HEAD_BIAS_SLOTS][C] fp32 (caller-zeroed; sx_unet_view(SX_T_HEAD_BIAS_SUMS) inside a network workspace) += per-channel
 * sums of d_pred as stored, spread over the slots; *bias_done (HOST int, nullable) = 1 when this configuration's kernels took
 * them (single scale, grad_ds 2, C <= 8, 8-wide rows), else 0 and bias_sums is untouched.  With 1, pass
 * SX_BWD_HEAD_BIAS_READY to sx_unet_decode_backward, which then skips its own pass over d_pred.  No sx_loss_ev_l1 after it:
 * that call adds to d_pred. */
#define SX_HEAD_BIAS_SLOTS 64
int sx_loss_forward_backward_bias(int dtype, int B, int C, int H, int W, const void* pred, int pred_ld,
                                  const float* target, const float* mask, const float* w, const float* cw,
                                  int base_kind, float huber_beta, float lam_w, float lam_grad, int grad_magnitude,
                                  int grad_kind, int multiscale, float ms_weight, int grad_ds, const float* den,
                                  void* workspace, float* out, void* d_pred, int dpred_ld, float grad_scale,
                                  float* bias_sums, int* bias_done, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Output normalizers and masked metrics (data.py:96-110,139-143,157-163,187-223; losses.py:226-253).
 * Normalizer kinds per channel; kinds/mu/sigma/eps/scale are HOST arrays of length C (<= SX_MAX_NORM_CH).
 *   sx_norm_apply      : inverse != 0: normalised -> physical (MultiChannelNormalizer.inverse), else
 *                        .transform.  `in` has element strides (sb,sc,sh,sw); out is (B,C,H,W) fp32.
 *   sx_masked_err_sums : denorm == 0: out += {sum|p-t|*m, sum((p-t)*m)^2, sum m over pixels}  (mae_norm)
 *                        denorm != 0: out += batch_error_sums_ev's {abs_sum, sq_sum, px}.
 * --------------------------------------------------------------------------------------------- */
#define SX_MAX_NORM_CH 32
typedef enum sx_norm_kind { SX_NORM_LOG = 0, SX_NORM_LINEAR = 1, SX_NORM_SYMLOG = 2 } sx_norm_kind;
int sx_norm_apply(int dtype, int inverse, int B, int C, int H, int W, const void* in, int64_t sb, int64_t sc,
                  int64_t sh, int64_t sw, const float* mask, const int* kinds, const float* mu, const float* sigma,
                  const float* eps, const float* scale, float* out, void* stream);
int sx_masked_err_sums(int dtype, int denorm, int B, int C, int H, int W, const void* pred, int64_t sb, int64_t sc,
                       int64_t sh, int64_t sw, const float* target, const float* mask, const int* kinds, const float* mu,
                       const float* sigma, const float* eps, const float* scale, float* out, void* stream);

/* sx_loss_ev_l1: the optional physical-space term of train_unet's loss (train.py:200-205, lam_ev > 0):
 *   loss_ev = sum |inverse(pred)*m - inverse(target)*m| * m / max(sum m, 1e-8)   (numerator over channels, denominator pixels)
 * den = DEVICE pointer to sum(mask) of the whole (global) batch (den[0] of sx_loss_prepare).  out[0] (nullable) += lam * loss_ev;
 * d_pred (nullable; NHWC `dtype`) += lam * grad_scale * d loss_ev / d pred -- call after sx_loss_forward_backward. */
int sx_loss_ev_l1(int dtype, int B, int C, int H, int W, const void* pred, int pred_ld, const float* target,
                  const float* mask, const int* kinds, const float* mu, const float* sigma, const float* eps,
                  const float* scale, const float* den, float lam, float* out, void* d_pred, int dpred_ld,
                  float grad_scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer (train.py:82,226-233): grad-norm, clip_grad_norm_, AdamW over one flat fp32 arena.
 * --------------------------------------------------------------------------------------------- */
/* The gradient norm is summed in a fixed order (bit-identical on every data-parallel rank) through block partials that
 * live in CALLER-OWNED device memory: `parts` = SX_SUMSQ_PARTS floats.  The library keeps no mutable device state for
 * this, so steps that overlap on one GPU (two engines, two streams) cannot see each other's norm.
 * sx_grad_sumsq: out[0] += sum g^2 (caller zeroes out).
 * sx_grad_sumsq_partial: its first half alone; returns the number of partials written (>= 1) or a negative status.
 * sx_adamw_step: gnorm_sq != NULL clips with that sum of squares; gnorm_sq == NULL never clips, whatever max_norm says.
 * sx_adamw_step_fused: clip_grad_norm_ + AdamW in two launches instead of three -- every block of the AdamW launch adds up
 *   the partials of sx_grad_sumsq_partial(g, n, parts) (same stream, just before) itself; nparts must be that call's
 *   return value (checked against n). */
#define SX_SUMSQ_PARTS 256
int sx_grad_sumsq(const float* g, int64_t n, float* out, float* parts, void* stream);
int sx_grad_sumsq_partial(const float* g, int64_t n, float* parts, void* stream);
int sx_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, float max_norm,
                  const float* gnorm_sq, float grad_mul, float* gnorm_out, float* clear_after, void* stream);
int sx_adamw_step_fused(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int step, float max_norm,
                        const float* parts, int nparts, float grad_mul, float* gnorm_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batched inverse problems through the frozen network (scripts/eval_inverse_cycle_conditional_unet.py):
 * every reference quantity that is per case (B = 1) is per SAMPLE here.
 * sx_inverse_fit: weighted_masked_mse (:105-115) and masked_mae (:125-129) per sample + d fit / d pred.
 *   pred NHWC `dtype` (pred_ld); target (B,C,H,W) fp32; mask (B,H,W) fp32; channel_weights[C] (0 for channels outside the
 *   --fields selection); selected[C] (1/0, NULL = all) only affects mae.  fit[B], mae[B] (nullable), d_pred nullable.
 * sx_inverse_adam_step: loss = fit + reg*mean(p^2) -> Adam -> clamp(+-clip) -> early-stop test, for every active sample
 *   (:375-411; cosine learning rate `lr` for this step supplied by the caller, `step` is 1-based). */
int sx_inverse_fit(int dtype, int B, int C, int H, int W, const void* pred, int pred_ld, const float* target,
                   const float* mask, const float* channel_weights, const float* selected, float* fit, float* mae,
                   void* d_pred, int dpred_ld, float grad_scale, void* stream);
int sx_inverse_adam_step(int N, int P, float* p, const float* grad, float* m, float* v, const float* fit, float lr,
                         float beta1, float beta2, float eps, int step, float reg, float clip, int es_patience,
                         float es_tol, float* best_loss, int* wait, int* active, float* loss_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-network entry points: UNet.encode / UNet.decode (models.py:173-203) and their backward,
 * orchestrated in C++ so that a forward or backward pass is ONE call from the host language.
 *
 * params / grads: HOST arrays of device pointers to the fp32 parameters (gradients) in the
 * reference's state_dict order:  encoders.{i}: net.0.weight, net.1.weight, net.1.bias,
 * net.3.weight, net.4.weight, net.4.bias, shortcut.weight; bot: same 7; up_convs.{j}: weight,
 * bias; decoders.{j}: same 7; out.weight, out.bias; then (iff P > 0) film_enc.{i}, film_bot,
 * film_dec.{j}: net.0.weight, net.0.bias, net.2.weight, net.2.bias.  sx_unet_num_params() gives
 * the table length.
 * wpack: sx_unet_wpack_bytes() bytes, filled by sx_unet_pack_weights() whenever weights change.
 * ws   : sx_unet_workspace_bytes() bytes; holds every activation that backward needs, so it must
 *        stay untouched between a forward and its backward.  sx_unet_view() describes the
 *        tensors inside it that callers read or write (skips, bottleneck, prediction, gradients).
 * --------------------------------------------------------------------------------------------- */
typedef struct sx_unet_desc {
  int32_t in_ch, out_ch, base, depth, groups_gn, P, film_hidden;
  int32_t B, H, W;
  int32_t dtype;      /* sx_dtype of activations / packed weights */
  int32_t use_film;   /* 0: no FiLM; 1: FiLM applied (params given and P > 0); 2: as 1, and the caller promises that
                         sx_unet_encode and sx_unet_decode get the SAME film_params: encode then evaluates every
                         generator in one launch and decode none */
} sx_unet_desc;

typedef enum sx_unet_tensor {
  SX_T_INPUT = 0,      /* packed input x, NHWC                                     */
  SX_T_SKIP = 1,       /* index = encoder level                                    */
  SX_T_BOTTLENECK = 2,
  SX_T_PRED = 3,       /* head output, channels padded to a multiple of 8          */
  SX_T_DPRED = 4,      /* gradient w.r.t. PRED (same layout), written by the caller */
  SX_T_DSKIP = 5,      /* gradient w.r.t. skip `index` (written by decode backward) */
  SX_T_DBOTTLENECK = 6,
  SX_T_FILM_DPARAMS = 7, /* fp32 [B][P]: d loss / d params, accumulated by both backward halves */
  SX_T_HEAD_BIAS_SUMS = 9, /* fp32 [SX_HEAD_BIAS_SLOTS][out_ch]: the head's bias-gradient partial sums (zeroed with the backward
                              sums; sx_loss_forward_backward_bias may fill them before sx_unet_decode_backward)           */
  SX_T_DINPUT = 8        /* gradient w.r.t. the input image x (layout of INPUT), written by sx_unet_encode_backward
                            when SX_BWD_INPUT_GRAD is set */
} sx_unet_tensor;

typedef struct sx_view { int64_t offset_bytes; int32_t ld, C, H, W; } sx_view;

int     sx_unet_num_params(const sx_unet_desc* d);
int64_t sx_unet_wpack_bytes(const sx_unet_desc* d);
int64_t sx_unet_workspace_bytes(const sx_unet_desc* d);
int     sx_unet_view(const sx_unet_desc* d, int which, int index, sx_view* out);
int     sx_unet_pack_weights(const sx_unet_desc* d, const float* const* params, void* wpack, void* stream);
/* x: (B,in_ch,H,W) fp32 with element strides; film_params: (B,P) fp32 contiguous or NULL */
int     sx_unet_encode(const sx_unet_desc* d, const float* const* params, const void* wpack, void* ws,
                       const float* x, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                       const float* film_params, void* stream);
/* sx_unet_pack_weights followed by sx_unet_encode in one call (a training step repacks every step: train.py:221-233 change the
 * weights).  The weight packs, the input pack and every FiLM generator run as block ranges of ONE launch where their tables
 * fit the kernel-argument segment (<= 24 paired 3x3 / 1x1 weights, <= 8 others, <= 8 generators); otherwise as the two calls. */
int     sx_unet_pack_encode(const sx_unet_desc* d, const float* const* params, void* wpack, void* ws,
                            const float* x, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                            const float* film_params, void* stream);
/* drop: HOST array of `depth` device pointers to fp32 [B][C] dropout factors (entries or the array may be NULL) */
int     sx_unet_decode(const sx_unet_desc* d, const float* const* params, const void* wpack, void* ws,
                       const float* film_params, const float* const* drop, void* stream);
/* flags for the backward calls */
#define SX_BWD_WEIGHT_GRADS 1  /* compute parameter gradients into `grads` */
#define SX_BWD_ACCUMULATE   2  /* add into `grads` instead of overwriting   */
#define SX_BWD_ZERO_DPARAMS 4  /* zero FILM_DPARAMS first (the first backward half to run sets it) */
#define SX_BWD_DEFER_FILM   8  /* pass to BOTH halves: the decoder's FiLM generators (their parameter gradients and their
                                  share of FILM_DPARAMS) and the decoder half's GroupNorm affine and bias gradients are finished
                                  inside sx_unet_encode_backward (one launch for all of them); film_dec.* and
                                  decoders.*.net.{1,4}.*, up_convs.*.bias, out.bias gradients are complete only after that call */
#define SX_BWD_NO_DPARAMS       64 /* skip FILM_DPARAMS (the gradient w.r.t. the FiLM inputs): a training step does not read it */
#define SX_BWD_INPUT_GRAD      128 /* sx_unet_encode_backward: also back-propagate to the input image (SX_T_DINPUT); the
                                      reference gets it from autograd when x.requires_grad (models.py:205) */
#define SX_BWD_SUMS_CLEARED    256 /* sx_unet_decode_backward (with SX_BWD_DEFER_FILM): the backward sums / bias scratch of this workspace
                                      are already zero because sx_unet_encode ran on it SINCE the last backward with use_film == 2 and
                                      FiLM parameters (its FiLM launch clears them together with the statistics): no memset dispatch.
                                      Ignored (the memset runs) for any other descriptor.                                          */
#define SX_BWD_HEAD_BIAS_READY 512 /* sx_unet_decode_backward (with SX_BWD_SUMS_CLEARED): SX_T_HEAD_BIAS_SUMS already holds the sums of
                                      d_pred (sx_loss_forward_backward_bias): no channel-sum launch                                */
#define SX_BWD_ONLY_BOTTLENECK  16 /* sx_unet_encode_backward only: back-propagate the bottleneck block and finish ITS convolution
                                      weight gradients (bot.net.{0,3}.weight, bot.shortcut.weight), then return ...            */
#define SX_BWD_AFTER_BOTTLENECK 32 /* ... and a second call with this flag does everything else (the encoder levels, every
                                      deferred reduction, the bottleneck's GroupNorm affines, FiLM).  For data-parallel callers:
                                      the all-reduce of the bottleneck weights (40 % of the parameters) can start in between.
                                      Without either flag the call does both parts.                                         */
int     sx_unet_decode_backward(const sx_unet_desc* d, const float* const* params, float* const* grads,
                                const void* wpack, void* ws, const float* film_params,
                                const float* const* drop, int flags, void* stream);
int     sx_unet_encode_backward(const sx_unet_desc* d, const float* const* params, float* const* grads,
                                const void* wpack, void* ws, const float* film_params, int flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SOLPEX_HIP_H_ */
