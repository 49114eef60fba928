/* ssunet_hip.h -- C-ABI of the MI355X (gfx950) hot-path library for ssUnet-GAN training.
 *
 * The reference (ideafisher/ssUnet-GAN) is pure Python on stock PyTorch ops; it has no FFI
 * layer.  The drop-in boundary a maintainer sees is the Python nn.Module / train() surface
 * (the modules under ssunet-gan_amd/ mirror scripts/); THIS header is the build-defined C-ABI one
 * level below it: one entry point per stock torch op the reference's hot
 * path dispatches.  Each entry cites the reference call site(s) whose torch op it replaces
 * (paths relative to the reference's scripts/).
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless noted;
 *   - every call enqueues on `stream` (a hipStream_t passed as void*) and returns at once;
 *   - return value: 0 = ok, otherwise a negative SSG_E* code or a positive hipError_t;
 *     ssg_last_error() returns a thread-local message.  Nothing throws across the ABI;
 *   - the library allocates nothing and keeps no state between calls: callers own all
 *     buffers including workspaces (sizes via the *_workspace_bytes helpers);
 *   - activations are fp32 "NHWC with pixel stride": element (n,y,x,c) of a tensor with
 *     H x W pixels lives at ((n*H + y)*W + x)*ld + c, ld >= C, ld % 4 == 0, 16-B aligned.
 */
#ifndef SSUNET_HIP_H
#define SSUNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSG_OK 0
#define SSG_EINVAL (-1)   /* bad argument / unsupported shape */
#define SSG_EALIGN (-2)   /* pointer or stride alignment */

#define SSG_ACT_NONE 0
#define SSG_ACT_RELU 1
#define SSG_ACT_LRELU 2
#define SSG_ACT_SWISH 3   /* x*sigmoid(x) (efficientnet_pytorch/utils.py:37-48); batch-norm apply / backward only */

#define SSG_MAX_TAPS 9

const char* ssg_last_error(void);
int ssg_abi_version(void);

/* ------------------------------------------------------------------ convolution (MFMA)
 * Implicit-GEMM convolution, fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
 * Replaces F.conv2d forward and its input-gradient at: archs.py:210,212,218 (BasicBlock
 * 3x3 + 1x1 shortcut), archs.py:593-615 (1x1 heads, final), normalization.py:90-96 (SPADE
 * 3x3), models_seg_gan.py:37-39 (discriminator 3x3 s1/s2), models_seg_gan.py:281-283
 * (fc1/fc2 as 1x1 on a 1x1 grid).  torch.cat at archs.py:651-667 is absorbed by the
 * second input pointer.
 *
 * One launch computes, for every pixel (n,gy,gx) of a GH x GW grid and every co < Cout:
 *   acc = sum_t sum_c  in[n, gy*in_sy + dy[t], gx*in_sx + dx[t], c] * w[co, k(t,c)]
 * (out-of-image taps read 0), then  v = acc (+bias[co]) (+res[pixel,co]) -> act -> out at
 * pixel (gy*out_sy + out_oy, gx*out_sx + out_ox) of an OH x OW image.  `in` is the
 * channel-concatenation of in1 (C1 channels) and in2 (C2 channels, may be NULL/0).
 * Forward conv, stride-1 dgrad (taps mirrored) and stride-2 dgrad (4 parity launches)
 * are all instances of this.
 *
 * Weight operand `w` is a packed matrix [Cout][Kp] (row stride Kp floats, Kp % 16 == 0)
 * produced by ssg_pack_weights, whose K order must match `kmode`:
 *   kmode 0 (requires Cin % 16 == 0): k = (c/16)*ntaps*16 + t*16 + c%16
 *   kmode 1 (Cin % 4 == 0)          : k = t*Cin + c, zero padded to Kp
 */
typedef struct {
  const float* in1; const float* in2;
  int C1, C2, ld1, ld2;
  int N, H, W;              /* input image size */
  const float* w; int Kp; int kmode;
  const float* bias;        /* [Cout] or NULL */
  const float* res; int ldr;/* residual, indexed like out, or NULL */
  float* out; int Cout, ldo;
  int GH, GW;               /* pixel grid of this launch */
  int OH, OW;               /* output image size */
  int in_sy, in_sx, out_sy, out_sx, out_oy, out_ox;
  int ntaps; int dy[SSG_MAX_TAPS]; int dx[SSG_MAX_TAPS];   /* each in [-2, 5] */
  int act; float slope;
  double* bnpart;           /* optional batch-norm statistics of the OUTPUT (acc + bias, before res / act): one row [2][Cout]
                             * (sum, sum of squares; fp64) per M-tile, ssg_conv2d_bnpart_rows(d) rows; NULL = off */
  float* ws; int64_t ws_bytes;  /* optional split-K workspace (ssg_conv2d_workspace_bytes(d) bytes, 16-byte aligned); NULL = never split */
  const void* w_split;      /* optional: the same weights split into three bf16 terms by ssg_pack_weights_split_bf16x3 for the column
                             * tile ssg_conv2d_split_bn(d) reports; non-NULL selects the split-operand kernel (below).  NULL = fp32 MFMA */
  int parity_merge;         /* 1 = the launch is the whole input gradient of a 3x3 stride-2 pad-1 conv (models_seg_gan.py:37-39, s2 blocks):
                             * its 9 taps are the four output-parity classes in the order (0,0) | (0,1) x2 | (1,0) x2 | (1,1) x4, tap t of
                             * class (py, px) accumulates into output pixel (2 gy + py, 2 gx + px); GH x GW = ceil(OH/2) x ceil(OW/2),
                             * out_sy = out_sx = 2, out_oy = out_ox = 0, no bias / res / bnpart, w_split for a 64-column tile.  Only
                             * where ssg_conv2d_split_bn(d) returns 64 for such a descriptor; otherwise the caller launches the classes
                             * one by one (parity_merge = 0, ntaps = 1 / 2 / 2 / 4, out_oy / out_ox = the class).  0 = plain conv. */
  /* ABI 8: fused batch-norm apply on the INPUT (archs.py:229-230, relu(bn1(conv1(x))) feeding conv2): the launch convolves
   * act_in(in1[p][c] * in_scale[c] + in_shift[c]) -- the tensor ssg_bn_apply_f32 would have written, same fp32 fma, never
   * materialised; zero padding applies to the ACTIVATED tensor.  in_scale = NULL: off.  Only where ssg_conv2d_in_affine_ok(d)
   * returns 1 (split-operand k32 kernels, one input pointer, C1 <= 512, in_act none / ReLU / leaky ReLU). */
  const float* in_scale; const float* in_shift; int in_act; float in_slope;
  /* ABI 9: batch-norm BACKWARD statistics in the epilogue of the input-gradient launch that produces the gradient of an activation
   * y = act(x * scale[c] + shift[c]) (archs.py:229-230: the dgrad of conv2 produces d(relu(bn1(c1)))): the launch multiplies its result by
   * act'(.) recomputed from bwd_x (= c1, same shape and pixel order as `out`, pixel stride bwd_ldx) with bwd_scale / bwd_shift, WRITES THE
   * MASKED GRADIENT g, and fills `bnpart` (which must be set) with per-tile rows (sum g, sum g * (x - bwd_mean[c])) instead of the forward's
   * (sum, sum of squares): what ssg_bn_bwd_reduce_f32 would have read g and x again for.  bwd_x = NULL: off.  Only where
   * ssg_conv2d_bwd_stats_ok(d) returns 1 (the wide split-operand k32 tiles; no res, act, in_scale). */
  const float* bwd_x; int bwd_ldx; const float* bwd_scale; const float* bwd_shift; const float* bwd_mean; int bwd_act; float bwd_slope;
} ssg_conv_desc;

/* fp32 convolution on the bf16 matrix pipe (3x3, unit stride; archs.py:210,212 and their input gradients, models_seg_gan.py:37-39
 * stride-1 blocks): every fp32 operand is x = x1 + x2 + x3 with bf16 terms (24 significand bits, exact residuals) and the six
 * products of order >= 2^-16 go through v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- one fp32 ulp per product is left
 * out, results agree with the fp32-MFMA kernel to fp32 accumulation accuracy (tests/test_split_gpu.py), at 16/6 of its
 * matrix rate.  ssg_conv2d_split_bn: 0 when the launch for `d` has no split-operand kernel, else the column tile (64 / 128)
 * its weights must be split for: `w_split` = ssg_pack_weights_split_bf16x3(d->w rows [R][Kp] in kmode 0, R, Kp, BN) --
 * ssg_pack_weights_split_bytes bytes, layout [ceil(R/BN)][Kp/16][BN][96 B].
 * ABI 7 (round 4): ssg_conv2d_split_bn may also return 1128 or 1064 -- the launch goes to the 32-channel-chunk kernel on
 * v_mfma_f32_16x16x32_bf16 (conv_igemm_halo_k32.hip: 8 x 32-pixel x 128-channel tiles of 512 threads, or 4 x 32 x 64 of 256), whose
 * weights are packed by the same two functions with BN = that code: layout [R/bn][Kp/32 steps = chunk32 * 9 + tap][bn/16
 * fragments][3 planes][64 lanes][16 B] (bn = code - 1000; R % bn == 0, Kp % 288 == 0).  A third code, 2064 (16 x 32-pixel x 64-channel
 * tiles of 512 threads), reads the pack of 1064: pack with BN = 1064.  ssg_conv_set_k32_mode(0 / 1 / 2): never /
 * where the grid fills the chip (default, SSG_K32) / wherever the shape is legal (tests). */
int ssg_conv2d_split_bn(const ssg_conv_desc* d);
int ssg_conv2d_in_affine_ok(const ssg_conv_desc* d);   /* ABI 8: 1 when the launch for `d` (w_split set) takes in_scale / in_shift */
int ssg_conv2d_bwd_stats_ok(const ssg_conv_desc* d);   /* ABI 9: 1 when the launch for `d` (w_split set) takes bwd_x ... */
int ssg_conv_set_k32_mode(int mode);
int64_t ssg_pack_weights_split_bytes(int R, int Kp, int BN);
int ssg_pack_weights_split_bf16x3(const float* w_packed, int R, int Kp, int BN, void* out, void* stream);

/* Launches whose pixel-tile count leaves most of the chip idle (the 16x16 / 32x32 levels of archs.py:583-589, the
 * Cout <= 64 input gradients of SPADE's gamma|beta conv normalization.py:94-96, batch-1 inference api.py:322) split the
 * reduction over 16-channel chunks into slabs; an ordered second stage adds the slabs and applies bias / res / act
 * (bitwise reproducible).  Returns 0 when the launch for `d` would not split. */
int64_t ssg_conv2d_workspace_bytes(const ssg_conv_desc* d);

int ssg_conv2d_igemm_f32(const ssg_conv_desc* d, void* stream);
/* Dispatching entry point (what the host side calls): convs with <= 8 channels on one side and
 * unit strides (SPADE's C->3->h->C chain normalization.py:90-96, the 3-channel image / logit /
 * mask layers archs.py:210,615, models_seg_gan.py:37) run on HBM-bound VALU kernels; everything
 * else goes to the MFMA implicit GEMM above.  Same descriptor, same semantics.
 * ssg_conv2d_kernel_id: 0..2 = conv_igemm_kernel<128,128>/<256,64>/<256,32> (register-staged),
 * 30/31/32 = conv_igemm_halo_kernel<128,128>/<256,64>/<128,64> (LDS-resident halo tile: the default for the 9 taps of
 * a 3x3 window at unit stride), 33/34 = conv_igemm_halo16_kernel<128,128>/<128,64> (the same on 8x16-pixel tiles, images <= 16 wide), 20/21/22 = conv_igemm_dma_kernel<128,128>/<256,64>/<128,64> (LDS-DMA pipeline, the default for Cin % 16 == 0
 * and Cout > 32), 12/13 = thin4 kernels on the 4x4x1 MFMA (4-channel input / Cout <= 4 with Cin % 64 == 0),
 * 10 = thin small-Cout (VALU; profiling labels). */
int ssg_conv2d_f32(const ssg_conv_desc* d, void* stream);
int ssg_conv2d_kernel_id(const ssg_conv_desc* d);
/* Rows of `bnpart` the launch for `d` writes, or 0 when the kernel `d` maps to has no statistics epilogue (the caller then
 * runs ssg_bn_stats_f32).  The batch norm that follows the conv (archs.py:211,213; models_seg_gan.py:43) takes its
 * (sum x, sum x^2) from these rows (ssg_bn_stats_from_partials_f32): one full read of the conv output less. */
int ssg_conv2d_bnpart_rows(const ssg_conv_desc* d);

/* Weight packing from the reference's OIHW parameter layout [O][I][KH][KW] (the
 * state_dict layout of nn.Conv2d, archs.py:210 etc.) into the [R][Kp] operand above.
 *   transpose = 0: rows R = O, reduce over I  (forward)
 *   transpose = 1: rows R = I, reduce over O  (input gradient)
 * tap t reads kernel position (ky[t], kx[t]).  `Cred_pad` is the reduced-channel count
 * rounded up to a multiple of 4 (pad channels get zero weights). */
int ssg_pack_weights_f32(const float* w_oihw, int O, int I, int KH, int KW,
                         int transpose, int ntaps, const int* ky, const int* kx,
                         int kmode, int Cred_pad, int Kp, float* out, void* stream);
/* The same with every weight multiplied by 1 / *sigma (device scalar): a spectrally normalised conv (spectral_norm.py:86-88)
 * packs `weight_orig` with its sigma, so W / sigma is never materialised in OIHW form. */
int ssg_pack_weights_scaled_f32(const float* w_oihw, int O, int I, int KH, int KW,
                                int transpose, int ntaps, const int* ky, const int* kx,
                                int kmode, int Cred_pad, int Kp, const float* sigma, float* out, void* stream);

/* Weight gradient (replaces conv2d's weight-grad at the same call sites).
 *   dw[co, c, ky[t], kx[t]] = sum_{n,gy,gx} dout[n,gy,gx,co] * in[n, gy*in_sy+dy[t], gx*in_sx+dx[t], c]
 * Two deterministic stages: split-K partial slabs into `ws`, then an ordered reduce that
 * writes the OIHW gradient (only c < Cin_real channels; pad channels are dropped).
 * The bias gradient is ssg_channel_sum_f32 over dout. */
typedef struct {
  const float* in1; const float* in2; int C1, C2, ld1, ld2;
  int N, H, W;
  const float* dout; int Cout, ldd; int GH, GW;
  int in_sy, in_sx;
  int ntaps; int dy[SSG_MAX_TAPS]; int dx[SSG_MAX_TAPS]; int ky[SSG_MAX_TAPS]; int kx[SSG_MAX_TAPS];
  int KH, KW, Cin_real;
  float* dw_oihw;           /* [Cout][Cin_real][KH][KW] */
  float* ws; int64_t ws_bytes;
  int flags;                /* bit 0: 3x3 stride-1 weight gradients multiply on the bf16 matrix pipe with both fp32 operands split into
                             * three bf16 terms (wgrad_halo_x3_kernel: fp32-class accuracy, see ssg_conv_desc.w_split); 0 = fp32 MFMA */
  /* ABI 8: the x operand is act_in(in1 * in_scale[c] + in_shift[c]) (see ssg_conv_desc.in_scale): the weight gradient of a conv
   * whose input was a fused batch-norm apply.  NULL = off; only where ssg_conv2d_wgrad_in_affine_ok(d) returns 1. */
  const float* in_scale; const float* in_shift; int in_act; float in_slope;
} ssg_wgrad_desc;

int64_t ssg_conv2d_wgrad_workspace_bytes(const ssg_wgrad_desc* d);
int ssg_conv2d_wgrad_in_affine_ok(const ssg_wgrad_desc* d);   /* ABI 8 */
int ssg_conv2d_wgrad_f32(const ssg_wgrad_desc* d, void* stream);
/* 0..2 = wgrad_kernel<128,128>/<128,64>/<128,32>, 30/31 = wgrad_halo_kernel<32,128>/<64,64> (3x3 stride-1
 * window kept in LDS; default for those), 20/21 = wgrad_dma_kernel<128,128>/<128,64> (default
 * for Cout > 32), 40/41 = wgrad_halo_x3_kernel<32,128>/<64,64> (the halo kernels with split operands, flags bit 0),
 * 15/16 = wgrad4_kernel (4x4x1 MFMA: dout <= 4 channels / in = 4 channels, the default for
 * those shapes), 13/14 = the opt-in VALU variants */
int ssg_conv2d_wgrad_kernel_id(const ssg_wgrad_desc* d);
/* 60 = wgrad_k32_kernel (conv_wgrad_k32.hip, round 4): the split-operand weight gradient of 3x3 stride-1 convs on
 * v_mfma_f32_16x16x32_bf16 (64-channel multiples on every side; flags bit 0).  ssg_wgrad_set_k32_mode(0 / 1) switches it off / on
 * at run time (default on, SSG_WGRAD_K32). */
int ssg_wgrad_set_k32_mode(int mode);

/* Attention gate of AttUNet (archs.py:115-144, `x * psi`): y[p][c] = x[p][c] * sigmoid(g[p]) with one gate
 * logit per pixel (g is the 1-channel BatchNorm output, pixel stride ldg); the sigmoid is folded in.
 * backward: dx = dy * s, dg[p] = (sum_c dy*x) * s * (1 - s) (written as a 4-float pixel: value, 0, 0, 0). */
int ssg_pixel_gate_fwd_f32(const float* x, int ldx, const float* g, int ldg, int64_t P, int C, float* y, int ldy, void* stream);
int ssg_pixel_gate_bwd_f32(const float* x, int ldx, const float* g, int ldg, const float* dy, int lddy, int64_t P, int C,
                           float* dx, int lddx, float* dg, int lddg, void* stream);

/* Skinny fully-connected forward (replaces nn.Linear's forward at models_seg_gan.py:281-283, fc1/fc2 of the
 * discriminator): y[n][o] = act(sum_k x[n][k] * w[o][k] + bias[o]).  `w` is the parameter itself ([O][K] row
 * major, no packing); the weight matrix is streamed once (HBM-bound), K slices are summed in order by a
 * finishing kernel (deterministic).  k % 4 == 0, ldx % 4 == 0, x and w 16-B aligned.  The input and weight
 * gradients stay on ssg_conv2d_f32 / ssg_conv2d_wgrad_f32 (1x1 conv on a 1 x n image). */
int64_t ssg_linear_fwd_workspace_bytes(int n, int k, int o);
int ssg_linear_fwd_f32(const float* x, int n, int k, int ldx, const float* w, int o, const float* bias, int act,
                       float slope, float* y, int ldy, float* ws, int64_t ws_bytes, void* stream);
/* dw[o][k] = sum_n dy[n][o] * x[n][k] (the weight gradient of nn.Linear, models_seg_gan.py:281-283), contiguous [O][K] output,
 * summed over n in order; x rows of stride ldx (k % 4 == 0, 16-byte aligned), dy rows of stride ldd. */
int ssg_linear_wgrad_f32(const float* x, int n, int k, int ldx, const float* dy, int o, int ldd, float* dw, void* stream);

/* ------------------------------------------------------------------ bf16 pointwise convolution (BASELINE config 4)
 * SURVEY.md 8(b) `conv2d_{fwd,dgrad,wgrad}_nhwc_bf16` for the 1x1 layers of the EfficientNet MBConv blocks
 * (efficientnet_pytorch/model.py:40-58: _expand_conv, _project_conv; model.py:172 _conv_head): bf16 operands on
 * v_mfma_f32_32x32x16_bf16, fp32 accumulation.  bf16 activations are NHWC-with-stride tensors of 16-bit elements whose
 * channel count and pixel stride are multiples of 8 (16-byte rows).  Weights stay fp32 in the module; per use they are
 * packed to bf16 rows [rows_pad][Kp] (rows_pad % 128 == 0, Kp % 32 == 0, zero padded): transpose 0 = [O][I] for the
 * forward, transpose 1 = [I][O] for the input gradient.
 *   ssg_gemm_bf16        out[p][n] = sum_k x[p][k] * w[n][k] (+ res[p][n]), out/res bf16      (forward and dgrad)
 *   ssg_gemm_wgrad_bf16  dw[m][n]  = sum_p dy[p][m] * x[p][n], fp32 [M][N] (= OIHW of a 1x1 conv); split-K over pixels
 *                        with fp32 slabs in `ws` and an ordered reduce (bitwise reproducible) */
int ssg_pack_weights_bf16(const float* w, int O, int I, int transpose, int rows_pad, int Kp, void* out, void* stream);
int ssg_gemm_bf16(const void* x, int64_t P, int K, int ldx, const void* w_packed, int Kp, int N,
                  const void* res, int ldr, void* out, int ldo, void* stream);
int64_t ssg_gemm_wgrad_bf16_workspace_bytes(int64_t P, int M, int N);
int ssg_gemm_wgrad_bf16(const void* dy, int ldd, const void* x, int ldx, int64_t P, int M, int N, float* dw,
                        void* ws, int64_t ws_bytes, void* stream);

/* bf16 dense k x k conv of a <= 4-channel image (the EfficientNet stem, efficientnet_pytorch/model.py:162,206: 3 -> 32..48, k = 3, s = 2,
 * TF-SAME padding utils.py:123-146): SURVEY.md 8(b) `conv2d_{fwd,dgrad,wgrad}_nhwc_bf16`, k = 3.  x: fp32 NHWC image with 4-channel pixel
 * rows (the model input); image values and weights are rounded to bf16 before they multiply, fp32 accumulation, bf16 output [N, OH, OW, Cout]
 * (Cout % 8 == 0).  w_oihw / dw_oihw: fp32 [Cout][Cin][KH][KW].  wgrad: ws of ssg_conv2d_thin_bf16_wgrad_workspace_bytes bytes.
 * dgrad writes the fp32 image gradient [N, H, W, 4] (channels >= Cin: zeros). */
int ssg_conv2d_thin_bf16(const float* x, int N, int H, int W, int ldx, const float* w_oihw, int Cout, int Cin, int KH, int KW,
                         int stride, int pad_t, int pad_l, int OH, int OW, void* y, int ldy, void* stream);
int64_t ssg_conv2d_thin_bf16_wgrad_workspace_bytes(int N, int OH, int OW, int Cout, int KH, int KW);
int ssg_conv2d_thin_bf16_wgrad(const float* x, int N, int H, int W, int ldx, const void* dy, int lddy, int Cout, int Cin,
                               int KH, int KW, int stride, int pad_t, int pad_l, int OH, int OW, float* dw_oihw, void* ws,
                               void* stream);
int ssg_conv2d_thin_bf16_dgrad(const void* dy, int lddy, int N, int H, int W, const float* w_oihw, int Cout, int Cin, int KH, int KW,
                               int stride, int pad_t, int pad_l, int OH, int OW, float* dx, int lddx, void* stream);

/* bf16 twins of the HBM-bound kernels the MBConv block needs (same arguments and arithmetic as their _f32 namesakes below:
 * tensors are bf16 in HBM, arithmetic is fp32 in registers, statistics fp64; C % 4 == 0, 8-byte loads per lane).
 * Batch norm with act = SSG_ACT_SWISH fuses `swish(bn(x))` (model.py:75,80) into the apply pass; its backward recomputes the
 * pre-activation from x and (scale, shift) (y must be NULL), so the BN output before the swish is never stored. */
int ssg_bn_stats_bf16(const void* x, int64_t P, int C, int ld, double* sums, int with_count, void* ws, void* stream);
int ssg_bn_apply_bf16(const void* x, int64_t P, int C, int ld, const float* scale, const float* shift,
                      const void* res, int ldr, int act, float slope, void* y, int ldy, void* stream);
int ssg_bn_bwd_reduce_bf16(const void* x, const void* y, const void* dy, int64_t P, int C, int ldx, int ldy, int lddy,
                           const float* mean, const float* invstd, const float* scale, const float* shift,
                           int act, float slope, double* sums, int with_count, void* ws, void* stream);
int ssg_bn_bwd_apply_bf16(const void* x, const void* y, const void* dy, int64_t P, int C, int ldx, int ldy, int lddy,
                          const float* mean, const float* invstd, const float* weight, const float* scale, const float* shift,
                          const double* sums, double count, int act, float slope, void* dx, int lddx, void* dres, int lddres,
                          float* dweight, float* dbias, void* stream);
int ssg_dwconv2d_fwd_bf16(const void* in, int N, int H, int W, int C, int ld, const float* w, const float* bias, int KH, int KW,
                          int stride, int pad_top, int pad_left, int OH, int OW, void* out, int ldo, void* stream);
int ssg_dwconv2d_dgrad_bf16(const void* dout, int lddo, int N, int H, int W, int C, const float* w, int KH, int KW, int stride,
                            int pad_top, int pad_left, int OH, int OW, void* dx, int lddx, void* stream);
int ssg_dwconv2d_wgrad_bf16(const void* in, int N, int H, int W, int C, int ld, const void* dout, int lddo, int KH, int KW,
                            int stride, int pad_top, int pad_left, int OH, int OW, float* dw, void* ws, void* stream);
int ssg_channel_scale_fwd_bf16(const void* x, int ldx, const float* s, int N, int64_t S, int C, void* y, int ldy, void* stream);
int ssg_sample_channel_sum_bf16(const void* a, int lda, const void* b, int ldb, int N, int64_t S, int C, float scale,
                                float* out, void* ws, void* stream);
int ssg_broadcast_rows_bf16(const float* s, int N, int64_t S, int C, float scale, void* y, int ldy, void* stream);
int ssg_add_bf16(const void* a, int lda, const void* b, int ldb, int64_t P, int C, void* out, int ldo, void* stream);
int ssg_convert_f32_to_bf16(const float* src, int ldsrc, int64_t P, int C, void* dst, int lddst, void* stream);
int ssg_convert_bf16_to_f32(const void* src, int ldsrc, int64_t P, int C, float* dst, int lddst, void* stream);

/* ------------------------------------------------------------------ layout helpers
 * NCHW (the reference's layout at the boundary: dataset.py:144 tensors, G logits) <->
 * internal NHWC-with-stride.  Pad channels [C, ld) are written as 0. */
int ssg_nchw_to_nhwc_f32(const float* src, int N, int C, int H, int W, float* dst, int ld, void* stream);
int ssg_nhwc_to_nchw_f32(const float* src, int ld, int N, int C, int H, int W, float* dst, void* stream);

/* ------------------------------------------------------------------ batch norm (HBM-bound)
 * Training-mode BatchNorm2d (archs.py:211,213; models_seg_gan.py:43), eps/momentum as the
 * module holds them.  Split so a cross-rank all-reduce of the partial sums (sync-BN,
 * batchnorm.py:63-64,104-107) fits between stage 1 and stage 2.
 *   stage 1: sums[0:C] = sum_p x[p,c], sums[C:2C] = sum_p x[p,c]^2  (fp64, deterministic)
 *   stage 2: mean, invstd; scale = w*invstd, shift = b - mean*scale; running stats update
 *            var_mode 0: invstd = 1/sqrt(var_b + eps)        (torch, batchnorm.py:52-55)
 *            var_mode 1: invstd = clamp(var_b, eps)^-1/2     (sync branch, batchnorm.py:127)
 *   apply  : y = x*scale[c] + shift[c] (+res) -> act
 * Sync-BN with unequal local batches: with_count != 0 makes stage 1 also write sums[2C] = P (so `sums` holds 2C+1
 * doubles and the rank's pixel count is all-reduced with the sums); stage 2 and the backward apply take the count from
 * sums[2C] when their `count` argument is <= 0.
 */
int64_t ssg_bn_workspace_bytes(int64_t P, int C);
int ssg_bn_stats_f32(const float* x, int64_t P, int C, int ld, double* sums, int with_count, void* ws, void* stream);
/* stage 1 from the conv epilogue's per-tile rows (ssg_conv_desc.bnpart); count > 0 also writes sums[2C] = count */
int64_t ssg_bn_stats_from_partials_workspace_bytes(int rows, int C);
int ssg_bn_stats_from_partials_f32(const double* part, int rows, int C, double* sums, double count, void* ws, void* stream);
int ssg_bn_finalize_f32(const double* sums, double count, int C, const float* weight, const float* bias,
                        float eps, float momentum, int var_mode,
                        float* running_mean, float* running_var,
                        float* mean, float* invstd, float* scale, float* shift, void* stream);

/* Local (un-synchronised) batch norm in one call: the statistics passes finish with the per-channel constants and the running
 * estimates of ssg_bn_finalize_f32 (same arithmetic, same bits), one launch less per batch-norm forward
 * (archs.py:211,213, models_seg_gan.py:43, efficientnet_pytorch/model.py:75,80,90).  `count` of the from-partials form =
 * pixels the partial rows cover. */
typedef struct ssg_bn_fin {
  const float* weight; const float* bias;     /* may be NULL (1 / 0) */
  float eps, momentum; int var_mode;          /* var_mode as ssg_bn_finalize_f32 */
  float* running_mean; float* running_var;    /* may be NULL */
  float* mean; float* invstd; float* scale; float* shift;
} ssg_bn_fin;
int ssg_bn_stats_finalize_f32(const float* x, int64_t P, int C, int ld, const ssg_bn_fin* fin, void* ws, void* stream);
int ssg_bn_stats_finalize_bf16(const void* x, int64_t P, int C, int ld, const ssg_bn_fin* fin, void* ws, void* stream);
int ssg_bn_stats_from_partials_finalize_f32(const double* part, int rows, int C, double count, const ssg_bn_fin* fin, void* ws, void* stream);
int ssg_bn_apply_f32(const float* x, int64_t P, int C, int ld, const float* scale, const float* shift,
                     const float* res, int ldr, int act, float slope, float* y, int ldy, void* stream);
/* backward: g = dy masked by the activation (y>0 ? 1 : slope), dres = g (if wanted);
 *   stage 1: sums[0:C] = sum g, sums[C:2C] = sum g*xhat   (fp64; all-reduced for sync-BN)
 *   apply  : dx = scale*(g - sums0/count - xhat*sums1/count); dweight = sums1, dbias = sums0
 * The activation mask comes from the saved output y, or -- y = NULL, for a forward WITHOUT residual -- is
 * recomputed from x with the (scale, shift) the forward used (one tensor read less per kernel). */
int ssg_bn_bwd_reduce_f32(const float* x, const float* y, const float* dy, int64_t P, int C,
                          int ldx, int ldy, int lddy, const float* mean, const float* invstd,
                          const float* scale, const float* shift,
                          int act, float slope, double* sums, int with_count, void* ws, void* stream);
int ssg_bn_bwd_apply_f32(const float* x, const float* y, const float* dy, int64_t P, int C,
                         int ldx, int ldy, int lddy, const float* mean, const float* invstd,
                         const float* weight, const float* scale, const float* shift,
                         const double* sums, double count, int act, float slope,
                         float* dx, int lddx, float* dres, int lddres,
                         float* dweight, float* dbias, void* stream);
/* eval-mode / generic per-channel affine: y = x*scale + shift -> act (also used for bias+act) */

/* ------------------------------------------------------------------ pool / unpool / upsample
 * MaxPool2d(2,2,return_indices) archs.py:571,628-643; MaxUnpool2d(2,2) archs.py:572,648-659;
 * Upsample(x2, bilinear, align_corners=True) archs.py:573,664,667.  The window argmax is kept
 * as one byte (0..3 = dy*2+dx) per output element; ties resolve to the first in scan order
 * and NaN wins, as ATen's CPU max_pool2d does. */
int ssg_maxpool2x2_fwd_f32(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy, uint8_t* idx, void* stream);
int ssg_maxpool2x2_bwd_f32(const float* dy, int lddy, const uint8_t* idx, int N, int H, int W, int C, float* dx, int lddx, void* stream);
/* dx = res + maxpool backward(dy): the gradient of an encoder output that feeds both the pool and a skip connection
 * (archs.py:628-667) in one pass; res is laid out like dx. */
int ssg_maxpool2x2_bwd_add_f32(const float* dy, int lddy, const uint8_t* idx, const float* res, int ldr, int N, int H, int W, int C,
                               float* dx, int lddx, void* stream);
int ssg_maxunpool2x2_fwd_f32(const float* x, int ldx, const uint8_t* idx, int N, int OH, int OW, int C, float* y, int ldy, void* stream);
int ssg_maxunpool2x2_bwd_f32(const float* dy, int lddy, const uint8_t* idx, int N, int OH, int OW, int C, float* dx, int lddx, void* stream);
int ssg_upsample2x_bilinear_fwd_f32(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy, void* stream);
int ssg_upsample2x_bilinear_bwd_f32(const float* dy, int lddy, int N, int H, int W, int C, float* dx, int lddx, void* stream);
int ssg_upsample2x_nearest_fwd_f32(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy, void* stream);
int ssg_upsample2x_nearest_bwd_f32(const float* dy, int lddy, int N, int H, int W, int C, float* dx, int lddx, void* stream);
/* AdaptiveAvgPool2d((6,6)) + flatten in NCHW order (models_seg_gan.py:277,294-295):
 * y[n, c*36 + oy*6 + ox]. */
int ssg_adaptive_avgpool_flat_fwd_f32(const float* x, int N, int H, int W, int C, int ldx, int OHW, float* y, void* stream);
int ssg_adaptive_avgpool_flat_bwd_f32(const float* dy, int N, int H, int W, int C, int OHW, float* dx, int lddx, void* stream);

/* ------------------------------------------------------------------ elementwise
 * SPADE modulate out = x*(1+gamma)+beta (normalization.py:120) and its gradients
 * dx = dy*(1+gamma), dgamma = dy*x, dbeta = dy.  gb holds gamma in channels [0,C) and beta
 * in [C,2C) of one 2C-channel tensor (the fused gamma|beta conv output). */
/* SPADE with the gamma|beta conv and the modulation in ONE kernel (normalization.py:117-120): `d` describes the 3x3 conv from the
 * 4-channel (padded) activation to 2C channels whose packed weight rows are [gamma 0..C-1 | beta C..2C-1] (bias likewise, no
 * residual / activation), but d->out receives  x * (1 + gamma) + beta  (C channels, pixel stride d->ldo) and `gamma` (incl. its
 * bias; pixel stride ldg) is kept for the backward pass; the [N,2C,H,W] tensor never exists.  ssg_spade_conv_modulate_ok(d):
 * 1 when this shape runs on the fused kernel (4-channel input, 3x3, >= 65536 pixels), else the caller runs ssg_conv2d_f32 +
 * ssg_spade_modulate_fwd_f32. */
int ssg_spade_conv_modulate_ok(const ssg_conv_desc* d);
int ssg_spade_conv_modulate_f32(const ssg_conv_desc* d, const float* x, int ldx, float* gamma, int ldg, void* stream);
int ssg_spade_modulate_fwd_f32(const float* x, int ldx, const float* gb, int ldgb, int64_t P, int C, float* y, int ldy, void* stream);
int ssg_spade_modulate_bwd_f32(const float* x, int ldx, const float* gb, int ldgb, const float* dy, int lddy, int64_t P, int C,
                               float* dx, int lddx, float* dgb, int lddgb, void* stream);
/* the same pass that also returns the bias gradients of the gamma / beta convs (normalization.py:94-96):
 * sums[0:C] = sum_p dy*x, sums[C:2C] = sum_p dy, fp64, deterministic; ws: ssg_bn_workspace_bytes(P, C) bytes.  Only the gamma
 * half of `gb` is read (ldgb >= C: a gamma-only tensor from ssg_spade_conv_modulate_f32 is fine). */
int ssg_spade_modulate_bwd_sums_f32(const float* x, int ldx, const float* gb, int ldgb, const float* dy, int lddy, int64_t P,
                                    int C, float* dx, int lddx, float* dgb, int lddgb, double* sums, void* ws, void* stream);
/* activation backward: dx = dy * (y > 0 ? 1 : slope) (+ add) */
int ssg_act_bwd_f32(const float* y, int ldy, const float* dy, int lddy, int64_t P, int C, int act, float slope, float* dx, int lddx, void* stream);
/* dst[p, 0:C] = src[p, 0:C]: channel-slice copy (materialised torch.cat of > 2 tensors, archs.py:910-925) */
int ssg_copy_channels_f32(const float* src, int ldsrc, int64_t P, int C, float* dst, int lddst, void* stream);
/* out = a + b (gradient accumulation across consumers) */
int ssg_add_f32(const float* a, const float* b, int64_t n, float* out, void* stream);
/* x[isnan(x)] = 0 in place, mask[i] = 1 where it was NaN (train_seg_gan.py:190) */
int ssg_nan_to_zero_f32(float* x, int64_t n, uint8_t* mask, void* stream);
/* out[i] = mask[i] ? 0 : g[i] (gradient of the masking above; out may alias g) */
int ssg_mask_zero_f32(const float* g, const uint8_t* mask, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------ fused segmentation loss
 * One pass over logits x and target t (both NHWC-with-stride, C channels, S = H*W pixels per
 * sample) produces everything train_seg_gan.py:191-198 needs:
 *   res[0] = BCEDiceLoss (losses.py:274-302)        res[1] = MSELoss (train_seg_gan.py:195)
 *   res[2] = StableBCE mean (losses.py:130-136)     res[3] = 1 - mean_n dice_n
 *   res[4] = iou_score on channels [mc0,C) (metrics.py:6-22)   res[5] = dice_coef (metrics.py:25-35)
 *   res[6] = bce-is-finite flag
 *   stats[n*3 + {0,1,2}] = per-sample (sum p*t, sum p, sum t)  (kept for backward)
 *   stats[3N + {0..4}]   = (|pred&tgt|, |pred|tgt|, sum p*t, sum p, sum t) on channels >= mc0
 *                          (metric partial sums, all-reduced across ranks); stats holds 3N+5 doubles
 * backward: dx = g_seg * dBCEDice/dx + g_mse * dMSE/dx + g_bce * dStableBCE/dx, g_* read from device scalars
 * (NULL = 0); res[3..7] carry no gradient. */
int64_t ssg_seg_loss_workspace_bytes(int N, int64_t S, int C);
int ssg_seg_loss_fwd_f32(const float* x, int ldx, const float* t, int ldt, int N, int64_t S, int C, int mc0,
                         float* res, double* stats, void* ws, void* stream);
int ssg_seg_loss_bwd_f32(const float* x, int ldx, const float* t, int ldt, int N, int64_t S, int C,
                         const float* res, const double* stats, const float* g_seg, const float* g_mse,
                         const float* g_bce, float* dx, int lddx, void* stream);
/* BCEWithLogitsLoss(mean) of n logits x[i*ldx] against a constant label
 * (train_seg_gan.py:204,221-222); backward writes dx[i*lddx] and zeros the pad columns. */
int ssg_bce_logits_const_fwd_f32(const float* x, int n, int ldx, float label, float* loss, void* stream);
int ssg_bce_logits_const_bwd_f32(const float* x, int n, int ldx, float label, const float* g, float* dx, int lddx, void* stream);

/* ------------------------------------------------------------------ optimizer
 * clip_gradient (srgan_utils.py:186-195: elementwise clamp to [-clip, clip]) fused with the
 * torch.optim.Adam update (train_seg_gan.py:212-215,230-233) over a list of tensors.
 * ptrs: device array of 4*ntensors pointers {param, grad, exp_avg, exp_avg_sq}; sizes and
 * block_map as produced by the host helper in ssunet-gan_amd/optim.py.  clip <= 0 = off. */
int ssg_clamp_adam_multi_f32(const void* const* ptrs, const int64_t* sizes, const int32_t* blk_tensor,
                             const int32_t* blk_chunk, int nblocks, float clip, double lr, double beta1,
                             double beta2, double eps, double weight_decay, double bias_corr1,
                             double bias_corr2_sqrt, void* stream);
int ssg_clamp_f32(float* x, int64_t n, float lo, float hi, void* stream);
/* clamp member `which` (0 param, 1 grad, 2 exp_avg, 3 exp_avg_sq) of every record of `ptrs` in one
 * launch -- stage 1 clamps the WEIGHTS to +-clip every step (train.py:111-112) */
int ssg_clamp_multi_f32(const void* const* ptrs, const int64_t* sizes, const int32_t* blk_tensor, const int32_t* blk_chunk,
                        int nblocks, int which, float lo, float hi, void* stream);

/* per-channel sums over pixels: out[c] = sum_p x[p,c] (bias gradients) */
int ssg_channel_sum_f32(const float* x, int64_t P, int C, int ld, float* out, void* ws, void* stream);

/* ------------------------------------------------------------------ unwired per-op kernels
 * Depthwise convolution (groups = C): EfficientNet MBConv k in {3,5}, s in {1,2} with TF "same"
 * static padding (efficientnet_pytorch/model.py:46-50, utils.py:123-146) and the 9x9 gate conv of
 * xresidualblock.py:16.  `w` is the reference's [C][1][KH][KW] parameter tensor as is.
 *   fwd  : out[n,oy,ox,c] = bias[c] + sum_k in[n, oy*s+ky-pad_top, ox*s+kx-pad_left, c] * w[c,ky,kx]
 *   dgrad: gradient w.r.t. in;  wgrad: gradient w.r.t. w (fp64 two-stage reduction, deterministic) */
int ssg_dwconv2d_fwd_f32(const float* in, int N, int H, int W, int C, int ld, const float* w, const float* bias, int KH, int KW,
                         int stride, int pad_top, int pad_left, int OH, int OW, float* out, int ldo, void* stream);
int ssg_dwconv2d_dgrad_f32(const float* dout, int lddo, int N, int H, int W, int C, const float* w, int KH, int KW, int stride,
                           int pad_top, int pad_left, int OH, int OW, float* dx, int lddx, void* stream);
int64_t ssg_dwconv2d_wgrad_workspace_bytes(int N, int OH, int OW, int C, int KH, int KW);
int ssg_dwconv2d_wgrad_f32(const float* in, int N, int H, int W, int C, int ld, const float* dout, int lddo, int KH, int KW,
                           int stride, int pad_top, int pad_left, int OH, int OW, float* dw, void* ws, void* stream);
/* element-wise: swish x*sigmoid(x) with the reference's backward (efficientnet_pytorch/utils.py:37-48),
 * sigmoid (SE gate, model.py:79), Gaussian exp(-x^2) (xresidualblock.py:5-7) */
#define SSG_UNARY_SWISH 0
#define SSG_UNARY_SIGMOID 1
#define SSG_UNARY_GAUSSIAN 2
int ssg_unary_fwd_f32(const float* x, int ldx, int64_t P, int C, int op, float* y, int ldy, void* stream);
int ssg_unary_bwd_f32(const float* x, int ldx, const float* dy, int lddy, int64_t P, int C, int op, float* dx, int lddx, void* stream);
/* y = a*b (xresidualblock.py:23) and its gradients */
int ssg_mul_fwd_f32(const float* a, int lda, const float* b, int ldb, int64_t P, int C, float* y, int ldy, void* stream);
int ssg_mul_bwd_f32(const float* a, int lda, const float* b, int ldb, const float* dy, int lddy, int64_t P, int C,
                    float* da, int ldda, float* db, int lddb, void* stream);
/* squeeze-excite plumbing (model.py:76-80): y[n,p,c] = x[n,p,c]*s[n,c]; out[n,c] = scale*sum_p a[n,p,c]*(b?b[n,p,c]:1)
 * (global average pool and the gate's gradient); y[n,p,c] = scale*s[n,c] (average-pool backward) */
int ssg_channel_scale_fwd_f32(const float* x, int ldx, const float* s, int N, int64_t S, int C, float* y, int ldy, void* stream);
int64_t ssg_sample_channel_sum_workspace_bytes(int N, int64_t S, int C);
int ssg_sample_channel_sum_f32(const float* a, int lda, const float* b, int ldb, int N, int64_t S, int C, float scale, float* out,
                               void* ws, void* stream);
int ssg_broadcast_rows_f32(const float* s, int N, int64_t S, int C, float scale, float* y, int ldy, void* stream);
/* Spectral norm (spectral_norm.py:38-88): power iteration on W [rows][cols] with in-place u [rows],
 * v [cols]; W_out = W / sigma (W_out may be NULL: sigma only), sigma = u.(W v).  Backward treats u, v as constants:
 * dW = dWsn/sigma - (sum(dWsn.W)/sigma^2) u v^T. */
int64_t ssg_spectral_norm_workspace_bytes(int rows, int cols);
int ssg_spectral_norm_fwd_f32(const float* W, int rows, int cols, float* u, float* v, int n_power_iterations, double eps,
                              float* W_out, float* sigma, void* ws, void* stream);
int ssg_spectral_norm_bwd_f32(const float* dWsn, const float* W, int rows, int cols, const float* u, const float* v,
                              const float* sigma, float* dW, void* ws, void* stream);

/* ------------------------------------------------------------------ squeeze-excite gate (se_gate.hip)
 * efficientnet_pytorch/model.py:83-87: x_squeezed = adaptive_avg_pool2d(x, 1); _se_expand(swish(_se_reduce(x_squeezed)));
 * x = sigmoid(x_squeezed) * x.  These entry points replace the two 1x1 convs on the pooled [N][C] vector, the swish between
 * them and the sigmoid after them (the pooling and the gating product stay ssg_sample_channel_sum_* / ssg_channel_scale_*):
 *   h_pre[n][s] = b1[s] + sum_c w1[s][c] * sq[n][c]          (w1 = _se_reduce.weight [S][C][1][1], b1 may be NULL)
 *   gate[n][c]  = sigmoid(b2[c] + sum_s w2[c][s] * swish(h_pre[n][s]))   (w2 = _se_expand.weight [C][S][1][1])
 * h_pre ([N][S], dense) is kept for the backward.  Range: N <= 16, S <= 256, N*S <= 2048 (ssg_se_gate_ok; outside it the
 * caller keeps the conv path).  Backward: from dgate it writes dsq [N][C] (stride lds_), dw1 [S][C], db1 [S] (may be NULL),
 * dw2 [C][S], db2 [C] (may be NULL).  tmp = ssg_se_gate_workspace_floats(N, C, S) floats of scratch in either direction
 * (partial sums per 64-channel chunk; dz in the backward; the entry points cannot check its size).  All sums run in a fixed
 * order.  Alignment: when S % 4 == 0 the rows of w2 (and dw2 in the backward) are accessed 16 bytes at a time, so w2 / dw2 must
 * be 16-byte aligned (SSG_EALIGN otherwise) -- a view at an odd offset inside a flat parameter buffer is refused, not mis-read. */
int ssg_se_gate_ok(int N, int C, int S);
int64_t ssg_se_gate_workspace_floats(int N, int C, int S);
int ssg_se_gate_fwd_f32(const float* sq, int ldq, int N, int C, const float* w1, const float* b1, const float* w2, const float* b2,
                        int S, float* h_pre, float* gate, int ldg, float* tmp, void* stream);
int ssg_se_gate_bwd_f32(const float* dgate, int ldd, const float* gate, int ldg, const float* h_pre, const float* sq, int ldq, int N, int C,
                        const float* w1, const float* w2, int S, float* dsq, int lds_, float* dw1, float* db1, float* dw2, float* db2,
                        float* tmp, void* stream);

/* ------------------------------------------------------------------ calibration (bench only)
 * Measured ceilings of the device the job runs on: back-to-back fp32 32x32x2 MFMA issue
 * (FLOPs = blocks*4*iters*16*4096; scratch holds blocks*256 floats) and a 16-B/lane HBM copy. */
int ssg_tool_mfma_peak_f32(float* scratch, int blocks, int iters, void* stream);
/* the same loop with v_mfma_f32_32x32x16_bf16 (32768 FLOP each): the ceiling of the split-operand kernels on this device */
int ssg_tool_mfma_peak_bf16(float* scratch, int blocks, int iters, void* stream);
/* ... on caller operands (`data`: 64 KiB of bf16 values), which toggle the multiplier inputs: the power-limited rate */
int ssg_tool_mfma_peak_bf16_data(float* scratch, int blocks, int iters, const void* data, void* stream);
int ssg_tool_mfma_peak_bf16_data16(float* scratch, int blocks, int iters, const void* data, void* stream);   /* the same on v_mfma_f32_16x16x32_bf16 */
int ssg_tool_copy_f32(const float* src, float* dst, int64_t n, void* stream);   /* one float4 per thread, non-temporal */

#ifdef __cplusplus
}
#endif
#endif
