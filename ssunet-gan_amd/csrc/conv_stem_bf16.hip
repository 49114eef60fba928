// Dense k x k convolution of a <= 4-channel image into a bf16 tensor: the EfficientNet stem (efficientnet_pytorch/model.py:162,206:
// Conv2d(3, 32..48, k=3, s=2, bias=False) with TensorFlow "SAME" padding, utils.py:123-146) in the bf16 tensor family of BASELINE
// config 4 -- SURVEY.md 8(b) `conv2d_{fwd,dgrad,wgrad}_nhwc_bf16`, k = 3.  The image is the model input (fp32 NHWC, 4-channel pixel
// rows: ops.as_nhwc); its values and the weights are ROUNDED TO bf16 before they multiply (what a bf16 MFMA operand is), products
// accumulate in fp32, the output is bf16.  27 MACs per output: the layer is bound by its 100-MB output (B4 at 4 x 1024^2), not by
// arithmetic, so the multiplies run on the vector unit.
//   forward : a thread = one output pixel x 8 output channels; weights [tap][4][Cout] in LDS.
//   wgrad   : dW[co][c][ky][kx] = sum over output pixels of dy[p][co] * x[p*s + tap][c]; a thread owns (tap, co) and 4 accumulators
//             (one per input channel) over a block's pixel range, partials [block][tap][co][4] reduced in fp64 by a second kernel.
//   dgrad   : gather over the taps that reach an input pixel (the image gradient: not needed by the encoder, present for the row).
#include "common.h"

namespace {

__device__ __forceinline__ float bf16_round(float v) { return (float)(ssg_bf16)v; }
__device__ __forceinline__ f32x4 bf16_round4(f32x4 v) { return f32x4{bf16_round(v[0]), bf16_round(v[1]), bf16_round(v[2]), bf16_round(v[3])}; }

constexpr int MAXT = 25;                                 // up to 5 x 5
constexpr int FWD_ITEMS = 16;                            // 256-thread slices of (pixel, channel octet) items per forward block

__global__ __launch_bounds__(256) void stem_bf16_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int ldx,
                                                            const float* __restrict__ w, int Cout, int Cin, int KH, int KW, int stride,
                                                            int pt, int pl, int OH, int OW, ssg_bf16* __restrict__ y, int ldy) {
  extern __shared__ float wl[];                          // [tap][4][Cout], bf16-rounded
  const int ntaps = KH * KW;
  for (int i = threadIdx.x; i < ntaps * 4 * Cout; i += 256) {
    const int co = i % Cout, c = (i / Cout) & 3, t = i / (4 * Cout);
    wl[i] = c < Cin ? bf16_round(w[((size_t)co * Cin + c) * ntaps + t]) : 0.f;
  }
  __syncthreads();
  const int CG = Cout >> 3;
  const unsigned total = (unsigned)N * OH * OW * CG;     // < 2^32 (launcher)
  // FWD_ITEMS consecutive 256-item slices per block: the weight table above is staged once for all of them
  for (int it = 0; it < FWD_ITEMS; ++it) {
  const unsigned id = ((unsigned)blockIdx.x * FWD_ITEMS + it) * 256u + threadIdx.x;
  if (id >= total) break;
  const unsigned pix = id / (unsigned)CG;
  const int cg = (int)(id - pix * CG);
  const int ox = (int)(pix % (unsigned)OW); const unsigned r = pix / (unsigned)OW;
  const int oy = (int)(r % (unsigned)OH), n = (int)(r / (unsigned)OH);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int ky = 0; ky < KH; ++ky) {
    const int iy = oy * stride - pt + ky;
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int kx = 0; kx < KW; ++kx) {
      const int ix = ox * stride - pl + kx;
      if ((unsigned)ix >= (unsigned)W) continue;
      const f32x4 xv = bf16_round4(*(const f32x4*)(x + ((size_t)(n * H + iy) * W + ix) * ldx));
      const float* wt = wl + (size_t)(ky * KW + kx) * 4 * Cout + cg * 8;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 w0 = *(const f32x4*)(wt + c * Cout), w1 = *(const f32x4*)(wt + c * Cout + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] = __builtin_fmaf(xv[c], w0[j], acc[j]); acc[4 + j] = __builtin_fmaf(xv[c], w1[j], acc[4 + j]); }
      }
    }
  }
  ssg_bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (ssg_bf16)acc[j];
  *(ssg_bf16x8*)(y + (size_t)pix * ldy + cg * 8) = o;
  }
}

// A thread owns (tap, 8 output channels) x 4 input channels = 32 accumulators and every PL-th pixel of the block's range: one 16-byte
// load of dy (8 bf16) and one of x (4 fp32) per 32 multiply-adds.  (One (tap, channel) pair per thread -- 2 loads per 4 multiply-adds
// -- ran at the rate of the texture addresser: 0.9 ms at B4 / 4 x 1024^2.)  partial[(block * PL + lane)][tap * Cout + co] = f32x4 over
// the input channels.
constexpr int WG_PL_MAX = 4;                             // pixel lanes per block: as many as 256 threads hold, at most 4
__global__ __launch_bounds__(256) void stem_bf16_wgrad_kernel(const float* __restrict__ x, int N, int H, int W, int ldx,
                                                              const ssg_bf16* __restrict__ dy, int lddy, int Cout, int KH, int KW,
                                                              int stride, int pt, int pl, int OH, int OW, int per_block,
                                                              int WG_PL, f32x4* __restrict__ partial) {
  extern __shared__ f32x4 red[];                         // [WG_PL - 1][ntaps * Cout]: the lanes' sums meet here
  const int CG = Cout >> 3, ntaps = KH * KW, groups = ntaps * CG, combos = ntaps * Cout;
  const int P = N * OH * OW;                              // < 2^31 (launcher)
  const int p0 = blockIdx.x * per_block;
  const int p1 = min(p0 + per_block, P);
  const int nthr = groups * WG_PL;                        // <= 256 (launcher)
  const int t0 = threadIdx.x;
  const int q = t0 / groups, gidx = t0 - q * groups;      // pixel lane, (tap, channel octet)
  const int tap = gidx / CG, cg = gidx - tap * CG;
  const int ky = tap / KW, kx = tap - ky * KW;
  f32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (t0 < nthr && p0 + q < p1) {
    int p = p0 + q;
    int ox = p % OW, r = p / OW;
    int oy = r % OH, n = r / OH;
    for (; p < p1; p += WG_PL) {
      const int iy = oy * stride - pt + ky, ix = ox * stride - pl + kx;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
        const f32x4 xv = bf16_round4(*(const f32x4*)(x + ((size_t)(n * H + iy) * W + ix) * ldx));
        const ssg_bf16x8 g = *(const ssg_bf16x8*)(dy + (size_t)p * lddy + cg * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)g[j] * xv;
      }
      ox += WG_PL;
      while (ox >= OW) { ox -= OW; if (++oy == OH) { oy = 0; ++n; } }
    }
  }
  if (t0 < nthr && q > 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(size_t)(q - 1) * combos + tap * Cout + cg * 8 + j] = acc[j];
  }
  __syncthreads();
  if (t0 < nthr && q == 0) {
    f32x4* dst = partial + (size_t)blockIdx.x * combos + (size_t)tap * Cout + cg * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f32x4 v = acc[j];
      for (int l = 0; l < WG_PL - 1; ++l) v += red[(size_t)l * combos + tap * Cout + cg * 8 + j];
      dst[j] = v;
    }
  }
}

// one wave per (tap, output channel): lanes stride over the blocks' partials, fp64 butterfly
__global__ __launch_bounds__(64) void stem_bf16_wgrad_reduce_kernel(const f32x4* __restrict__ partial, int blocks, int Cout, int Cin,
                                                                    int ntaps, float* __restrict__ dw /* [Cout][Cin][ntaps] */) {
  const int combos = ntaps * Cout;
  const int t0 = blockIdx.x;
  double s[4] = {0, 0, 0, 0};
  for (int b = threadIdx.x; b < blocks; b += 64) {
    const f32x4 v = partial[(size_t)b * combos + t0];
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] += (double)v[c];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s[c] += __shfl_xor(s[c], m, 64);
  if (threadIdx.x == 0) {
    const int tap = t0 / Cout, co = t0 - tap * Cout;
    for (int c = 0; c < Cin; ++c) dw[((size_t)co * Cin + c) * ntaps + tap] = (float)s[c];
  }
}

// dx[n][iy][ix][c] = sum over taps and output channels of dy[n][oy][ox][co] * w[co][c][ky][kx], iy = oy*s - pt + ky
__global__ __launch_bounds__(256) void stem_bf16_dgrad_kernel(const ssg_bf16* __restrict__ dy, int lddy, int N, int H, int W,
                                                              const float* __restrict__ w, int Cout, int Cin, int KH, int KW, int stride,
                                                              int pt, int pl, int OH, int OW, float* __restrict__ dx, int lddx) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (long long)N * H * W) return;
  const int ix = (int)(pix % W); const long long r = pix / W;
  const int iy = (int)(r % H), n = (int)(r / H);
  const int ntaps = KH * KW;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int ky = 0; ky < KH; ++ky) {
    const int ty = iy + pt - ky;
    if (ty < 0 || ty % stride) continue;
    const int oy = ty / stride;
    if (oy >= OH) continue;
    for (int kx = 0; kx < KW; ++kx) {
      const int tx = ix + pl - kx;
      if (tx < 0 || tx % stride) continue;
      const int ox = tx / stride;
      if (ox >= OW) continue;
      const ssg_bf16* g = dy + ((size_t)(n * OH + oy) * OW + ox) * lddy;
      for (int co = 0; co < Cout; ++co) {
        const float gv = (float)g[co];
        for (int c = 0; c < Cin; ++c) acc[c] = __builtin_fmaf(gv, bf16_round(w[((size_t)co * Cin + c) * ntaps + ky * KW + kx]), acc[c]);
      }
    }
  }
  *(f32x4*)(dx + (size_t)pix * lddx) = acc;
}

int stem_blocks(long long P) {
  long long b = (P + 1023) / 1024;                        // ~1024 output pixels per block (256 per pixel lane)
  if (b > 2048) b = 2048;
  return b < 1 ? 1 : (int)b;
}

int check(const void* x, int N, int H, int W, int ldx, const void* w, int Cout, int Cin, int KH, int KW, int stride, int OH, int OW, const char* what) {
  SSG_REQUIRE(x && w && N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, SSG_EINVAL, "%s: bad args", what);
  SSG_REQUIRE(Cin >= 1 && Cin <= 4 && ldx >= 4 && ldx % 4 == 0 && Cout >= 8 && Cout % 8 == 0, SSG_EINVAL,
              "%s: Cin <= 4 on 4-channel pixel rows and Cout %% 8 == 0 (Cin %d, ldx %d, Cout %d)", what, Cin, ldx, Cout);
  SSG_REQUIRE(KH >= 1 && KW >= 1 && KH * KW <= MAXT && stride >= 1 && stride <= 2, SSG_EINVAL, "%s: window %d x %d stride %d", what, KH, KW, stride);
  return SSG_OK;
}

}  // namespace

extern "C" int ssg_conv2d_thin_bf16(const float* x, int N, int H, int W, int ldx, const float* w_oihw, int Cout, int Cin, int KH, int KW,
                                    int stride, int pad_t, int pad_l, int OH, int OW, void* y, int ldy, void* stream) {
  int rc = check(x, N, H, W, ldx, w_oihw, Cout, Cin, KH, KW, stride, OH, OW, "conv2d_thin_bf16");
  if (rc != SSG_OK) return rc;
  SSG_REQUIRE(y && ldy >= Cout && ldy % 8 == 0 && ssg_aligned16(x) && ssg_aligned16(y), SSG_EALIGN, "conv2d_thin_bf16: output rows / alignment");
  const long long threads = (long long)N * OH * OW * (Cout / 8);
  SSG_REQUIRE(threads < (1ll << 32) - 65536, SSG_EINVAL, "conv2d_thin_bf16: too large");
  const size_t lds = (size_t)KH * KW * 4 * Cout * sizeof(float);
  SSG_REQUIRE(lds <= 64 * 1024, SSG_EINVAL, "conv2d_thin_bf16: weights do not fit LDS (Cout %d)", Cout);
  hipLaunchKernelGGL(stem_bf16_fwd_kernel, dim3((unsigned)ssg_cdiv(threads, 256 * FWD_ITEMS)), dim3(256), lds, (hipStream_t)stream,
                     x, N, H, W, ldx, w_oihw, Cout, Cin, KH, KW, stride, pad_t, pad_l, OH, OW, (ssg_bf16*)y, ldy);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int64_t ssg_conv2d_thin_bf16_wgrad_workspace_bytes(int N, int OH, int OW, int Cout, int KH, int KW) {
  if (N <= 0 || OH <= 0 || OW <= 0 || Cout <= 0) return 0;
  return (int64_t)stem_blocks((long long)N * OH * OW) * KH * KW * Cout * 16;
}

extern "C" int ssg_conv2d_thin_bf16_wgrad(const float* x, int N, int H, int W, int ldx, const void* dy, int lddy, int Cout, int Cin,
                                          int KH, int KW, int stride, int pad_t, int pad_l, int OH, int OW, float* dw_oihw, void* ws,
                                          void* stream) {
  int rc = check(x, N, H, W, ldx, dy, Cout, Cin, KH, KW, stride, OH, OW, "conv2d_thin_bf16_wgrad");
  if (rc != SSG_OK) return rc;
  SSG_REQUIRE(dw_oihw && ws && lddy >= Cout && ssg_aligned16(x) && ssg_aligned16(ws), SSG_EALIGN, "conv2d_thin_bf16_wgrad: pointers");
  const long long P = (long long)N * OH * OW;
  const int groups = KH * KW * (Cout / 8);
  SSG_REQUIRE(P < (1ll << 31) && groups <= 256, SSG_EINVAL, "conv2d_thin_bf16_wgrad: %lld pixels / %d (tap, channel-octet) groups", P, groups);
  const int WG_PL = 256 / groups > WG_PL_MAX ? WG_PL_MAX : 256 / groups;
  const int blocks = stem_blocks(P);
  const int per_block = (int)((P + blocks - 1) / blocks);
  SSG_REQUIRE(lddy % 8 == 0 && ssg_aligned16(dy), SSG_EALIGN, "conv2d_thin_bf16_wgrad: dy rows");
  const size_t lds = (size_t)(WG_PL - 1) * KH * KW * Cout * 16;
  SSG_REQUIRE(lds <= 64 * 1024, SSG_EINVAL, "conv2d_thin_bf16_wgrad: Cout %d too wide for the in-block reduce", Cout);
  hipLaunchKernelGGL(stem_bf16_wgrad_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, N, H, W, ldx,
                     (const ssg_bf16*)dy, lddy, Cout, KH, KW, stride, pad_t, pad_l, OH, OW, per_block, WG_PL, (f32x4*)ws);
  SSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(stem_bf16_wgrad_reduce_kernel, dim3((unsigned)(KH * KW * Cout)), dim3(64), 0, (hipStream_t)stream,
                     (const f32x4*)ws, blocks, Cout, Cin, KH * KW, dw_oihw);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_conv2d_thin_bf16_dgrad(const void* dy, int lddy, int N, int H, int W, const float* w_oihw, int Cout, int Cin, int KH, int KW,
                                          int stride, int pad_t, int pad_l, int OH, int OW, float* dx, int lddx, void* stream) {
  int rc = check(dy, N, H, W, lddx, w_oihw, Cout, Cin, KH, KW, stride, OH, OW, "conv2d_thin_bf16_dgrad");
  if (rc != SSG_OK) return rc;
  SSG_REQUIRE(dx && lddy >= Cout && ssg_aligned16(dx), SSG_EALIGN, "conv2d_thin_bf16_dgrad: pointers");
  hipLaunchKernelGGL(stem_bf16_dgrad_kernel, dim3((unsigned)ssg_cdiv((long long)N * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const ssg_bf16*)dy, lddy, N, H, W, w_oihw, Cout, Cin, KH, KW, stride, pad_t, pad_l, OH, OW, dx, lddx);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
