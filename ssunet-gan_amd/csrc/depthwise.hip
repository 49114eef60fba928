// Depthwise convolution + the element-wise pieces of the (unwired) MBConv / xResidualBlock ops for
// gfx950.  All HBM-bound: one thread per 16-byte channel quad of one pixel, taps walked in
// registers, weights read from the reference's [C][1][KH][KW] parameter (L1/L2 resident).
// ABI + reference citations: include/ssunet_hip.h (ssg_dwconv2d_*, ssg_unary_*, ssg_mul_*,
// ssg_channel_scale_*, ssg_global_avgpool_*).
#include "common.h"

namespace {

int elem_grid(long long total) { return ssg_elem_grid(total, 2); }
#define GRID_STRIDE(i, total) SSG_CHUNK_LOOP(i, total)      // contiguous chunk per block (common.h)

template <typename T>
struct DwArgs {
  const T* in; const float* w; const float* bias; const T* dout; T* out;
  int N, H, W, C, ld, KH, KW, stride, pt, pl, OH, OW, ldo;
};

template <typename T>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const DwArgs<T> a) {
  const int CQ = a.C / 4, KK = a.KH * a.KW;
  const long long total = (long long)a.N * a.OH * a.OW * CQ;
  GRID_STRIDE(i, total) {
    const int cq = (int)(i % CQ); long long r = i / CQ;
    const int ox = (int)(r % a.OW); r /= a.OW;
    const int oy = (int)(r % a.OH); const int n = (int)(r / a.OH);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) acc = *(const f32x4*)(a.bias + 4 * cq);
    const float* wp = a.w + (size_t)4 * cq * KK;
    for (int ky = 0; ky < a.KH; ++ky) {
      const int iy = oy * a.stride + ky - a.pt;
      if ((unsigned)iy >= (unsigned)a.H) continue;
      for (int kx = 0; kx < a.KW; ++kx) {
        const int ix = ox * a.stride + kx - a.pl;
        if ((unsigned)ix >= (unsigned)a.W) continue;
        const f32x4 v = ld4(a.in + ((size_t)(n * a.H + iy) * a.W + ix) * a.ld + 4 * cq);
        const int t = ky * a.KW + kx;
        acc[0] += v[0] * wp[t]; acc[1] += v[1] * wp[KK + t]; acc[2] += v[2] * wp[2 * KK + t]; acc[3] += v[3] * wp[3 * KK + t];
      }
    }
    st4(a.out + ((size_t)(n * a.OH + oy) * a.OW + ox) * a.ldo + 4 * cq, acc);
  }
}

// dx[n,y,x,c] = sum_{ky,kx} dout[n,(y+pt-ky)/s,(x+pl-kx)/s,c] * w[c,ky,kx]  (where divisible, in range)
template <typename T>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const DwArgs<T> a) {
  const int CQ = a.C / 4, KK = a.KH * a.KW;
  const long long total = (long long)a.N * a.H * a.W * CQ;
  GRID_STRIDE(i, total) {
    const int cq = (int)(i % CQ); long long r = i / CQ;
    const int x = (int)(r % a.W); r /= a.W;
    const int y = (int)(r % a.H); const int n = (int)(r / a.H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* wp = a.w + (size_t)4 * cq * KK;
    for (int ky = 0; ky < a.KH; ++ky) {
      const int ty = y + a.pt - ky;
      if (ty < 0 || ty % a.stride) continue;
      const int oy = ty / a.stride;
      if (oy >= a.OH) continue;
      for (int kx = 0; kx < a.KW; ++kx) {
        const int tx = x + a.pl - kx;
        if (tx < 0 || tx % a.stride) continue;
        const int ox = tx / a.stride;
        if (ox >= a.OW) continue;
        const f32x4 g = ld4(a.dout + ((size_t)(n * a.OH + oy) * a.OW + ox) * a.ldo + 4 * cq);
        const int t = ky * a.KW + kx;
        acc[0] += g[0] * wp[t]; acc[1] += g[1] * wp[KK + t]; acc[2] += g[2] * wp[2 * KK + t]; acc[3] += g[3] * wp[3 * KK + t];
      }
    }
    st4(a.out + ((size_t)(n * a.H + y) * a.W + x) * a.ld + 4 * cq, acc);
  }
}

// dw[c, t] = sum_pixels dout[p, c] * in[p*s + t, c]: column reduction per tap.  grid = (parts, channel
// groups, taps); fp64 block combine; ordered second stage (deterministic).
constexpr int DW_TQ = 16;                 // channel quads per block row
template <typename T>
__global__ __launch_bounds__(256) void dw_wgrad_partial_kernel(const DwArgs<T> a, long long rows_per_part, double* __restrict__ part) {
  __shared__ double red[256][4];
  const int tid = threadIdx.x, tq = tid % DW_TQ, pr = tid / DW_TQ, PR = 256 / DW_TQ;
  const int CQ = a.C / 4, cq = blockIdx.y * DW_TQ + tq;
  const int t = blockIdx.z, ky = t / a.KW, kx = t - ky * a.KW;
  const long long P = (long long)a.N * a.OH * a.OW;
  const long long p0 = (long long)blockIdx.x * rows_per_part;
  long long p1 = p0 + rows_per_part; if (p1 > P) p1 = P;
  typedef typename SsgAcc<T>::type acc_t;
  acc_t s[4] = {0, 0, 0, 0};
  if (cq < CQ) {
    for (long long p = p0 + pr; p < p1; p += PR) {
      const int ox = (int)(p % a.OW); const long long r = p / a.OW;
      const int oy = (int)(r % a.OH); const int n = (int)(r / a.OH);
      const int iy = oy * a.stride + ky - a.pt, ix = ox * a.stride + kx - a.pl;
      if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
        const f32x4 g = ld4(a.dout + (size_t)p * a.ldo + 4 * cq);
        const f32x4 v = ld4(a.in + ((size_t)(n * a.H + iy) * a.W + ix) * a.ld + 4 * cq);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += (acc_t)g[e] * (acc_t)v[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[tid][e] = (double)s[e];
  __syncthreads();
  if (pr == 0 && cq < CQ) {
    double tot[4] = {0, 0, 0, 0};
    for (int r = 0; r < PR; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) tot[e] += red[r * DW_TQ + tq][e];
    double* dst = part + ((size_t)blockIdx.x * gridDim.z + t) * a.C + 4 * cq;
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[e] = tot[e];
  }
}
// Second stages: 8 outputs x 32 part-lanes per 256-thread block.  Lane k of an output adds partial rows k, k+32, ... in row
// order, then the 32 lane sums are folded by a fixed xor tree: a fixed summation order (bitwise reproducible) with 1/32 of
// the serial chain of one thread per output (the one-thread version took 19-30 us per launch, all of it dependent loads).
__device__ __forceinline__ double fold32(double v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}
__global__ __launch_bounds__(256) void dw_wgrad_final_kernel(const double* __restrict__ part, int parts, int KK, int C, float* __restrict__ dw) {
  const int lane = threadIdx.x & 31;
  const int idx = blockIdx.x * 8 + (threadIdx.x >> 5);           // over KK*C, idx = t*C + c
  double s = 0;
  if (idx < KK * C)
    for (int b = lane; b < parts; b += 32) s += part[(size_t)b * KK * C + idx];
  s = fold32(s);
  if (lane == 0 && idx < KK * C) {
    const int t = idx / C, c = idx - t * C;
    dw[(size_t)c * KK + t] = (float)s;
  }
}

// ---------------------------------------------------------------- stride-1 depthwise, register tiled
// The one-thread-per-output kernels above issue KH*KW 16-B loads (+ 4 weight dwords) per output quad: at 9x9
// (xResidualBlock) they are L1-issue bound at ~1.5 TFLOP/s.  For unit stride:
//   * forward and input gradient are the same kernel (the gradient is the correlation with the flipped kernel and
//     pad' = K-1-pad): a thread owns XT = 4 consecutive outputs of one channel quad, loads each input row segment
//     (XT+KW-1 quads) once per kernel row and reuses it for all KW taps; the block's weights sit in LDS as
//     [tap][16 quads][4] so one ds_read_b128 serves 4 channels x XT outputs;
//   * the weight gradient runs one kernel ROW of taps per workgroup (grid.z = KH, not KH*KW): dout is re-read KH
//     times instead of KH*KW, KW x 4 fp64 accumulators per thread, ordered two-stage reduction as before.
template <typename T>
struct DwS1Args {
  const T* src; const float* w; const float* bias; T* dst;
  int N, SH, SW, DH, DW_, C, lds_, ldd, KH, pt, pl, flip;      // source image SH x SW, destination DH x DW_
};
constexpr int DW_XT = 4;

// S = 2 (round 3): the same tiling for the stride-2 forward (the four stage transitions of an EfficientNet, 259 us each at B4 /
// 1024^2 on the one-thread-per-output kernel): a thread's 4 outputs read (XT - 1) * 2 + KW source columns per kernel row.
template <int KW, typename T, int S = 1>
__global__ __launch_bounds__(256) void dw_s1_kernel(const DwS1Args<T> a) {
  extern __shared__ float w_s[];                         // [KH*KW][16][4]
  const int KK = a.KH * KW;
  const int tid = threadIdx.x, tq = tid & 15, xg = tid >> 4;
  const int cq0 = blockIdx.z * 16;
  for (int idx = tid; idx < KK * 64; idx += 256) {
    const int t = idx >> 6, c = idx & 63;
    const int ch = cq0 * 4 + c;
    w_s[idx] = ch < a.C ? a.w[(size_t)ch * KK + (a.flip ? KK - 1 - t : t)] : 0.f;
  }
  __syncthreads();
  const int cq = cq0 + tq;
  if (cq * 4 >= a.C) return;
  const int oy = blockIdx.y % a.DH, n = blockIdx.y / a.DH;
  const int x0 = blockIdx.x * (16 * DW_XT) + xg * DW_XT;
  if (x0 >= a.DW_) return;
  f32x4 acc[DW_XT];
  const f32x4 b0 = a.bias ? *(const f32x4*)(a.bias + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < DW_XT; ++j) acc[j] = b0;
  for (int ky = 0; ky < a.KH; ++ky) {
    const int iy = oy * S + ky - a.pt;
    if ((unsigned)iy >= (unsigned)a.SH) continue;
    const T* row = a.src + ((size_t)(n * a.SH + iy) * a.SW) * a.lds_ + 4 * cq;
    constexpr int NV = (DW_XT - 1) * S + KW;
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int ix = x0 * S + j - a.pl;
      v[j] = (unsigned)ix < (unsigned)a.SW ? ld4(row + (size_t)ix * a.lds_) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* wrow = w_s + (ky * KW) * 64 + tq * 4;
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {
      const f32x4 wv = *(const f32x4*)(wrow + kx * 64);
#pragma unroll
      for (int j = 0; j < DW_XT; ++j) acc[j] += v[j * S + kx] * wv;
    }
  }
  T* orow = a.dst + ((size_t)(n * a.DH + oy) * a.DW_) * a.ldd + 4 * cq;
#pragma unroll
  for (int j = 0; j < DW_XT; ++j)
    if (x0 + j < a.DW_) st4(orow + (size_t)(x0 + j) * a.ldd, acc[j]);
}

template <int KW, typename T, int S = 1>
int launch_dw_s1(const DwS1Args<T>& a, hipStream_t st) {
  const dim3 grid((unsigned)((a.DW_ + 16 * DW_XT - 1) / (16 * DW_XT)), (unsigned)(a.N * a.DH), (unsigned)((a.C / 4 + 15) / 16));
  hipLaunchKernelGGL((dw_s1_kernel<KW, T, S>), grid, dim3(256), (size_t)a.KH * KW * 64 * sizeof(float), st, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
// stride-2 forward (k3 / k5: the shapes EfficientNet has); -1 = not covered
template <typename T>
int dw_s2_fwd_dispatch(const DwS1Args<T>& a, int KW, hipStream_t st) {
  if ((long long)a.N * a.DH > 65535 || (a.C / 4 + 15) / 16 > 65535) return -1;
  if (KW == 3) return launch_dw_s1<3, T, 2>(a, st);
  if (KW == 5) return launch_dw_s1<5, T, 2>(a, st);
  return -1;
}

// Stride-2 input gradient, same tiling: dx[y][x] = sum over (ky, kx) with (y + pt - ky) and (x + pl - kx) even of
// dout[(y + pt - ky) / 2][(x + pl - kx) / 2] * w[ky][kx].  A thread owns 4 consecutive x of one channel quad; per kernel row of
// matching parity it loads the <= 2 + (KW + 1) / 2 dout columns its outputs touch once and applies the taps whose parity fits
// (x0 is a multiple of 4, so with the parity of pl as a template parameter every parity test and register index below is a
// compile-time constant after unrolling).
constexpr int floor_half(int t) { return (t - (t & 1)) / 2; }
template <int KW, typename T, int PLODD>
__global__ __launch_bounds__(256) void dw_dgrad_s2_kernel(const DwS1Args<T> a) {        // src = dout (SH x SW), dst = dx (DH x DW_)
  extern __shared__ float w_s[];                         // [KH*KW][16][4]
  const int KK = a.KH * KW;
  const int tid = threadIdx.x, tq = tid & 15, xg = tid >> 4;
  const int cq0 = blockIdx.z * 16;
  for (int idx = tid; idx < KK * 64; idx += 256) {
    const int t = idx >> 6, c = idx & 63;
    const int ch = cq0 * 4 + c;
    w_s[idx] = ch < a.C ? a.w[(size_t)ch * KK + t] : 0.f;
  }
  __syncthreads();
  const int cq = cq0 + tq;
  if (cq * 4 >= a.C) return;
  const int y = blockIdx.y % a.DH, n = blockIdx.y / a.DH;
  const int x0 = blockIdx.x * (16 * DW_XT) + xg * DW_XT;                      // multiple of 4
  if (x0 >= a.DW_) return;
  f32x4 acc[DW_XT];
#pragma unroll
  for (int j = 0; j < DW_XT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // dout columns touched: (x0 + j + pl - kx) / 2 for j in [0, 4), kx in [0, KW): from floor((x0 + pl - KW + 1) / 2) up
  constexpr int NV = (DW_XT + KW) / 2 + 1;
  const int base = (x0 + a.pl - (KW - 1) - ((x0 + a.pl - (KW - 1)) & 1)) / 2;  // floor of a possibly negative half
  for (int ky = 0; ky < a.KH; ++ky) {
    const int ty = y + a.pt - ky;
    if (ty < 0 || (ty & 1)) continue;
    const int oy = ty >> 1;
    if (oy >= a.SH) continue;
    const T* row = a.src + ((size_t)(n * a.SH + oy) * a.SW) * a.lds_ + 4 * cq;
    f32x4 v[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int ox = base + q;
      v[q] = (unsigned)ox < (unsigned)a.SW ? ld4(row + (size_t)ox * a.lds_) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* wrow = w_s + (ky * KW) * 64 + tq * 4;
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {
      const f32x4 wv = *(const f32x4*)(wrow + kx * 64);
#pragma unroll
      for (int j = 0; j < DW_XT; ++j) {
        // tx = x0 + j + pl - kx; with pl = 2m + PLODD and x0 % 4 == 0 its parity and its column relative to `base` are constants
        const int e = j + PLODD - kx;
        if ((e & 1) == 0) acc[j] += v[floor_half(e) - floor_half(PLODD - (KW - 1))] * wv;
      }
    }
  }
  T* orow = a.dst + ((size_t)(n * a.DH + y) * a.DW_) * a.ldd + 4 * cq;
#pragma unroll
  for (int j = 0; j < DW_XT; ++j)
    if (x0 + j < a.DW_) st4(orow + (size_t)(x0 + j) * a.ldd, acc[j]);
}
template <typename T>
int dw_s2_dgrad_dispatch(const DwS1Args<T>& a, int KW, hipStream_t st) {
  if ((long long)a.N * a.DH > 65535 || (a.C / 4 + 15) / 16 > 65535) return -1;
  const dim3 grid((unsigned)((a.DW_ + 16 * DW_XT - 1) / (16 * DW_XT)), (unsigned)(a.N * a.DH), (unsigned)((a.C / 4 + 15) / 16));
  const size_t lds = (size_t)a.KH * KW * 64 * sizeof(float);
  const bool odd = (a.pl & 1) != 0;
  if (KW == 3) { if (odd) hipLaunchKernelGGL((dw_dgrad_s2_kernel<3, T, 1>), grid, dim3(256), lds, st, a); else hipLaunchKernelGGL((dw_dgrad_s2_kernel<3, T, 0>), grid, dim3(256), lds, st, a); }
  else if (KW == 5) { if (odd) hipLaunchKernelGGL((dw_dgrad_s2_kernel<5, T, 1>), grid, dim3(256), lds, st, a); else hipLaunchKernelGGL((dw_dgrad_s2_kernel<5, T, 0>), grid, dim3(256), lds, st, a); }
  else return -1;
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
// returns -1 if the shape is not covered (caller falls back to the generic kernel)
template <typename T>
int dw_s1_dispatch(const DwS1Args<T>& a, int KW, hipStream_t st) {
  if ((long long)a.N * a.DH > 65535 || (a.C / 4 + 15) / 16 > 65535) return -1;     // grid.y / grid.z limits
  switch (KW) {
    case 3: return launch_dw_s1<3, T>(a, st);
    case 5: return launch_dw_s1<5, T>(a, st);
    case 7: return launch_dw_s1<7, T>(a, st);
    case 9: return launch_dw_s1<9, T>(a, st);
    default: return -1;
  }
}

// weight gradient, one kernel row ky per blockIdx.z: s[kx][e] += dout[p][c] * in[p + (ky, kx)][c]
// Round 3: a thread takes FOUR consecutive output pixels of a row per iteration (4 dout quads + 3 S + KW input quads for
// 16 KW multiply-adds per channel, where one pixel at a time cost 1 + KW loads for 4 KW): the kernel was bound by load issue
// (PMC: 1.6 TB/s on `dw_wgrad_s1<5, bf16>`).  `rows_per_part` counts such x-quads.
template <int KW, typename T, int S>
__global__ __launch_bounds__(256) void dw_wgrad_s1_kernel(const DwArgs<T> a, long long rows_per_part, double* __restrict__ part) {
  __shared__ double red[256][4];
  const int tid = threadIdx.x, tq = tid % DW_TQ, pr = tid / DW_TQ, PR = 256 / DW_TQ;
  const int CQ = a.C / 4, cq = blockIdx.y * DW_TQ + tq;
  const int ky = blockIdx.z;
  const int OW4 = (a.OW + 3) / 4;
  const long long U = (long long)a.N * a.OH * OW4;
  const long long u0 = (long long)blockIdx.x * rows_per_part;
  long long u1 = u0 + rows_per_part; if (u1 > U) u1 = U;
  typedef typename SsgAcc<T>::type acc_t;
  acc_t s[KW][4];
#pragma unroll
  for (int kx = 0; kx < KW; ++kx)
#pragma unroll
    for (int e = 0; e < 4; ++e) s[kx][e] = 0;
  if (cq < CQ) {
    for (long long u = u0 + pr; u < u1; u += PR) {
      const int xq = (int)(u % OW4); const long long r = u / OW4;
      const int oy = (int)(r % a.OH); const int n = (int)(r / a.OH);
      const int iy = oy * S + ky - a.pt;
      if ((unsigned)iy >= (unsigned)a.H) continue;
      const int ox0 = xq * 4;
      const T* grow = a.dout + ((size_t)(n * a.OH + oy) * a.OW) * a.ldo + 4 * cq;
      f32x4 g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = ox0 + j < a.OW ? ld4(grow + (size_t)(ox0 + j) * a.ldo) : f32x4{0.f, 0.f, 0.f, 0.f};
      const T* row = a.in + ((size_t)(n * a.H + iy) * a.W) * a.ld + 4 * cq;
      constexpr int NV = 3 * S + KW;
      f32x4 v[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int ix = ox0 * S + q - a.pl;
        v[q] = (unsigned)ix < (unsigned)a.W ? ld4(row + (size_t)ix * a.ld) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int kx = 0; kx < KW; ++kx)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) s[kx][e] += (acc_t)g[j][e] * (acc_t)v[j * S + kx][e];
    }
  }
  for (int kx = 0; kx < KW; ++kx) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) red[tid][e] = (double)s[kx][e];
    __syncthreads();
    if (pr == 0 && cq < CQ) {
      double tot[4] = {0, 0, 0, 0};
      for (int r = 0; r < PR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[e] += red[r * DW_TQ + tq][e];
      double* dst = part + ((size_t)blockIdx.x * (a.KH * KW) + ky * KW + kx) * a.C + 4 * cq;
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = tot[e];
    }
  }
}
// (All K x K taps in one workgroup -- K*K*4 accumulators per thread, dout and the input read once instead of K times -- was built
//  and measured in round 3: B4 bf16 137.6 -> 124.0 images/s, fp32 94.2 -> 90.8: 222-256 registers and a K*K-round reduction
//  epilogue cost more than the K-fold re-read, which the L2 serves.)
template <int KW, typename T>
void launch_dw_wgrad_s1(const DwArgs<T>& a, long long parts, long long rpp, double* ws, hipStream_t st) {
  const dim3 grid((unsigned)parts, (unsigned)((a.C / 4 + DW_TQ - 1) / DW_TQ), (unsigned)a.KH);
  if (a.stride == 2) hipLaunchKernelGGL((dw_wgrad_s1_kernel<KW, T, 2>), grid, dim3(256), 0, st, a, rpp, ws);
  else hipLaunchKernelGGL((dw_wgrad_s1_kernel<KW, T, 1>), grid, dim3(256), 0, st, a, rpp, ws);
}

// ---------------------------------------------------------------- unary ops (fwd / bwd)
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void unary_fwd_kernel(const float* __restrict__ x, int ldx, long long P, int C, int op, float* __restrict__ y, int ldy) {
  const int CQ = C / 4;
  GRID_STRIDE(i, P * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    const f32x4 v = *(const f32x4*)(x + p * ldx + 4 * cq);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = v[e];
      o[e] = op == SSG_UNARY_SWISH ? z * sigm(z) : (op == SSG_UNARY_SIGMOID ? sigm(z) : expf(-(z * z)));
    }
    *(f32x4*)(y + p * ldy + 4 * cq) = o;
  }
}
__global__ __launch_bounds__(256) void unary_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                        long long P, int C, int op, float* __restrict__ dx, int lddx) {
  const int CQ = C / 4;
  GRID_STRIDE(i, P * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    const f32x4 v = *(const f32x4*)(x + p * ldx + 4 * cq);
    const f32x4 g = *(const f32x4*)(dy + p * lddy + 4 * cq);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = v[e];
      float d;
      if (op == SSG_UNARY_SWISH) { const float s = sigm(z); d = s * (1.f + z * (1.f - s)); }      // utils.py:45-48
      else if (op == SSG_UNARY_SIGMOID) { const float s = sigm(z); d = s * (1.f - s); }
      else d = -2.f * z * expf(-(z * z));                                                          // xresidualblock.py:5-7
      o[e] = g[e] * d;
    }
    *(f32x4*)(dx + p * lddx + 4 * cq) = o;
  }
}
__global__ __launch_bounds__(256) void mul_fwd_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                      long long P, int C, float* __restrict__ y, int ldy) {
  const int CQ = C / 4;
  GRID_STRIDE(i, P * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    *(f32x4*)(y + p * ldy + 4 * cq) = *(const f32x4*)(a + p * lda + 4 * cq) * *(const f32x4*)(b + p * ldb + 4 * cq);
  }
}
__global__ __launch_bounds__(256) void mul_bwd_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                      const float* __restrict__ dy, int lddy, long long P, int C,
                                                      float* __restrict__ da, int ldda, float* __restrict__ db, int lddb) {
  const int CQ = C / 4;
  GRID_STRIDE(i, P * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    const f32x4 g = *(const f32x4*)(dy + p * lddy + 4 * cq);
    *(f32x4*)(da + p * ldda + 4 * cq) = g * *(const f32x4*)(b + p * ldb + 4 * cq);
    *(f32x4*)(db + p * lddb + 4 * cq) = g * *(const f32x4*)(a + p * lda + 4 * cq);
  }
}

// y[n,p,c] = x[n,p,c] * s[n,c]  (squeeze-excite gate / drop-connect scale); bwd: dx = g*s, ds = sum_p g*x
template <typename T>
__global__ __launch_bounds__(256) void chscale_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ s, long long S,
                                                          int N, int C, T* __restrict__ y, int ldy) {
  const int CQ = C / 4;
  GRID_STRIDE(i, (long long)N * S * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    const int n = (int)(p / S);
    st4(y + p * ldy + 4 * cq, ld4(x + p * ldx + 4 * cq) * *(const f32x4*)(s + (size_t)n * C + 4 * cq));
  }
}
// per-sample column reduction: out[n,c] = scale * sum_p a[n,p,c] * (b ? b[n,p,c] : 1).  Two deterministic stages:
// grid.z slices of the pixel range write fp64 partials [n][slice][C], a finishing kernel adds them in slice order
// (one slice per (channel block, sample) left 12 workgroups on a 600-MB tensor: 0.74 ms per call in EfficientNet-B4).
template <typename T>
__global__ __launch_bounds__(256) void sample_colsum_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb,
                                                            long long S, int C, long long rows_per_slice, double* __restrict__ part) {
  __shared__ double red[256][4];
  const int tid = threadIdx.x, tq = tid % DW_TQ, pr = tid / DW_TQ, PR = 256 / DW_TQ;
  const int CQ = C / 4, cq = blockIdx.x * DW_TQ + tq, n = blockIdx.y;
  const long long p0 = (long long)blockIdx.z * rows_per_slice;
  long long p1 = p0 + rows_per_slice; if (p1 > S) p1 = S;
  typedef typename SsgAcc<T>::type acc_t;
  acc_t s[4] = {0, 0, 0, 0};
  if (cq < CQ)
    for (long long p = p0 + pr; p < p1; p += PR) {
      const size_t row = (size_t)n * S + p;
      f32x4 v = ld4(a + row * lda + 4 * cq);
      if (b) v = v * ld4(b + row * ldb + 4 * cq);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += (acc_t)v[e];
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[tid][e] = (double)s[e];
  __syncthreads();
  if (pr == 0 && cq < CQ) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double t = 0;
      for (int r = 0; r < PR; ++r) t += red[r * DW_TQ + tq][e];
      part[((size_t)n * gridDim.z + blockIdx.z) * C + 4 * cq + e] = t;
    }
  }
}
__global__ __launch_bounds__(256) void sample_colsum_final_kernel(const double* __restrict__ part, int slices, int NC, int C, float scale,
                                                                  float* __restrict__ out) {
  const int lane = threadIdx.x & 31;
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
  double t = 0;
  if (i < NC) {
    const int n = i / C, c = i - n * C;
    for (int z = lane; z < slices; z += 32) t += part[((size_t)n * slices + z) * C + c];
  }
  t = fold32(t);
  if (lane == 0 && i < NC) out[i] = (float)(t * scale);
}
int sample_colsum_slices(int N, long long S, int C) {
  const long long blocks = (long long)((C / 4 + DW_TQ - 1) / DW_TQ) * N;
  long long z = 2048 / blocks;                          // ~8 workgroups per CU
  const long long maxz = S / (16 * (256 / DW_TQ));      // at least 16 rows per thread
  if (z > maxz) z = maxz;
  if (z > 1024) z = 1024;
  if (z < 1) z = 1;
  return (int)z;
}
template <typename T>
__global__ __launch_bounds__(256) void bcast_rows_kernel(const float* __restrict__ s, long long S, int N, int C, float scale,
                                                         T* __restrict__ y, int ldy) {
  const int CQ = C / 4;
  GRID_STRIDE(i, (long long)N * S * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    const int n = (int)(p / S);
    st4(y + p * ldy + 4 * cq, scale * *(const f32x4*)(s + (size_t)n * C + 4 * cq));
  }
}

int dw_check(const char* what, const void* in, int N, int H, int W, int C, int ld, int KH, int KW, int stride) {
  SSG_REQUIRE(in && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C, SSG_EINVAL, "%s: bad tensor", what);
  SSG_REQUIRE(KH >= 1 && KH <= 11 && KW >= 1 && KW <= 11 && stride >= 1 && stride <= 4, SSG_EINVAL, "%s: kernel/stride", what);
  return SSG_OK;
}

}  // namespace

namespace {

template <typename T>
int dwconv_fwd_impl(const T* in, int N, int H, int W, int C, int ld, const float* w, const float* bias, int KH, int KW,
                    int stride, int pad_top, int pad_left, int OH, int OW, T* out, int ldo, void* stream) {
  int rc = dw_check("dwconv_fwd", in, N, H, W, C, ld, KH, KW, stride);
  if (rc) return rc;
  SSG_REQUIRE(w && out && OH > 0 && OW > 0 && ldo % 4 == 0, SSG_EINVAL, "dwconv_fwd: bad output");
  if (stride == 1 && ssg_aligned16(in) && ssg_aligned16(out)) {
    const DwS1Args<T> s1{in, w, bias, out, N, H, W, OH, OW, C, ld, ldo, KH, pad_top, pad_left, 0};
    rc = dw_s1_dispatch<T>(s1, KW, (hipStream_t)stream);
    if (rc >= 0) return rc;
  }
  if (stride == 2 && KH == KW && ssg_aligned16(in) && ssg_aligned16(out)) {
    const DwS1Args<T> s2{in, w, bias, out, N, H, W, OH, OW, C, ld, ldo, KH, pad_top, pad_left, 0};
    rc = dw_s2_fwd_dispatch<T>(s2, KW, (hipStream_t)stream);
    if (rc >= 0) return rc;
  }
  DwArgs<T> a{in, w, bias, nullptr, out, N, H, W, C, ld, KH, KW, stride, pad_top, pad_left, OH, OW, ldo};
  hipLaunchKernelGGL(dw_fwd_kernel<T>, dim3(elem_grid((long long)N * OH * OW * (C / 4))), dim3(256), 0, (hipStream_t)stream, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

template <typename T>
int dwconv_dgrad_impl(const T* dout, int lddo, int N, int H, int W, int C, const float* w, int KH, int KW, int stride,
                      int pad_top, int pad_left, int OH, int OW, T* dx, int lddx, void* stream) {
  int rc = dw_check("dwconv_dgrad", dout, N, OH, OW, C, lddo, KH, KW, stride);
  if (rc) return rc;
  SSG_REQUIRE(w && dx && H > 0 && W > 0 && lddx % 4 == 0, SSG_EINVAL, "dwconv_dgrad: bad output");
  if (stride == 1 && ssg_aligned16(dout) && ssg_aligned16(dx)) {
    // dx = correlation of dout with the flipped kernel, pads K-1-pad
    const DwS1Args<T> s1{dout, w, nullptr, dx, N, OH, OW, H, W, C, lddo, lddx, KH, KH - 1 - pad_top, KW - 1 - pad_left, 1};
    rc = dw_s1_dispatch<T>(s1, KW, (hipStream_t)stream);
    if (rc >= 0) return rc;
  }
  if (stride == 2 && KH == KW && ssg_aligned16(dout) && ssg_aligned16(dx)) {
    const DwS1Args<T> s2{dout, w, nullptr, dx, N, OH, OW, H, W, C, lddo, lddx, KH, pad_top, pad_left, 0};
    rc = dw_s2_dgrad_dispatch<T>(s2, KW, (hipStream_t)stream);
    if (rc >= 0) return rc;
  }
  DwArgs<T> a{nullptr, w, nullptr, dout, dx, N, H, W, C, lddx, KH, KW, stride, pad_top, pad_left, OH, OW, lddo};
  hipLaunchKernelGGL(dw_dgrad_kernel<T>, dim3(elem_grid((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

template <typename T>
int dwconv_wgrad_impl(const T* in, int N, int H, int W, int C, int ld, const T* dout, int lddo, int KH, int KW,
                      int stride, int pad_top, int pad_left, int OH, int OW, float* dw, void* ws, void* stream) {
  int rc = dw_check("dwconv_wgrad", in, N, H, W, C, ld, KH, KW, stride);
  if (rc) return rc;
  SSG_REQUIRE(dout && dw && ws && OH > 0 && OW > 0, SSG_EINVAL, "dwconv_wgrad: bad args");
  const bool tiled = (stride == 1 && (KW == 3 || KW == 5 || KW == 7 || KW == 9)) || (stride == 2 && (KW == 3 || KW == 5));
  // work units: pixels, or x-quads of 4 consecutive output pixels for the tiled kernels; the partial buffer is sized for the
  // pixel count (ssg_dwconv2d_wgrad_workspace_bytes), which bounds both
  const long long P = tiled ? (long long)N * OH * ((OW + 3) / 4) : (long long)N * OH * OW;
  long long parts = (P + 255) / 256; if (parts > 256) parts = 256; if (parts < 1) parts = 1;
  const long long rpp = (P + parts - 1) / parts;
  parts = (P + rpp - 1) / rpp;
  DwArgs<T> a{in, nullptr, nullptr, dout, nullptr, N, H, W, C, ld, KH, KW, stride, pad_top, pad_left, OH, OW, lddo};
  hipStream_t st = (hipStream_t)stream;
  if (tiled) {
    if (KW == 3) launch_dw_wgrad_s1<3, T>(a, parts, rpp, (double*)ws, st);
    else if (KW == 5) launch_dw_wgrad_s1<5, T>(a, parts, rpp, (double*)ws, st);
    else if (KW == 7) launch_dw_wgrad_s1<7, T>(a, parts, rpp, (double*)ws, st);
    else launch_dw_wgrad_s1<9, T>(a, parts, rpp, (double*)ws, st);
  } else
  hipLaunchKernelGGL(dw_wgrad_partial_kernel<T>, dim3((unsigned)parts, (unsigned)((C / 4 + DW_TQ - 1) / DW_TQ), (unsigned)(KH * KW)), dim3(256), 0, st,
                     a, rpp, (double*)ws);
  SSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(dw_wgrad_final_kernel, dim3((unsigned)((KH * KW * C + 7) / 8)), dim3(256), 0, st, (const double*)ws, (int)parts, KH * KW, C, dw);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

}  // namespace

extern "C" int ssg_dwconv2d_fwd_f32(const float* in, int N, int H, int W, int C, int ld, const float* w, const float* bias, int KH, int KW,
                                    int stride, int pad_top, int pad_left, int OH, int OW, float* out, int ldo, void* stream) {
  return dwconv_fwd_impl<float>(in, N, H, W, C, ld, w, bias, KH, KW, stride, pad_top, pad_left, OH, OW, out, ldo, stream);
}
extern "C" int ssg_dwconv2d_fwd_bf16(const void* in, int N, int H, int W, int C, int ld, const float* w, const float* bias, int KH, int KW,
                                     int stride, int pad_top, int pad_left, int OH, int OW, void* out, int ldo, void* stream) {
  return dwconv_fwd_impl<ssg_bf16>((const ssg_bf16*)in, N, H, W, C, ld, w, bias, KH, KW, stride, pad_top, pad_left, OH, OW, (ssg_bf16*)out, ldo, stream);
}

extern "C" int ssg_dwconv2d_dgrad_f32(const float* dout, int lddo, int N, int H, int W, int C, const float* w, int KH, int KW, int stride,
                                      int pad_top, int pad_left, int OH, int OW, float* dx, int lddx, void* stream) {
  return dwconv_dgrad_impl<float>(dout, lddo, N, H, W, C, w, KH, KW, stride, pad_top, pad_left, OH, OW, dx, lddx, stream);
}
extern "C" int ssg_dwconv2d_dgrad_bf16(const void* dout, int lddo, int N, int H, int W, int C, const float* w, int KH, int KW, int stride,
                                       int pad_top, int pad_left, int OH, int OW, void* dx, int lddx, void* stream) {
  return dwconv_dgrad_impl<ssg_bf16>((const ssg_bf16*)dout, lddo, N, H, W, C, w, KH, KW, stride, pad_top, pad_left, OH, OW, (ssg_bf16*)dx, lddx, stream);
}

extern "C" int64_t ssg_dwconv2d_wgrad_workspace_bytes(int N, int OH, int OW, int C, int KH, int KW) {
  long long P = (long long)N * OH * OW;
  long long parts = (P + 255) / 256; if (parts > 256) parts = 256; if (parts < 1) parts = 1;
  return parts * KH * KW * (int64_t)C * (int64_t)sizeof(double);
}

extern "C" int ssg_dwconv2d_wgrad_f32(const float* in, int N, int H, int W, int C, int ld, const float* dout, int lddo, int KH, int KW,
                                      int stride, int pad_top, int pad_left, int OH, int OW, float* dw, void* ws, void* stream) {
  return dwconv_wgrad_impl<float>(in, N, H, W, C, ld, dout, lddo, KH, KW, stride, pad_top, pad_left, OH, OW, dw, ws, stream);
}
extern "C" int ssg_dwconv2d_wgrad_bf16(const void* in, int N, int H, int W, int C, int ld, const void* dout, int lddo, int KH, int KW,
                                       int stride, int pad_top, int pad_left, int OH, int OW, float* dw, void* ws, void* stream) {
  return dwconv_wgrad_impl<ssg_bf16>((const ssg_bf16*)in, N, H, W, C, ld, (const ssg_bf16*)dout, lddo, KH, KW, stride, pad_top, pad_left, OH, OW, dw, ws,
                                     stream);
}

extern "C" int ssg_unary_fwd_f32(const float* x, int ldx, int64_t P, int C, int op, float* y, int ldy, void* stream) {
  SSG_REQUIRE(x && y && P > 0 && C > 0 && C % 4 == 0 && op >= 0 && op <= 2, SSG_EINVAL, "unary_fwd: bad args");
  hipLaunchKernelGGL(unary_fwd_kernel, dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, (long long)P, C, op, y, ldy);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_unary_bwd_f32(const float* x, int ldx, const float* dy, int lddy, int64_t P, int C, int op, float* dx, int lddx, void* stream) {
  SSG_REQUIRE(x && dy && dx && P > 0 && C > 0 && C % 4 == 0 && op >= 0 && op <= 2, SSG_EINVAL, "unary_bwd: bad args");
  hipLaunchKernelGGL(unary_bwd_kernel, dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, lddy, (long long)P, C, op, dx, lddx);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_mul_fwd_f32(const float* a, int lda, const float* b, int ldb, int64_t P, int C, float* y, int ldy, void* stream) {
  SSG_REQUIRE(a && b && y && P > 0 && C > 0 && C % 4 == 0, SSG_EINVAL, "mul_fwd: bad args");
  hipLaunchKernelGGL(mul_fwd_kernel, dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, (long long)P, C, y, ldy);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_mul_bwd_f32(const float* a, int lda, const float* b, int ldb, const float* dy, int lddy, int64_t P, int C,
                               float* da, int ldda, float* db, int lddb, void* stream) {
  SSG_REQUIRE(a && b && dy && da && db && P > 0 && C > 0 && C % 4 == 0, SSG_EINVAL, "mul_bwd: bad args");
  hipLaunchKernelGGL(mul_bwd_kernel, dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, dy, lddy, (long long)P, C, da, ldda, db, lddb);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
namespace {
template <typename T>
int channel_scale_impl(const T* x, int ldx, const float* s, int N, int64_t S, int C, T* y, int ldy, void* stream) {
  SSG_REQUIRE(x && s && y && N > 0 && S > 0 && C > 0 && C % 4 == 0, SSG_EINVAL, "channel_scale: bad args");
  hipLaunchKernelGGL(chscale_fwd_kernel<T>, dim3(elem_grid((long long)N * S * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, s, (long long)S, N, C, y, ldy);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
template <typename T>
int sample_channel_sum_impl(const T* a, int lda, const T* b, int ldb, int N, int64_t S, int C, float scale, float* out, void* ws, void* stream) {
  SSG_REQUIRE(a && out && ws && N > 0 && S > 0 && C > 0 && C % 4 == 0, SSG_EINVAL, "sample_channel_sum: bad args");
  const int z = sample_colsum_slices(N, S, C);
  const long long rps = (S + z - 1) / z;
  hipLaunchKernelGGL(sample_colsum_kernel<T>, dim3((unsigned)((C / 4 + DW_TQ - 1) / DW_TQ), (unsigned)N, (unsigned)z), dim3(256), 0, (hipStream_t)stream,
                     a, lda, b, ldb, (long long)S, C, rps, (double*)ws);
  SSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(sample_colsum_final_kernel, dim3((unsigned)((N * C + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (const double*)ws, z, N * C, C,
                     scale, out);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
template <typename T>
int broadcast_rows_impl(const float* s, int N, int64_t S, int C, float scale, T* y, int ldy, void* stream) {
  SSG_REQUIRE(s && y && N > 0 && S > 0 && C > 0 && C % 4 == 0, SSG_EINVAL, "broadcast_rows: bad args");
  hipLaunchKernelGGL(bcast_rows_kernel<T>, dim3(elem_grid((long long)N * S * (C / 4))), dim3(256), 0, (hipStream_t)stream, s, (long long)S, N, C, scale, y, ldy);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
}  // namespace

extern "C" int ssg_channel_scale_fwd_f32(const float* x, int ldx, const float* s, int N, int64_t S, int C, float* y, int ldy, void* stream) {
  return channel_scale_impl<float>(x, ldx, s, N, S, C, y, ldy, stream);
}
extern "C" int ssg_channel_scale_fwd_bf16(const void* x, int ldx, const float* s, int N, int64_t S, int C, void* y, int ldy, void* stream) {
  return channel_scale_impl<ssg_bf16>((const ssg_bf16*)x, ldx, s, N, S, C, (ssg_bf16*)y, ldy, stream);
}
extern "C" int64_t ssg_sample_channel_sum_workspace_bytes(int N, int64_t S, int C) {
  return (int64_t)N * sample_colsum_slices(N, S, C) * C * (int64_t)sizeof(double);
}
extern "C" int ssg_sample_channel_sum_f32(const float* a, int lda, const float* b, int ldb, int N, int64_t S, int C, float scale, float* out,
                                          void* ws, void* stream) {
  return sample_channel_sum_impl<float>(a, lda, b, ldb, N, S, C, scale, out, ws, stream);
}
extern "C" int ssg_sample_channel_sum_bf16(const void* a, int lda, const void* b, int ldb, int N, int64_t S, int C, float scale, float* out,
                                           void* ws, void* stream) {
  return sample_channel_sum_impl<ssg_bf16>((const ssg_bf16*)a, lda, (const ssg_bf16*)b, ldb, N, S, C, scale, out, ws, stream);
}
extern "C" int ssg_broadcast_rows_f32(const float* s, int N, int64_t S, int C, float scale, float* y, int ldy, void* stream) {
  return broadcast_rows_impl<float>(s, N, S, C, scale, y, ldy, stream);
}
extern "C" int ssg_broadcast_rows_bf16(const float* s, int N, int64_t S, int C, float scale, void* y, int ldy, void* stream) {
  return broadcast_rows_impl<ssg_bf16>(s, N, S, C, scale, (ssg_bf16*)y, ldy, stream);
}
