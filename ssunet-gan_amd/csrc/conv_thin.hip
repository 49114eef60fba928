// "Thin" convolutions for gfx950: one side of the conv has <= 8 channels (SPADE's C->3->h->C
// chain, the 3-channel image/logit/mask layers, the final 1x1).  On the MFMA path these waste
// 8-16x of the array on padding and are really HBM-bound streaming ops, so they run on the VALU
// with the wide tensor read/written exactly once in whole 16-byte-per-lane rows:
//   T1 small-Cout : 16 lanes share one pixel (4 channels each), per-lane partial dot products for
//                   <= 8 outputs, combined with 4 xor-shuffles; weights sit in LDS as [tap][co][C].
//   T2 small-Cin  : Cout/4 lanes share one pixel, each owns 4 output channels; the <= 8 input
//                   channels of a tap are one or two broadcast 16-B loads; weights in LDS [k][Cout].
//   T3/T4 wgrad   : the same two lane mappings with the 9x4x4 gradient block of a lane held in
//                   registers over a pixel range, combined across pixel groups by shuffles + LDS,
//                   written as split-K slabs for the ordered (deterministic) reducer.
// ABI: reached through ssg_conv2d_f32 / ssg_conv2d_wgrad_f32 (include/ssunet_hip.h).
#include "common.h"
#include "conv_thin.h"
#include <stdlib.h>

namespace {

struct ThinArgs {
  const float* in; const float* w; const float* bias; const float* res; float* out;
  int C, ld, N, H, W, Kp, kmode, ldr, Cout, ldo, ntaps;
  unsigned long long tap_bits;
  int act; float slope;
};

__device__ __forceinline__ int kidx(int kmode, int ntaps, int C, int t, int c) {
  return kmode == 0 ? ((c >> 4) * ntaps * 16 + t * 16 + (c & 15)) : (t * C + c);
}

// ---------------------------------------------------------------- T1: small Cout (<= CO)
template <int CO>
__global__ __launch_bounds__(256) void thin_small_cout_kernel(const ThinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wl[];          // [ntaps][CO][C]
  const int tid = threadIdx.x;
  for (int i = tid; i < a.ntaps * CO * a.C; i += 256) {
    const int c = i % a.C; const int r = i / a.C; const int co = r % CO; const int t = r / CO;
    wl[i] = co < a.Cout ? a.w[(size_t)co * a.Kp + kidx(a.kmode, a.ntaps, a.C, t, c)] : 0.f;
  }
  __syncthreads();
  const int l16 = tid & 15, pg = tid >> 4;
  const int tiles_x = (a.W + 15) >> 4, tiles_y = (a.H + 7) >> 3;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y; const int n = b / tiles_y;
  const int x = tx * 16 + pg;
  for (int r = 0; r < 8; ++r) {
    const int y = ty * 8 + r;
    const bool pok = (x < a.W) && (y < a.H);
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;
    for (int cc = 0; cc < a.C; cc += 64) {
      const int c = cc + 4 * l16;
      if (c < a.C && pok) {
        for (int t = 0; t < a.ntaps; ++t) {
          const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
          const int iy = y + (tb & 7) - 2, ix = x + (tb >> 3) - 2;
          if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            const f32x4 v = *(const f32x4*)(a.in + ((size_t)(n * a.H + iy) * a.W + ix) * a.ld + c);
            const float* wp = wl + (size_t)t * CO * a.C + c;
#pragma unroll
            for (int co = 0; co < CO; ++co) {
              const f32x4 w4 = *(const f32x4*)(wp + co * a.C);
              acc[co] += v[0] * w4[0] + v[1] * w4[1] + v[2] * w4[2] + v[3] * w4[3];
            }
          }
        }
      }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      float v = acc[co];
      v += __shfl_xor(v, 8, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 1, 16);
      acc[co] = v;
    }
    if (pok && l16 == 0) {
      const size_t pix = (size_t)(n * a.H + y) * a.W + x;
      const int cpad = (a.Cout + 3) & ~3;
#pragma unroll
      for (int q = 0; q < CO / 4; ++q) {
        if (4 * q < cpad) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int co = 4 * q + e;
            float v = 0.f;
            if (co < a.Cout) {
              v = acc[co] + (a.bias ? a.bias[co] : 0.f);
              if (a.res) v += a.res[pix * a.ldr + co];
              v = ssg_act(v, a.act, a.slope);
            }
            o[e] = v;
          }
          *(f32x4*)(a.out + pix * a.ldo + 4 * q) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- T2: small Cin (CI = 4 or 8)
template <int CI>
__global__ __launch_bounds__(256) void thin_small_cin_kernel(const ThinArgs a, int coq_lanes, long long npix) {
  extern __shared__ __attribute__((aligned(16))) float wl[];          // [ntaps*CI][CoutP]
  const int tid = threadIdx.x;
  const int CoutP = coq_lanes * 4;
  const int K = a.ntaps * CI;
  for (int i = tid; i < K * CoutP; i += 256) {
    const int co = i % CoutP, k = i / CoutP;
    const int t = k / CI, c = k - t * CI;
    wl[i] = co < a.Cout ? a.w[(size_t)co * a.Kp + kidx(a.kmode, a.ntaps, CI, t, c)] : 0.f;
  }
  __syncthreads();
  const int coq = tid % coq_lanes, pg = tid / coq_lanes, ppb = 256 / coq_lanes;
  const int cpad = (a.Cout + 3) & ~3;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (a.bias)
#pragma unroll
    for (int e = 0; e < 4; ++e) if (4 * coq + e < a.Cout) bv[e] = a.bias[4 * coq + e];
  const int HW = a.H * a.W;
  for (unsigned p = blockIdx.x * ppb + pg; p < (unsigned)npix; p += gridDim.x * ppb) {
    const int n = (int)(p / (unsigned)HW); const int rem = (int)(p - (unsigned)n * (unsigned)HW);
    const int y = rem / a.W, x = rem - y * a.W;
    f32x4 acc = bv;
    for (int t = 0; t < a.ntaps; ++t) {
      const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
      const int iy = y + (tb & 7) - 2, ix = x + (tb >> 3) - 2;
      if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
        const float* ip = a.in + ((size_t)(n * a.H + iy) * a.W + ix) * a.ld;
        const float* wp = wl + (size_t)t * CI * CoutP + 4 * coq;
#pragma unroll
        for (int q = 0; q < CI / 4; ++q) {
          const f32x4 v = *(const f32x4*)(ip + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc += v[e] * *(const f32x4*)(wp + (4 * q + e) * CoutP);
        }
      }
    }
    if (4 * coq < cpad) {
      if (a.res) {
        const float* rp = a.res + (size_t)p * a.ldr + 4 * coq;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (4 * coq + e < a.Cout) acc[e] += rp[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = (4 * coq + e < a.Cout) ? ssg_act(acc[e], a.act, a.slope) : 0.f;
      *(f32x4*)(a.out + (size_t)p * a.ldo + 4 * coq) = acc;
    }
  }
}

// ---------------------------------------------------------------- T3: wgrad, dout has <= 4 channels
struct ThinWgArgs {
  const float* in; const float* dout; float* ws;
  int C, ld, N, H, W, Cout, ldd, ntaps; unsigned long long tap_bits;
  long long npix, pix_per_block;
};

__global__ __launch_bounds__(256) void thin_wgrad_small_cout_kernel(const ThinWgArgs a) {
  // lanes: 16 per pixel group (4 channels each) of the 64-channel chunk blockIdx.y; 16 pixel groups
  __shared__ float red[4][16][144 + 1];
  const int tid = threadIdx.x, l16 = tid & 15, pg = tid >> 4, wave = tid >> 6;
  const int c = blockIdx.y * 64 + 4 * l16;
  const bool cok = c < a.C;
  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int co = 0; co < 4; ++co) acc[t][co] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long p0 = (long long)blockIdx.x * a.pix_per_block;
  long long p1 = p0 + a.pix_per_block; if (p1 > a.npix) p1 = a.npix;
  const int HW = a.H * a.W;
  for (unsigned p = (unsigned)p0 + pg; p < (unsigned)p1; p += 16) {
    const int n = (int)(p / (unsigned)HW); const int rem = (int)(p - (unsigned)n * (unsigned)HW);
    const int y = rem / a.W, x = rem - y * a.W;
    const f32x4 d = *(const f32x4*)(a.dout + (size_t)p * a.ldd);
    if (cok) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t < a.ntaps) {
          const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
          const int iy = y + (tb & 7) - 2, ix = x + (tb >> 3) - 2;
          if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            const f32x4 v = *(const f32x4*)(a.in + ((size_t)(n * a.H + iy) * a.W + ix) * a.ld + c);
#pragma unroll
            for (int co = 0; co < 4; ++co) acc[t][co] += d[co] * v;
          }
        }
      }
    }
  }
  // combine the 4 pixel groups of a wave (lanes +16, +32), then the 4 waves through LDS
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[t][co][e];
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        acc[t][co][e] = v;
      }
  if ((tid & 63) < 16) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int co = 0; co < 4; ++co)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][l16][(t * 4 + co) * 4 + e] = acc[t][co][e];
  }
  __syncthreads();
  // slab layout (shared with wgrad_reduce_kernel): [split][row = t*C + c][Cout]
  float* slab = a.ws + (size_t)blockIdx.x * a.ntaps * a.C * a.Cout;
  for (int i = tid; i < 16 * 144; i += 256) {
    const int l = i / 144, j = i - l * 144;
    const int t = j / 16, co = (j >> 2) & 3, e = j & 3;
    const int cc = blockIdx.y * 64 + 4 * l + e;
    if (t < a.ntaps && co < a.Cout && cc < a.C)
      slab[((size_t)t * a.C + cc) * a.Cout + co] = (red[0][l][j] + red[1][l][j]) + (red[2][l][j] + red[3][l][j]);
  }
}

// ---------------------------------------------------------------- T4: wgrad, in has 4 channels
__global__ __launch_bounds__(256) void thin_wgrad_small_cin_kernel(const ThinWgArgs a, int coq_lanes) {
  extern __shared__ __attribute__((aligned(16))) float red2[];       // [waves or groups][144][CoutP] staged in rounds
  const int tid = threadIdx.x;
  const int coq = tid % coq_lanes, pg = tid / coq_lanes, ngroups = 256 / coq_lanes;
  const bool cok = 4 * coq < ((a.Cout + 3) & ~3);
  f32x4 acc[9][4];                                                    // [tap][ci] x 4 couts
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) acc[t][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long p0 = (long long)blockIdx.x * a.pix_per_block;
  long long p1 = p0 + a.pix_per_block; if (p1 > a.npix) p1 = a.npix;
  const int HW = a.H * a.W;
  for (unsigned p = (unsigned)p0 + pg; p < (unsigned)p1; p += ngroups) {
    const int n = (int)(p / (unsigned)HW); const int rem = (int)(p - (unsigned)n * (unsigned)HW);
    const int y = rem / a.W, x = rem - y * a.W;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    if (cok) d = *(const f32x4*)(a.dout + (size_t)p * a.ldd + 4 * coq);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t < a.ntaps) {
        const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
        const int iy = y + (tb & 7) - 2, ix = x + (tb >> 3) - 2;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
          const f32x4 v = *(const f32x4*)(a.in + ((size_t)(n * a.H + iy) * a.W + ix) * a.ld);
#pragma unroll
          for (int ci = 0; ci < 4; ++ci) acc[t][ci] += v[ci] * d;
        }
      }
    }
  }
  // reduce across the pixel groups: groups inside one wave by xor-shuffles, waves through LDS
  const int CoutP = coq_lanes * 4;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[t][ci][e];
        for (int off = coq_lanes; off < 64; off <<= 1) v += __shfl_xor(v, off);
        acc[t][ci][e] = v;
      }
  // after the shuffles every wave holds one full set of sums (lane -> cout quad = lane % coq_lanes)
  const int wave = tid >> 6;
  const bool writer = (tid & 63) < coq_lanes;
  float* slab = a.ws + (size_t)blockIdx.x * a.ntaps * 4 * a.Cout;
  // LDS staging per tap to bound the footprint: [4 waves][4 ci][CoutP]
  for (int t = 0; t < a.ntaps; ++t) {
    __syncthreads();
    if (writer) {
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) if (tt == t) v = acc[tt][ci];
        *(f32x4*)(red2 + ((size_t)(wave * 4 + ci) * CoutP) + 4 * (tid & 63)) = v;
      }
    }
    __syncthreads();
    for (int i = tid; i < 4 * CoutP; i += 256) {
      const int ci = i / CoutP, co = i - ci * CoutP;
      const float s = (red2[(size_t)(0 * 4 + ci) * CoutP + co] + red2[(size_t)(1 * 4 + ci) * CoutP + co]) +
                      (red2[(size_t)(2 * 4 + ci) * CoutP + co] + red2[(size_t)(3 * 4 + ci) * CoutP + co]);
      if (co < a.Cout) slab[((size_t)t * 4 + ci) * a.Cout + co] = s;
    }
  }
}

ThinArgs make_args(const ssg_conv_desc* d) {
  ThinArgs a;
  a.in = d->in1; a.w = d->w; a.bias = d->bias; a.res = d->res; a.out = d->out;
  a.C = d->C1; a.ld = d->ld1; a.N = d->N; a.H = d->H; a.W = d->W; a.Kp = d->Kp; a.kmode = d->kmode;
  a.ldr = d->ldr; a.Cout = d->Cout; a.ldo = d->ldo; a.ntaps = d->ntaps;
  a.tap_bits = 0;
  for (int t = 0; t < d->ntaps; ++t)
    a.tap_bits |= (unsigned long long)(((d->dy[t] + 2) & 7) | (((d->dx[t] + 2) & 7) << 3)) << (6 * t);
  a.act = d->act; a.slope = d->slope;
  return a;
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Which thin kernels are enabled: bit0 = T1 small-Cout conv, bit1 = T2 small-Cin conv, bit2 = T3
// wgrad (dout thin), bit3 = T4 wgrad (in thin).  Round-1 measurements at 16x512^2 (ms per launch,
// thin vs MFMA path): T1 0.93 vs 1.85 (64->4); T2 0.99 vs 0.92 (3->64), 0.83 vs 0.58 (4->64);
// T3 4.0 vs 2.6 (64->3); T4 1.63 vs 1.0 (3->64).  Only T1 wins (and is ~6x closer to an fp64
// reference than the MFMA path: 16-lane partial sums + a shuffle tree instead of one long fmaf
// chain), so T1 is ON by default and T2-T4 are opt-in through SSG_THIN_MASK; all four are kept
// exact by tests/test_ops_gpu.py (which enables them in a subprocess).
int thin_mask() {
  static int m = -1;
  if (m < 0) {
    const char* e = getenv("SSG_THIN_MASK");
    m = e ? atoi(e) : 1;
  }
  return m;
}

}  // namespace

// 0 = not a thin case, 1 = small-Cout (T1), 2 = small-Cin (T2)
int ssg_thin_conv_kind(const ssg_conv_desc* d) {
  if ((long long)d->N * d->H * d->W >= (1ll << 31)) return 0;
  if (d->C2 != 0 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1 || d->out_oy || d->out_ox) return 0;
  if (d->GH != d->H || d->GW != d->W || d->OH != d->H || d->OW != d->W || d->bnpart) return 0;
  if ((thin_mask() & 1) && d->Cout <= 8 && d->C1 >= 16 && d->C1 <= 256 && d->C1 % 4 == 0) return 1;
  const int coq = (d->Cout + 3) / 4;
  if ((thin_mask() & 2) && (d->C1 == 4 || d->C1 == 8) && pow2(coq) && coq <= 64 && d->ldo % 4 == 0) return 2;
  return 0;
}

int ssg_thin_conv_launch(const ssg_conv_desc* d, int kind, hipStream_t st) {
  const ThinArgs a = make_args(d);
  if (kind == 1) {
    const int tiles = ((d->W + 15) / 16) * ((d->H + 7) / 8) * d->N;
    if (d->Cout <= 4) {
      const size_t sh = (size_t)d->ntaps * 4 * d->C1 * sizeof(float);
      hipLaunchKernelGGL(thin_small_cout_kernel<4>, dim3((unsigned)tiles), dim3(256), sh, st, a);
    } else {
      const size_t sh = (size_t)d->ntaps * 8 * d->C1 * sizeof(float);
      if (sh > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)thin_small_cout_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (e != hipSuccess) { ssg_set_error("thin conv: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
      }
      hipLaunchKernelGGL(thin_small_cout_kernel<8>, dim3((unsigned)tiles), dim3(256), sh, st, a);
    }
  } else {
    const int coq = (d->Cout + 3) / 4;
    const long long npix = (long long)d->N * d->H * d->W;
    const int ppb = 256 / coq;
    long long grid = (npix + ppb - 1) / ppb;
    if (grid > 256 * 12) grid = 256 * 12;
    const size_t sh = (size_t)d->ntaps * d->C1 * coq * 4 * sizeof(float);
    if (d->C1 == 4) hipLaunchKernelGGL(thin_small_cin_kernel<4>, dim3((unsigned)grid), dim3(256), sh, st, a, coq, npix);
    else hipLaunchKernelGGL(thin_small_cin_kernel<8>, dim3((unsigned)grid), dim3(256), sh, st, a, coq, npix);
  }
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

// wgrad: 0 = no, 3 = dout thin (T3), 4 = in thin (T4)
int ssg_thin_wgrad_kind(const ssg_wgrad_desc* d) {
  if ((long long)d->N * d->H * d->W >= (1ll << 31)) return 0;
  if (d->C2 != 0 || d->in_sy != 1 || d->in_sx != 1 || d->GH != d->H || d->GW != d->W) return 0;
  if ((thin_mask() & 4) && d->Cout <= 4 && d->C1 >= 16 && d->C1 % 4 == 0 && d->C1 <= 1024) return 3;
  const int coq = (d->Cout + 3) / 4;
  if ((thin_mask() & 8) && d->C1 == 4 && pow2(coq) && coq <= 64) return 4;
  return 0;
}

int ssg_thin_wgrad_splits(const ssg_wgrad_desc* d, long long* pix_per_block) {
  const long long npix = (long long)d->N * d->H * d->W;
  long long splits = 1024;
  long long ppb = (npix + splits - 1) / splits;
  if (ppb < 64) ppb = 64;
  ppb = (ppb + 15) / 16 * 16;
  splits = (npix + ppb - 1) / ppb;
  if (pix_per_block) *pix_per_block = ppb;
  return (int)splits;
}

int ssg_thin_wgrad_launch(const ssg_wgrad_desc* d, int kind, hipStream_t st) {
  ThinWgArgs a;
  a.in = d->in1; a.dout = d->dout; a.ws = d->ws; a.C = d->C1; a.ld = d->ld1; a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cout = d->Cout; a.ldd = d->ldd; a.ntaps = d->ntaps;
  a.tap_bits = 0;
  for (int t = 0; t < d->ntaps; ++t)
    a.tap_bits |= (unsigned long long)(((d->dy[t] + 2) & 7) | (((d->dx[t] + 2) & 7) << 3)) << (6 * t);
  a.npix = (long long)d->N * d->H * d->W;
  const int splits = ssg_thin_wgrad_splits(d, &a.pix_per_block);
  if (kind == 3) {
    hipLaunchKernelGGL(thin_wgrad_small_cout_kernel, dim3((unsigned)splits, (unsigned)((d->C1 + 63) / 64)), dim3(256), 0, st, a);
  } else {
    const int coq = (d->Cout + 3) / 4;
    const size_t sh = (size_t)16 * coq * 4 * sizeof(float);
    hipLaunchKernelGGL(thin_wgrad_small_cin_kernel, dim3((unsigned)splits), dim3(256), sh, st, a, coq);
  }
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
