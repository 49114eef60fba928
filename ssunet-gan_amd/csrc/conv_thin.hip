// "Thin" convolutions for gfx950: one side of the conv has <= 8 channels (SPADE's C->3->h->C
// chain, the 3-channel image/logit/mask layers, the final 1x1).  On the MFMA path these waste
// 8-16x of the array on padding and are really HBM-bound streaming ops, so they run on the VALU
// with the wide tensor read/written exactly once in whole 16-byte-per-lane rows:
//   T1 small-Cout : 16 lanes share one pixel (4 channels each), per-lane partial dot products for
//                   <= 8 outputs, combined with 4 xor-shuffles; weights sit in LDS as [tap][co][C].
// (Round 1 also carried VALU kernels for a <= 8-channel INPUT and for the two thin weight gradients, T2-T4: measured
//  slower than the 4x4x1-MFMA kernels of conv_thin4.hip / conv_wgrad4.hip on every shape, and removed in round 2.)
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

// SSG_THIN_MASK=0 switches the kernel off (A/B against the padded-MFMA path).  Round-1 measurement at 16x512^2, 64->4:
// 0.93 vs 1.85 ms per launch, and ~6x closer to an fp64 reference than the MFMA path (16-lane partial sums + a shuffle
// tree instead of one long fmaf chain).
int thin_mask() {
  static int m = -1;
  if (m < 0) {
    const char* e = getenv("SSG_THIN_MASK");
    m = e ? atoi(e) : 1;
  }
  return m;
}

}  // namespace

// 0 = not a thin case, 1 = small-Cout (T1)
int ssg_thin_conv_kind(const ssg_conv_desc* d) {
  if ((long long)d->N * d->H * d->W >= (1ll << 31)) return 0;
  if (d->C2 != 0 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1 || d->out_oy || d->out_ox) return 0;
  if (d->GH != d->H || d->GW != d->W || d->OH != d->H || d->OW != d->W) return 0;
  if ((thin_mask() & 1) && d->Cout <= 8 && d->C1 >= 16 && d->C1 <= 256 && d->C1 % 4 == 0) return 1;
  return 0;
}

int ssg_thin_conv_launch(const ssg_conv_desc* d, int kind, hipStream_t st) {
  (void)kind;
  const ThinArgs a = make_args(d);
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
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
