// Epilogue of the LDS-halo conv kernels (conv_igemm_halo.hip, conv_igemm_halo_x3.hip): bias / residual / activation / batch-norm
// partial rows / split-K slabs from 32x32 accumulator fragments.  C/D map of a 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma once
#include "common.h"
#include "conv_args.h"

template <int BM, int BN, int WAVES_M, int WAVES_N, int TWL, bool SPLIT>
__device__ __forceinline__ void ssg_halo_epilogue(const ConvArgs& a, f32x16 (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32], float* lds,
                                                  int n, int ty, int tx, int n0, int slab, int wm, int wn, int half, int l31) {
  constexpr int TW = 1 << TWL, TH = BM / TW;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  if (SPLIT) {                                           // raw partial sums of this slab; the reduce kernel finishes them
    const int ldw = (a.Cout + 3) & ~3;
    float* wsl = a.ws + (size_t)slab * a.N * a.GH * a.GW * ldw;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int co = n0 + wn * WTN + j * 32 + l31;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const int gy = ty * TH + (p >> TWL), gx = tx * TW + (p & (TW - 1));
          if (gy < a.GH && gx < a.GW && co < ldw) wsl[((size_t)(n * a.GH + gy) * a.GW + gx) * ldw + co] = co < a.Cout ? acc[i][j][r] : 0.f;
        }
      }
    }
    return;
  }
  const bool want_bn = a.bnpart != nullptr;
  if (want_bn) ssg_bnpart_begin();
  // Column sums for the batch-norm statistics.  var = E[x^2] - mean^2 cancels, so plain fp32 partials lose it once |mean| >> std
  // (ADVICE r2).  Each lane sums DEVIATIONS from a pivot (its first output of the column) in fp32 -- their rounding error is
  // relative to |v - pivot|, not |v| -- and converts to sums of v in fp64 once per column: S1 = s1 + n*c, S2 = s2 + 2*c*s1 + n*c^2.
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = n0 + wn * WTN + j * 32 + l31;
    const bool cok = co < a.Cout;
    const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
    float s1 = 0.f, s2 = 0.f; int nv = 0;
    const float piv = acc[0][j][0] + bv;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int gy = ty * TH + (p >> TWL), gx = tx * TW + (p & (TW - 1));
        if (gy < a.GH && gx < a.GW) {
          const size_t pix = ((size_t)(n * a.OH + gy * a.out_sy + a.out_oy) * a.OW + gx * a.out_sx + a.out_ox);
          float v = acc[i][j][r] + bv;
          if (want_bn) { const float dv = v - piv; s1 += dv; s2 += dv * dv; ++nv; }
          if (cok) {
            if (a.res) v += a.res[pix * a.ldr + co];
            if (a.act == SSG_ACT_RELU) v = v < 0.f ? 0.f : v;
            else if (a.act == SSG_ACT_LRELU) v = v > 0.f ? v : v * a.slope;
            a.out[pix * a.ldo + co] = v;
          } else if (co < ((a.Cout + 3) & ~3)) {
            a.out[pix * a.ldo + co] = 0.f;
          }
        }
      }
    }
    if (want_bn) ssg_bnpart_put<BN, WTN>(lds, j, s1, s2, piv, nv, wm, wn, half, l31);
  }
  if (want_bn) ssg_bnpart_finish<NI, WAVES_M, BN, WTN>(a, lds, (n * a.tiles_y + ty) * a.tiles_x + tx, n0, wm, wn, half, l31);
}
