// Non-finite operands on the split-operand (bf16x3) kernels.  x = x1 + x2 + x3 with bf16 terms turns x = +-inf into (inf, NaN, NaN),
// a finite |x| >= 2^128 - 2^119 (rounds to inf in bf16) into (inf, -inf, NaN), and even a sanitised (inf, 0, 0) meets inf * 0 = NaN
// against the zero low-order terms of any bf16-representable weight -- where the fp32 kernels (and the reference: losses.py:297-300
// keys on inf vs NaN, train_seg_gan.py:190 zeroes NaN only) produce +-inf or a finite value.  So the split kernels do not try to be
// right on such operands, they DETECT them: any such operand makes every accumulator it touches non-finite (inf * w is inf or
// NaN), so after the main loop a workgroup that holds a non-finite accumulator recomputes all its accumulators with plain fp32
// FMAs on the fp32 operands (same taps, same zero padding; the result class -- finite / +inf / -inf / NaN -- is the fp32 kernel's)
// and continues into the normal epilogue.  Costs one __syncthreads_or per tile when nothing is wrong; tiles that take the slow
// path run ~30x longer (a diverged run, not a production case).  tests/test_split_gpu.py::test_split_kernels_on_nonfinite_operands.
#pragma once
#include "common.h"
#include "conv_args.h"
#include "conv_wgrad_args.h"

__device__ __forceinline__ bool ssg_nonfinite(float v) { return !(__builtin_fabsf(v) <= 3.4028234663852886e38f); }

// One pre-epilogue accumulator value of a convolution launch: sum over taps [t_lo, t_hi) and all input channels of
// in[n][gy*in_sy + dy_t][gx*in_sx + dx_t][c] * w32[co][k(t, c)], fp32 FMAs, out-of-image taps contribute nothing.
// w32 = the fp32 packed weights [Cout][Kp] in kmode 0 (k = (c / 16 * ntaps + t) * 16 + c % 16).
// XF: the launch carries a fused input transform (ssg_conv_desc.in_scale); a template parameter, not a run-time test of a.in_scale --
// the test alone cost the k32 kernels 8 registers of their hot loops
template <bool XF = false>
__device__ inline float ssg_conv_slow_value(const ConvArgs& a, int n, int gy, int gx, int co, int t_lo, int t_hi) {
  if (co >= a.Cout) return 0.f;
  const float* wrow = a.w32 + (size_t)co * a.Kp;
  const int Cin = a.C1 + a.C2;
  float acc = 0.f;
  for (int t = t_lo; t < t_hi; ++t) {
    const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
    const int iy = gy * a.in_sy + (tb & 7) - 2, ix = gx * a.in_sx + (tb >> 3) - 2;
    // a tap outside the image multiplies ZEROS, it is not skipped: 0 * inf = NaN is what the zero padding of the fp32 kernels (and
    // of F.conv2d) produces under a non-finite weight
    const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    const size_t pix = inside ? (size_t)(n * a.H + iy) * a.W + ix : 0;
    for (int c = 0; c < Cin; c += 4) {
      f32x4 x = c < a.C1 ? *(const f32x4*)(a.in1 + pix * a.ld1 + c) : *(const f32x4*)(a.in2 + pix * a.ld2 + (c - a.C1));
      if constexpr (XF) {                                // fused batch-norm apply on the input: bn_apply_kernel's arithmetic
        x = x * *(const f32x4*)(a.in_scale + c) + *(const f32x4*)(a.in_shift + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = ssg_act(x[e], a.in_act, a.in_slope);
      }
      if (!inside) x = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 w = *(const f32x4*)(wrow + ((c >> 4) * a.ntaps + t) * 16 + (c & 15));
      acc = __builtin_fmaf(x[0], w[0], acc); acc = __builtin_fmaf(x[1], w[1], acc);
      acc = __builtin_fmaf(x[2], w[2], acc); acc = __builtin_fmaf(x[3], w[3], acc);
    }
  }
  return acc;
}

// Refill one 32x32 accumulator fragment (16 values per lane) on the slow path: `value(r)` computes element r; the values pass through
// LDS scratch (`scr` = this thread's column: element r at scr[r * nthreads]) so that the loop over r stays a loop -- indexing the
// accumulator registers dynamically would put them in scratch memory for the whole kernel.
template <class F>
__device__ __forceinline__ void ssg_slow_refill16(f32x16& acc, float* scr, int nthreads, F&& value) {
#pragma unroll 1
  for (int r = 0; r < 16; ++r) scr[r * nthreads] = value(r);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = scr[r * nthreads];
}

__device__ __forceinline__ bool ssg_nonfinite16(const f32x16& v) {
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 16; ++r) bad |= ssg_nonfinite(v[r]);
  return bad;
}

// the kernel's argument block, re-read from the kernarg segment at the point of use: its fields must not stay live in SGPRs across
// the hot loop for the sake of this cold path (that spilled registers of the k32 kernel)
template <class Args>
__device__ __forceinline__ const Args* ssg_reload_args() {
  const Args* ap = (const Args*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ap));
  return ap;
}

template <class F>
__device__ __forceinline__ void ssg_slow_refill4(f32x4& acc, float* scr, int nthreads, F&& value) {
#pragma unroll 1
  for (int r = 0; r < 4; ++r) scr[r * nthreads] = value(r);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = scr[r * nthreads];
}

// ---- weight gradients: one element dW[tap t][input channel c][output channel co] summed with fp32 FMAs over a workgroup's pixels
template <bool XF = false>
__device__ __forceinline__ float ssg_wgrad_slow_pixel(const WgArgs& a, int dyt, int dxt, int c, int co, int n, int gy, int gx, float acc) {
  const int iy = gy * a.in_sy + dyt, ix = gx * a.in_sx + dxt;
  const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;      // outside: a ZERO times dy (0 * inf = NaN, as the zero page of the fp32 kernels)
  const size_t pix = inside ? (size_t)(n * a.H + iy) * a.W + ix : 0;
  float x = c < a.C1 ? a.in1[pix * a.ld1 + c] : a.in2[pix * a.ld2 + (c - a.C1)];
  if constexpr (XF) x = ssg_act(x * a.in_scale[c] + a.in_shift[c], a.in_act, a.in_slope);   // ssg_wgrad_desc.in_scale
  if (!inside) x = 0.f;
  return __builtin_fmaf(x, a.dout[((size_t)(n * a.GH + gy) * a.GW + gx) * a.ldd + co], acc);
}

// pixels P0 <= P < P1 of the flat N x GH x GW grid (wgrad_dma_x3_kernel's slabs)
__device__ inline float ssg_wgrad_slow_value_flat(const WgArgs& a, int t, int c, int co, long long P0, long long P1) {
  if (co >= a.Cout || c >= a.C1 + a.C2 || t >= a.ntaps) return 0.f;
  const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
  const int dyt = (tb & 7) - 2, dxt = (tb >> 3) - 2;
  const long long GHW = (long long)a.GH * a.GW;
  int n = (int)(P0 / GHW); const int rem = (int)(P0 - n * GHW);
  int gy = rem / a.GW, gx = rem - gy * a.GW;
  float acc = 0.f;
  for (long long P = P0; P < P1; ++P) {
    acc = ssg_wgrad_slow_pixel(a, dyt, dxt, c, co, n, gy, gx, acc);
    if (++gx == a.GW) { gx = 0; if (++gy == a.GH) { gy = 0; ++n; } }
  }
  return acc;
}

// K-steps S0 <= S < S1 in column-strip order: S = (n * XB + strip) * GH + gy, a step = KPX pixels of row gy from column strip * KPX
// (wgrad_halo_x3_kernel: KPX = 16; wgrad_k32_kernel: KPX = 32)
template <bool XF = false>
__device__ inline float ssg_wgrad_slow_value_strips(const WgArgs& a, int t, int c, int co, long long S0, long long S1, int KPX) {
  if (co >= a.Cout || c >= a.C1 + a.C2 || t >= a.ntaps) return 0.f;
  const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
  const int dyt = (tb & 7) - 2, dxt = (tb >> 3) - 2;
  const int XB = (a.GW + KPX - 1) / KPX;
  float acc = 0.f;
  for (long long S = S0; S < S1; ++S) {
    const int gy = (int)(S % a.GH); const long long col = S / a.GH;
    const int xb = (int)(col % XB), n = (int)(col / XB);
    const int gx1 = (xb + 1) * KPX < a.GW ? (xb + 1) * KPX : a.GW;
    for (int gx = xb * KPX; gx < gx1; ++gx) acc = ssg_wgrad_slow_pixel<XF>(a, dyt, dxt, c, co, n, gy, gx, acc);
  }
  return acc;
}
