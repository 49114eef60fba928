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

__device__ __forceinline__ bool ssg_nonfinite(float v) { return !(__builtin_fabsf(v) <= 3.4028234663852886e38f); }

// One pre-epilogue accumulator value of a convolution launch: sum over taps [t_lo, t_hi) and all input channels of
// in[n][gy*in_sy + dy_t][gx*in_sx + dx_t][c] * w32[co][k(t, c)], fp32 FMAs, out-of-image taps contribute nothing.
// w32 = the fp32 packed weights [Cout][Kp] in kmode 0 (k = (c / 16 * ntaps + t) * 16 + c % 16).
__device__ inline float ssg_conv_slow_value(const ConvArgs& a, int n, int gy, int gx, int co, int t_lo, int t_hi) {
  if (co >= a.Cout) return 0.f;
  const float* wrow = a.w32 + (size_t)co * a.Kp;
  const int Cin = a.C1 + a.C2;
  float acc = 0.f;
  for (int t = t_lo; t < t_hi; ++t) {
    const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
    const int iy = gy * a.in_sy + (tb & 7) - 2, ix = gx * a.in_sx + (tb >> 3) - 2;
    if ((unsigned)iy >= (unsigned)a.H || (unsigned)ix >= (unsigned)a.W) continue;
    const size_t pix = (size_t)(n * a.H + iy) * a.W + ix;
    for (int c = 0; c < Cin; c += 4) {
      const f32x4 x = c < a.C1 ? *(const f32x4*)(a.in1 + pix * a.ld1 + c) : *(const f32x4*)(a.in2 + pix * a.ld2 + (c - a.C1));
      const f32x4 w = *(const f32x4*)(wrow + ((c >> 4) * a.ntaps + t) * 16 + (c & 15));
      acc = __builtin_fmaf(x[0], w[0], acc); acc = __builtin_fmaf(x[1], w[1], acc);
      acc = __builtin_fmaf(x[2], w[2], acc); acc = __builtin_fmaf(x[3], w[3], acc);
    }
  }
  return acc;
}
