// Element-wise kernels for gfx950 (HBM-bound, 16 B per lane): SPADE modulate fwd/bwd,
// activation backward, add, NaN masking, clamp, fused clamp+Adam.
// ABI + reference citations: include/ssunet_hip.h.
#include "common.h"

namespace {

int elem_grid(long long total) { return ssg_elem_grid(total, 2); }
#define GRID_STRIDE(i, total) SSG_CHUNK_LOOP(i, total)      // contiguous chunk per block (common.h)

__global__ __launch_bounds__(256) void modulate_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gb, int ldgb,
                                                           long long P, int C, float* __restrict__ y, int ldy) {
  const int CQ = C / 4;
  GRID_STRIDE(i, P * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    const f32x4 xv = *(const f32x4*)(x + p * ldx + 4 * cq);
    const f32x4 g = *(const f32x4*)(gb + p * ldgb + 4 * cq), b = *(const f32x4*)(gb + p * ldgb + C + 4 * cq);
    st4(y + p * ldy + 4 * cq, xv * (1.f + g) + b);
  }
}
__global__ __launch_bounds__(256) void modulate_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gb, int ldgb,
                                                           const float* __restrict__ dy, int lddy, long long P, int C,
                                                           float* __restrict__ dx, int lddx, float* __restrict__ dgb, int lddgb) {
  const int CQ = C / 4;
  GRID_STRIDE(i, P * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    const f32x4 xv = *(const f32x4*)(x + p * ldx + 4 * cq);
    const f32x4 g = *(const f32x4*)(gb + p * ldgb + 4 * cq);
    const f32x4 d = *(const f32x4*)(dy + p * lddy + 4 * cq);
    st4(dx + p * lddx + 4 * cq, d * (1.f + g));
    st4(dgb + p * lddgb + 4 * cq, d * xv);
    st4(dgb + p * lddgb + C + 4 * cq, d);
  }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ dy, int lddy,
                                                      long long P, int C, int act, float slope, float* __restrict__ dx, int lddx) {
  const int CQ = C / 4;
  GRID_STRIDE(i, P * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    const f32x4 yv = *(const f32x4*)(y + p * ldy + 4 * cq);
    f32x4 d = *(const f32x4*)(dy + p * lddy + 4 * cq);
#pragma unroll
    for (int e = 0; e < 4; ++e) if (!(yv[e] > 0.f)) d[e] *= (act == SSG_ACT_RELU ? 0.f : (act == SSG_ACT_LRELU ? slope : 1.f));
    st4(dx + p * lddx + 4 * cq, d);
  }
}
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float* __restrict__ o) {
  const long long nq = n / 4;
  GRID_STRIDE(i, nq) { *(f32x4*)(o + 4 * i) = *(const f32x4*)(a + 4 * i) + *(const f32x4*)(b + 4 * i); }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const long long j = nq * 4 + threadIdx.x; o[j] = a[j] + b[j]; }
}
__global__ __launch_bounds__(256) void nan_to_zero_kernel(float* __restrict__ x, long long n, uint8_t* __restrict__ mask) {
  GRID_STRIDE(i, n) { const float v = x[i]; const bool isn = v != v; if (isn) x[i] = 0.f; if (mask) mask[i] = isn ? 1 : 0; }
}
__global__ __launch_bounds__(256) void mask_zero_kernel(const float* g, const uint8_t* __restrict__ mask, long long n, float* out) {
  GRID_STRIDE(i, n) { out[i] = mask[i] ? 0.f : g[i]; }
}
__global__ __launch_bounds__(256) void clamp_kernel(float* __restrict__ x, long long n, float lo, float hi) {
  GRID_STRIDE(i, n) { float v = x[i]; v = v < lo ? lo : v; v = v > hi ? hi : v; x[i] = v; }   // NaN stays NaN (torch.clamp)
}

// ---------------------------------------------------------------- fused clamp + Adam, multi-tensor
// One workgroup per (tensor, 4096-element chunk); arithmetic order follows torch.optim.Adam's
// single-tensor path: lerp exp_avg, mul/addcmul exp_avg_sq, denom = sqrt(v)/bc2_sqrt + eps,
// param += -(lr/bc1) * m/denom.
constexpr int ADAM_CHUNK = 4096;

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float clip, float lr_bc1, float omb1, float b2,
                                          float omb2, float eps, float wd, float bc2s) {
  if (clip > 0.f) { g = g < -clip ? -clip : g; g = g > clip ? clip : g; }
  if (wd != 0.f) g = g + wd * p;
  m = m + (g - m) * omb1;
  v = v * b2;
  v = v + (omb2 * g) * g;
  const float denom = sqrtf(v) / bc2s + eps;
  p = p - lr_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void clamp_adam_kernel(const void* const* __restrict__ ptrs, const long long* __restrict__ sizes,
                                                         const int* __restrict__ blk_tensor, const int* __restrict__ blk_chunk,
                                                         float clip, float lr_bc1, float b1, float b2, float omb2, float eps, float wd, float bc2s) {
  const int t = blk_tensor[blockIdx.x];
  const long long base = (long long)blk_chunk[blockIdx.x] * ADAM_CHUNK;
  float* P = (float*)ptrs[4 * t + 0];
  float* G = (float*)ptrs[4 * t + 1];
  float* M = (float*)ptrs[4 * t + 2];
  float* V = (float*)ptrs[4 * t + 3];
  long long n = sizes[t] - base;
  if (n > ADAM_CHUNK) n = ADAM_CHUNK;
  const bool vec = ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0);
  if (vec) {
    const int nq = (int)(n / 4);
    for (int i = threadIdx.x; i < nq; i += 256) {
      const long long o = base + 4 * i;
      f32x4 p = *(f32x4*)(P + o), g = *(f32x4*)(G + o), m = *(f32x4*)(M + o), v = *(f32x4*)(V + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) { float pe = p[e], me = m[e], ve = v[e]; adam_elem(pe, g[e], me, ve, clip, lr_bc1, b1, b2, omb2, eps, wd, bc2s); p[e] = pe; m[e] = me; v[e] = ve; }
      *(f32x4*)(P + o) = p; *(f32x4*)(M + o) = m; *(f32x4*)(V + o) = v;
      if (clip > 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { float ge = g[e]; ge = ge < -clip ? -clip : ge; ge = ge > clip ? clip : ge; g[e] = ge; }
        *(f32x4*)(G + o) = g;                    // clip_gradient clamps .grad in place
      }
    }
    for (int i = nq * 4 + threadIdx.x; i < n; i += 256) {
      const long long o = base + i;
      float g = G[o];
      adam_elem(P[o], g, M[o], V[o], clip, lr_bc1, b1, b2, omb2, eps, wd, bc2s);
      if (clip > 0.f) { g = g < -clip ? -clip : g; g = g > clip ? clip : g; G[o] = g; }
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) {
      const long long o = base + i;
      float g = G[o];
      adam_elem(P[o], g, M[o], V[o], clip, lr_bc1, b1, b2, omb2, eps, wd, bc2s);
      if (clip > 0.f) { g = g < -clip ? -clip : g; g = g > clip ? clip : g; G[o] = g; }
    }
  }
}

__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, int lds_, long long P, int C,
                                                            float* __restrict__ dst, int ldd) {
  const int CQ = C / 4;
  GRID_STRIDE(i, P * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    *(f32x4*)(dst + p * ldd + 4 * cq) = *(const f32x4*)(src + p * lds_ + 4 * cq);
  }
}

// dst[p][c] = (TD) (a[p][c] (+ b[p][c])): dtype conversion between the fp32 and bf16 tensor families, and the bf16 residual add
template <typename TA, typename TD>
__global__ __launch_bounds__(256) void rows_convert_add_kernel(const TA* __restrict__ a, int lda, const TA* __restrict__ b, int ldb, long long P, int C,
                                                               TD* __restrict__ dst, int ldd) {
  const int CQ = C / 4;
  GRID_STRIDE(i, P * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    f32x4 v = ld4(a + p * lda + 4 * cq);
    if (b) v += ld4(b + p * ldb + 4 * cq);
    st4(dst + p * ldd + 4 * cq, v);
  }
}

__global__ __launch_bounds__(256) void clamp_multi_kernel(const void* const* __restrict__ ptrs, const long long* __restrict__ sizes,
                                                          const int* __restrict__ blk_tensor, const int* __restrict__ blk_chunk, int stride,
                                                          float lo, float hi) {
  const int t = blk_tensor[blockIdx.x];
  const long long base = (long long)blk_chunk[blockIdx.x] * ADAM_CHUNK;
  float* P = (float*)ptrs[stride * t];
  long long n = sizes[t] - base;
  if (n > ADAM_CHUNK) n = ADAM_CHUNK;
  for (int i = threadIdx.x; i < n; i += 256) { float v = P[base + i]; v = v < lo ? lo : v; v = v > hi ? hi : v; P[base + i] = v; }
}


// Attention gate (archs.py:138-144): y[p][c] = x[p][c] * sigmoid(g[p]), one gate value per pixel.
__global__ __launch_bounds__(256) void pixel_gate_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg,
                                                             long long P, int C, float* __restrict__ y, int ldy) {
  const int CQ = C / 4;
  GRID_STRIDE(i, P * CQ) {
    const long long p = i / CQ; const int cq = (int)(i - p * CQ);
    const float s = 1.f / (1.f + __expf(-g[p * ldg]));
    *(f32x4*)(y + p * ldy + 4 * cq) = *(const f32x4*)(x + p * ldx + 4 * cq) * s;
  }
}
// dx = dy * s;  dg[p] = (sum_c dy[p][c] * x[p][c]) * s * (1 - s).  16 lanes per pixel, fixed shuffle tree.
__global__ __launch_bounds__(256) void pixel_gate_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg,
                                                             const float* __restrict__ dy, int lddy, long long P, int C,
                                                             float* __restrict__ dx, int lddx, float* __restrict__ dg, int lddg) {
  const int CQ = C / 4;
  const int sub = threadIdx.x & 15;
  const long long groups = ((long long)gridDim.x * 256) >> 4;
  const long long iters = (P + groups - 1) / groups;
  for (long long it = 0; it < iters; ++it) {
    const long long p = (((long long)blockIdx.x * 256 + threadIdx.x) >> 4) + it * groups;
    const bool ok = p < P;
    float s = 0.f, acc = 0.f;
    if (ok) {
      s = 1.f / (1.f + __expf(-g[p * ldg]));
      for (int cq = sub; cq < CQ; cq += 16) {
        const f32x4 xv = *(const f32x4*)(x + p * ldx + 4 * cq), dv = *(const f32x4*)(dy + p * lddy + 4 * cq);
        acc += (dv[0] * xv[0] + dv[1] * xv[1]) + (dv[2] * xv[2] + dv[3] * xv[3]);
        *(f32x4*)(dx + p * lddx + 4 * cq) = dv * s;
      }
    }
    acc += __shfl_xor(acc, 8); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 1);
    if (ok && sub == 0) *(f32x4*)(dg + p * lddg) = f32x4{acc * s * (1.f - s), 0.f, 0.f, 0.f};
  }
}
}  // namespace

extern "C" int ssg_spade_modulate_fwd_f32(const float* x, int ldx, const float* gb, int ldgb, int64_t P, int C, float* y, int ldy, void* stream) {
  SSG_REQUIRE(x && gb && y && P > 0 && C > 0 && C % 4 == 0 && ldgb >= 2 * C, SSG_EINVAL, "modulate: bad args");
  hipLaunchKernelGGL(modulate_fwd_kernel, dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, gb, ldgb, (long long)P, C, y, ldy);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_spade_modulate_bwd_f32(const float* x, int ldx, const float* gb, int ldgb, const float* dy, int lddy, int64_t P, int C,
                                          float* dx, int lddx, float* dgb, int lddgb, void* stream) {
  SSG_REQUIRE(x && gb && dy && dx && dgb && P > 0 && C > 0 && C % 4 == 0 && ldgb >= 2 * C && lddgb >= 2 * C, SSG_EINVAL, "modulate_bwd: bad args");
  hipLaunchKernelGGL(modulate_bwd_kernel, dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, gb, ldgb, dy, lddy, (long long)P, C, dx, lddx, dgb, lddgb);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_act_bwd_f32(const float* y, int ldy, const float* dy, int lddy, int64_t P, int C, int act, float slope, float* dx, int lddx, void* stream) {
  SSG_REQUIRE(y && dy && dx && P > 0 && C > 0 && C % 4 == 0, SSG_EINVAL, "act_bwd: bad args");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, y, ldy, dy, lddy, (long long)P, C, act, slope, dx, lddx);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_add_f32(const float* a, const float* b, int64_t n, float* out, void* stream) {
  SSG_REQUIRE(a && b && out && n > 0, SSG_EINVAL, "add: bad args");
  SSG_REQUIRE(ssg_aligned16(a) && ssg_aligned16(b) && ssg_aligned16(out), SSG_EALIGN, "add: alignment");
  hipLaunchKernelGGL(add_kernel, dim3(elem_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, b, (long long)n, out);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_nan_to_zero_f32(float* x, int64_t n, uint8_t* mask, void* stream) {
  SSG_REQUIRE(x && n > 0, SSG_EINVAL, "nan_to_zero: bad args");
  hipLaunchKernelGGL(nan_to_zero_kernel, dim3(elem_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, mask);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_mask_zero_f32(const float* g, const uint8_t* mask, int64_t n, float* out, void* stream) {
  SSG_REQUIRE(g && mask && out && n > 0, SSG_EINVAL, "mask_zero: bad args");
  hipLaunchKernelGGL(mask_zero_kernel, dim3(elem_grid(n)), dim3(256), 0, (hipStream_t)stream, g, mask, (long long)n, out);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_clamp_f32(float* x, int64_t n, float lo, float hi, void* stream) {
  SSG_REQUIRE(x && n > 0, SSG_EINVAL, "clamp: bad args");
  hipLaunchKernelGGL(clamp_kernel, dim3(elem_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, lo, hi);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_clamp_adam_multi_f32(const void* const* ptrs, const int64_t* sizes, const int32_t* blk_tensor, const int32_t* blk_chunk,
                                        int nblocks, float clip, double lr, double beta1, double beta2, double eps, double weight_decay,
                                        double bias_corr1, double bias_corr2_sqrt, void* stream) {
  SSG_REQUIRE(ptrs && sizes && blk_tensor && blk_chunk && nblocks > 0, SSG_EINVAL, "adam: bad args");
  SSG_REQUIRE(bias_corr1 > 0.0 && bias_corr2_sqrt > 0.0, SSG_EINVAL, "adam: bias corrections");
  hipLaunchKernelGGL(clamp_adam_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, ptrs, (const long long*)sizes,
                     blk_tensor, blk_chunk, clip, (float)(lr / bias_corr1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                     (float)weight_decay, (float)bias_corr2_sqrt);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

/* Clamp tensor number `which` (0 = param, 1 = grad, ...) of every 4-pointer record in `ptrs` (the
 * layout ssg_clamp_adam_multi_f32 uses) to [lo, hi] in one launch: stage-1's per-step WEIGHT clamp
 * (train.py:111-112). */
extern "C" int ssg_clamp_multi_f32(const void* const* ptrs, const int64_t* sizes, const int32_t* blk_tensor, const int32_t* blk_chunk,
                                   int nblocks, int which, float lo, float hi, void* stream) {
  SSG_REQUIRE(ptrs && sizes && blk_tensor && blk_chunk && nblocks > 0 && which >= 0 && which < 4, SSG_EINVAL, "clamp_multi: bad args");
  hipLaunchKernelGGL(clamp_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const void* const*)(ptrs + which),
                     (const long long*)sizes, blk_tensor, blk_chunk, 4, lo, hi);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

/* dst[p, 0:C] = src[p, 0:C] for P pixels (channel-slice copy: materialises torch.cat of more than two
 * tensors, e.g. the dense skip connections of NestedUNet, archs.py:910-925). */
extern "C" int ssg_copy_channels_f32(const float* src, int ldsrc, int64_t P, int C, float* dst, int lddst, void* stream) {
  SSG_REQUIRE(src && dst && P > 0 && C > 0 && C % 4 == 0 && ldsrc % 4 == 0 && lddst % 4 == 0, SSG_EINVAL, "copy_channels: bad args");
  hipLaunchKernelGGL(copy_channels_kernel, dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, src, ldsrc, (long long)P, C, dst, lddst);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_pixel_gate_fwd_f32(const float* x, int ldx, const float* g, int ldg, int64_t P, int C, float* y, int ldy, void* stream) {
  SSG_REQUIRE(x && g && y && P > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, SSG_EINVAL, "pixel_gate_fwd: bad args");
  hipLaunchKernelGGL(pixel_gate_fwd_kernel, dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, g, ldg, (long long)P, C, y, ldy);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_pixel_gate_bwd_f32(const float* x, int ldx, const float* g, int ldg, const float* dy, int lddy, int64_t P, int C,
                                      float* dx, int lddx, float* dg, int lddg, void* stream) {
  SSG_REQUIRE(x && g && dy && dx && dg && P > 0 && C > 0 && C % 4 == 0, SSG_EINVAL, "pixel_gate_bwd: bad args");
  SSG_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && lddg % 4 == 0 && ssg_aligned16(dg), SSG_EALIGN, "pixel_gate_bwd: strides");
  hipLaunchKernelGGL(pixel_gate_bwd_kernel, dim3(elem_grid(P * 16)), dim3(256), 0, (hipStream_t)stream, x, ldx, g, ldg, dy, lddy, (long long)P, C,
                     dx, lddx, dg, lddg);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

/* dtype conversion between the fp32 and the bf16 tensor family (rows of C channels, C % 4 == 0), and the bf16 residual add */
extern "C" int ssg_convert_f32_to_bf16(const float* src, int ldsrc, int64_t P, int C, void* dst, int lddst, void* stream) {
  SSG_REQUIRE(src && dst && P > 0 && C > 0 && C % 4 == 0 && ldsrc % 4 == 0 && lddst % 4 == 0, SSG_EINVAL, "convert_f32_to_bf16: bad args");
  hipLaunchKernelGGL((rows_convert_add_kernel<float, ssg_bf16>), dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, src, ldsrc,
                     (const float*)nullptr, 0, (long long)P, C, (ssg_bf16*)dst, lddst);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_convert_bf16_to_f32(const void* src, int ldsrc, int64_t P, int C, float* dst, int lddst, void* stream) {
  SSG_REQUIRE(src && dst && P > 0 && C > 0 && C % 4 == 0 && ldsrc % 4 == 0 && lddst % 4 == 0, SSG_EINVAL, "convert_bf16_to_f32: bad args");
  hipLaunchKernelGGL((rows_convert_add_kernel<ssg_bf16, float>), dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const ssg_bf16*)src,
                     ldsrc, (const ssg_bf16*)nullptr, 0, (long long)P, C, dst, lddst);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_add_bf16(const void* a, int lda, const void* b, int ldb, int64_t P, int C, void* out, int ldo, void* stream) {
  SSG_REQUIRE(a && b && out && P > 0 && C > 0 && C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0, SSG_EINVAL, "add_bf16: bad args");
  hipLaunchKernelGGL((rows_convert_add_kernel<ssg_bf16, ssg_bf16>), dim3(elem_grid(P * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const ssg_bf16*)a,
                     lda, (const ssg_bf16*)b, ldb, (long long)P, C, (ssg_bf16*)out, ldo);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
