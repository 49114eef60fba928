// Skinny fully-connected forward: y[n][o] = act(sum_k x[n][k] * w[o][k] + bias[o]) for a handful of rows n
// (the discriminator's fc1: 16 x 18432 -> 1024, models_seg_gan.py:281-283).  The weight matrix (75 MB) is
// the only large operand and is read exactly once, straight from the parameter's [O][K] layout: HBM-bound.
// As an implicit-GEMM conv this shape is 8 workgroups walking K = 18432 serially (1.4 ms); here
//   * a wave owns 4 weight rows x one K slice: per 256-wide K step 4 coalesced 1-KiB weight loads and the
//     x rows (L2 resident), 4 x 16 x 4 FMAs per lane;
//   * the 64 per-lane accumulators are folded across the wave by a data-halving butterfly (63 shuffles,
//     lane L ends up with output (row L/16, sample L%16));
//   * K slices are summed in slice order by a second tiny kernel that adds bias and activation
//     (deterministic, no atomics).
#include "common.h"

namespace {

constexpr int RO = 4;      // weight rows per wave
constexpr int NB = 16;     // samples per pass

__global__ __launch_bounds__(256) void linear_partial_kernel(const float* __restrict__ x, int n, int k, int ldx,
                                                             const float* __restrict__ w, int o, int ob_count, int kchunk,
                                                             int n0, float* __restrict__ part /* [ks][NB][o] */) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ob = wid % ob_count, ks = wid / ob_count;
  const int k0 = ks * kchunk;
  if (k0 >= k) return;
  const int k1 = k0 + kchunk < k ? k0 + kchunk : k;
  float acc[RO * NB];
#pragma unroll
  for (int i = 0; i < RO * NB; ++i) acc[i] = 0.f;
  const float* wrow[RO]; bool rok[RO];
#pragma unroll
  for (int r = 0; r < RO; ++r) { rok[r] = ob * RO + r < o; wrow[r] = w + (size_t)(rok[r] ? ob * RO + r : 0) * k; }
  for (int kb = k0; kb < k1; kb += 256) {
    const int kk = kb + lane * 4;
    const bool kok = kk < k1;
    f32x4 wv[RO];
#pragma unroll
    for (int r = 0; r < RO; ++r) wv[r] = (kok && rok[r]) ? *(const f32x4*)(wrow[r] + kk) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const f32x4 xv = (kok && n0 + s < n) ? *(const f32x4*)(x + (size_t)(n0 + s) * ldx + kk) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < RO; ++r) {
        float a = acc[r * NB + s];
        a = fmaf(wv[r][0], xv[0], a); a = fmaf(wv[r][1], xv[1], a);
        a = fmaf(wv[r][2], xv[2], a); a = fmaf(wv[r][3], xv[3], a);
        acc[r * NB + s] = a;
      }
    }
  }
  // butterfly with data halving: after the step of width W a lane keeps the half selected by its own bit
#pragma unroll
  for (int width = 32; width >= 1; width >>= 1) {
    const bool hi = (lane & width) != 0;
#pragma unroll
    for (int i = 0; i < width; ++i) {
      const float keep = hi ? acc[i + width] : acc[i];
      const float send = hi ? acc[i] : acc[i + width];
      acc[i] = keep + __shfl_xor(send, width);
    }
  }
  const int r = lane / NB, s = lane % NB;
  if (ob * RO + r < o) part[((size_t)ks * NB + s) * o + ob * RO + r] = acc[0];
}

__global__ void linear_finish_kernel(const float* __restrict__ part, int ksplit, int n, int o, int n0, const float* __restrict__ bias,
                                     int act, float slope, float* __restrict__ y, int ldy) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NB * o) return;
  const int s = idx / o, c = idx - s * o;
  if (n0 + s >= n) return;
  float v = 0.f;
  for (int ks = 0; ks < ksplit; ++ks) v += part[((size_t)ks * NB + s) * o + c];
  if (bias) v += bias[c];
  if (act == SSG_ACT_RELU) v = v < 0.f ? 0.f : v;
  else if (act == SSG_ACT_LRELU) v = v > 0.f ? v : v * slope;
  y[(size_t)(n0 + s) * ldy + c] = v;
}

// Weight gradient of a linear layer, dw[o][k] = sum_n dy[n][o] * x[n][k] (models_seg_gan.py:281-283: fc1 is 18 432 -> 1024,
// a 75-MB gradient from a batch of 16).  Pure output streaming: one thread owns 4 consecutive k, keeps x[n][k..k+3] for a
// 16-sample slab in registers, and walks a tile of WO output rows (dy[n][o] is wave-uniform: scalar loads).  Fixed summation
// order over n.  As a 1x1 conv on a 1 x n grid this shape ran at 2.4 TFLOP/s (0.25 ms); it is write-bound at ~20 us.
constexpr int WO = 32;                     // output rows per workgroup
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int ldd,
                                                           int n, int k, int o, float* __restrict__ dw) {
  const int k0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int o0 = blockIdx.y * WO;
  if (k0 >= k) return;
  f32x4 acc[WO];
#pragma unroll
  for (int r = 0; r < WO; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int n0 = 0; n0 < n; n0 += 16) {
    f32x4 xv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xv[i] = n0 + i < n ? *(const f32x4*)(x + (size_t)(n0 + i) * ldx + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < WO; ++r) {
      if (o0 + r < o) {                    // wave-uniform
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float g = n0 + i < n ? dy[(size_t)(n0 + i) * ldd + o0 + r] : 0.f;
          acc[r] += xv[i] * g;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < WO; ++r)
    if (o0 + r < o) *(f32x4*)(dw + (size_t)(o0 + r) * k + k0) = acc[r];
}

void plan(int k, int o, int* ob_count, int* ksplit, int* kchunk) {
  *ob_count = (o + RO - 1) / RO;
  int ks = 2048 / *ob_count;                      // ~8 waves per CU
  if (ks < 1) ks = 1;
  int kc = ((k + ks - 1) / ks + 255) / 256 * 256; // K slice, a multiple of the 256-wide step
  if (kc < 1024) kc = 1024;
  *kchunk = kc;
  *ksplit = (k + kc - 1) / kc;
}

}  // namespace

extern "C" int64_t ssg_linear_fwd_workspace_bytes(int n, int k, int o) {
  int obc, ks, kc;
  plan(k, o, &obc, &ks, &kc);
  return (int64_t)ks * NB * o * (int64_t)sizeof(float);
}

extern "C" int ssg_linear_fwd_f32(const float* x, int n, int k, int ldx, const float* w, int o, const float* bias, int act,
                                  float slope, float* y, int ldy, float* ws, int64_t ws_bytes, void* stream) {
  SSG_REQUIRE(x && w && y && ws && n > 0 && k > 0 && o > 0, SSG_EINVAL, "linear: bad args");
  SSG_REQUIRE(k % 4 == 0 && ldx % 4 == 0 && ldx >= k && ssg_aligned16(x) && ssg_aligned16(w), SSG_EALIGN, "linear: K and row strides must be multiples of 4 floats");
  SSG_REQUIRE(ldy >= o, SSG_EINVAL, "linear: ldy < O");
  SSG_REQUIRE(ws_bytes >= ssg_linear_fwd_workspace_bytes(n, k, o), SSG_EINVAL, "linear: workspace too small");
  int obc, ks, kc;
  plan(k, o, &obc, &ks, &kc);
  hipStream_t st = (hipStream_t)stream;
  for (int n0 = 0; n0 < n; n0 += NB) {
    const int waves = obc * ks;
    hipLaunchKernelGGL(linear_partial_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, x, n, k, ldx, w, o, obc, kc, n0, ws);
    SSG_LAUNCH_CHECK();
    hipLaunchKernelGGL(linear_finish_kernel, dim3((unsigned)((NB * o + 255) / 256)), dim3(256), 0, st, ws, ks, n, o, n0, bias, act, slope, y, ldy);
    SSG_LAUNCH_CHECK();
  }
  return SSG_OK;
}

extern "C" int ssg_linear_wgrad_f32(const float* x, int n, int k, int ldx, const float* dy, int o, int ldd, float* dw, void* stream) {
  SSG_REQUIRE(x && dy && dw && n > 0 && k > 0 && o > 0, SSG_EINVAL, "linear wgrad: bad args");
  SSG_REQUIRE(k % 4 == 0 && ldx % 4 == 0 && ldx >= k && ldd >= o && ssg_aligned16(x) && ssg_aligned16(dw), SSG_EALIGN,
              "linear wgrad: K and the x row stride must be multiples of 4 floats, x / dw 16-byte aligned");
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3((unsigned)((k / 4 + 255) / 256), (unsigned)((o + WO - 1) / WO)), dim3(256), 0, (hipStream_t)stream,
                     x, ldx, dy, ldd, n, k, o, dw);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
