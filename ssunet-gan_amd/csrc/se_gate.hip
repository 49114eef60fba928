// Squeeze-excite gate of the MBConv block (efficientnet_pytorch/model.py:83-87):
//   gate[n][c] = sigmoid(b2[c] + sum_s w2[c][s] * swish(b1[s] + sum_k w1[s][k] * sq[n][k]))
// for the pooled [N][C] vector.  The reference runs it as two 1x1 convs on a 1x1 image plus two activation modules; as conv
// launches that is 6 kernels forward and 12 backward per block (weight packs, split-K slabs and their reducers for matrices of a
// few hundred KB), 32 blocks per EfficientNet-B4 step -- launch time, not work.  Here: 2 kernels forward, 3 backward, every sum in
// a fixed order (deterministic), fp32 throughout.  N <= 16 samples, S <= 256 squeezed channels, N*S <= 4096.
#include "common.h"

namespace {

constexpr int SE_MAXN = 16;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// h_pre[n][s] = b1[s] + sum_c w1[s][c] * sq[n][c]: one wave per squeezed channel s, lanes over c
__global__ __launch_bounds__(256) void se_hidden_kernel(const float* __restrict__ sq, int ldq, int N, int C, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, int S, float* __restrict__ h_pre) {
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= S) return;
  float acc[SE_MAXN];
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n) acc[n] = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float wv = w1[(size_t)s * C + c];
#pragma unroll
    for (int n = 0; n < SE_MAXN; ++n)
      if (n < N) acc[n] = fmaf(wv, sq[(size_t)n * ldq + c], acc[n]);
  }
  const float bv = b1 ? b1[s] : 0.f;
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n)
    if (n < N) {
      const float t = wave_sum(acc[n]);
      if (lane == 0) h_pre[n * S + s] = t + bv;
    }
}

// gate[n][c] = sigmoid(b2[c] + sum_s w2[c][s] * swish(h_pre[n][s])): one thread per channel c, swish(h) staged in LDS
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ h_pre, int N, int S, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, int C, float* __restrict__ gate, int ldg) {
  extern __shared__ float hs[];                          // [N][S]
  for (int i = threadIdx.x; i < N * S; i += 256) { const float z = h_pre[i]; hs[i] = z * ssg_sigmoid_fast(z); }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float acc[SE_MAXN];
  const float bv = b2 ? b2[c] : 0.f;
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n) acc[n] = bv;
  const float* wr = w2 + (size_t)c * S;
  for (int s = 0; s < S; ++s) {
    const float wv = wr[s];
#pragma unroll
    for (int n = 0; n < SE_MAXN; ++n)
      if (n < N) acc[n] = fmaf(wv, hs[n * S + s], acc[n]);
  }
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n)
    if (n < N) gate[(size_t)n * ldg + c] = ssg_sigmoid_fast(acc[n]);
}

// per channel c: dz[n][c] = dgate * g * (1 - g); db2[c] = sum_n dz; dw2[c][s] = sum_n dz[n][c] * swish(h_pre[n][s])
__global__ __launch_bounds__(256) void se_bwd_out_kernel(const float* __restrict__ dgate, int ldd, const float* __restrict__ gate, int ldg,
                                                         const float* __restrict__ h_pre, int N, int S, int C, float* __restrict__ dz,
                                                         float* __restrict__ dw2, float* __restrict__ db2) {
  extern __shared__ float hs[];
  for (int i = threadIdx.x; i < N * S; i += 256) { const float z = h_pre[i]; hs[i] = z * ssg_sigmoid_fast(z); }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float d[SE_MAXN];
  float sb = 0.f;
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n) {
    d[n] = 0.f;
    if (n < N) {
      const float g = gate[(size_t)n * ldg + c];
      d[n] = dgate[(size_t)n * ldd + c] * (g * (1.f - g));
      dz[(size_t)n * C + c] = d[n];
      sb += d[n];
    }
  }
  if (db2) db2[c] = sb;
  float* wr = dw2 + (size_t)c * S;
  for (int s = 0; s < S; ++s) {
    float t = 0.f;
#pragma unroll
    for (int n = 0; n < SE_MAXN; ++n)
      if (n < N) t = fmaf(d[n], hs[n * S + s], t);
    wr[s] = t;
  }
}

// per squeezed channel s (one wave): dh[n] = sum_c w2[c][s] * dz[n][c]; dhp[n][s] = dh[n] * swish'(h_pre[n][s]);
// db1[s] = sum_n dhp; dw1[s][c] = sum_n dhp[n][s] * sq[n][c]
__global__ __launch_bounds__(256) void se_bwd_hidden_kernel(const float* __restrict__ dz, const float* __restrict__ w2, const float* __restrict__ h_pre,
                                                            const float* __restrict__ sq, int ldq, int N, int S, int C,
                                                            float* __restrict__ dhp, float* __restrict__ dw1, float* __restrict__ db1) {
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= S) return;
  float acc[SE_MAXN];
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n) acc[n] = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float wv = w2[(size_t)c * S + s];
#pragma unroll
    for (int n = 0; n < SE_MAXN; ++n)
      if (n < N) acc[n] = fmaf(wv, dz[(size_t)n * C + c], acc[n]);
  }
  float sb = 0.f;
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n)
    if (n < N) {
      acc[n] = wave_sum(acc[n]) * ssg_swish_grad(h_pre[n * S + s]);          // every lane holds the total
      if (lane == 0) dhp[n * S + s] = acc[n];
      sb += acc[n];
    }
  if (lane == 0 && db1) db1[s] = sb;
  for (int c = lane; c < C; c += 64) {
    float t = 0.f;
#pragma unroll
    for (int n = 0; n < SE_MAXN; ++n)
      if (n < N) t = fmaf(acc[n], sq[(size_t)n * ldq + c], t);
    dw1[(size_t)s * C + c] = t;
  }
}

// per channel c: dsq[n][c] = sum_s w1[s][c] * dhp[n][s]
__global__ __launch_bounds__(256) void se_bwd_in_kernel(const float* __restrict__ dhp, const float* __restrict__ w1, int N, int S, int C,
                                                        float* __restrict__ dsq, int lds_) {
  extern __shared__ float hs[];
  for (int i = threadIdx.x; i < N * S; i += 256) hs[i] = dhp[i];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float acc[SE_MAXN];
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n) acc[n] = 0.f;
  for (int s = 0; s < S; ++s) {
    const float wv = w1[(size_t)s * C + c];
#pragma unroll
    for (int n = 0; n < SE_MAXN; ++n)
      if (n < N) acc[n] = fmaf(wv, hs[n * S + s], acc[n]);
  }
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n)
    if (n < N) dsq[(size_t)n * lds_ + c] = acc[n];
}

bool se_shape_ok(int N, int C, int S) { return N >= 1 && N <= SE_MAXN && S >= 1 && S <= 256 && N * S <= 4096 && C >= 1; }

}  // namespace

extern "C" int ssg_se_gate_ok(int N, int C, int S) { return se_shape_ok(N, C, S) ? 1 : 0; }

extern "C" int ssg_se_gate_fwd_f32(const float* sq, int ldq, int N, int C, const float* w1, const float* b1, const float* w2, const float* b2,
                                   int S, float* h_pre, float* gate, int ldg, void* stream) {
  SSG_REQUIRE(sq && w1 && w2 && h_pre && gate, SSG_EINVAL, "se_gate: null pointer");
  SSG_REQUIRE(se_shape_ok(N, C, S) && ldq >= C && ldg >= C, SSG_EINVAL, "se_gate: N=%d C=%d S=%d outside the kernel's range (N <= 16, S <= 256, N*S <= 4096)", N, C, S);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(se_hidden_kernel, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, st, sq, ldq, N, C, w1, b1, S, h_pre);
  SSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(se_gate_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), (size_t)N * S * sizeof(float), st, h_pre, N, S, w2, b2, C, gate, ldg);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_se_gate_bwd_f32(const float* dgate, int ldd, const float* gate, int ldg, const float* h_pre, const float* sq, int ldq, int N, int C,
                                   const float* w1, const float* w2, int S, float* dsq, int lds_, float* dw1, float* db1, float* dw2, float* db2,
                                   float* tmp, void* stream) {
  SSG_REQUIRE(dgate && gate && h_pre && sq && w1 && w2 && dsq && dw1 && dw2 && tmp, SSG_EINVAL, "se_gate_bwd: null pointer");
  SSG_REQUIRE(se_shape_ok(N, C, S) && ldd >= C && ldg >= C && ldq >= C && lds_ >= C, SSG_EINVAL, "se_gate_bwd: N=%d C=%d S=%d outside the kernel's range", N, C, S);
  hipStream_t st = (hipStream_t)stream;
  float* dz = tmp;                                       // [N][C]
  float* dhp = tmp + (size_t)N * C;                      // [N][S]
  const size_t lds_bytes = (size_t)N * S * sizeof(float);
  hipLaunchKernelGGL(se_bwd_out_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), lds_bytes, st, dgate, ldd, gate, ldg, h_pre, N, S, C, dz, dw2, db2);
  SSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(se_bwd_hidden_kernel, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, st, (const float*)dz, w2, h_pre, sq, ldq, N, S, C, dhp, dw1, db1);
  SSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(se_bwd_in_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), lds_bytes, st, (const float*)dhp, w1, N, S, C, dsq, lds_);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
