// Squeeze-excite gate of the MBConv block (efficientnet_pytorch/model.py:83-87):
//   gate[n][c] = sigmoid(b2[c] + sum_s w2[c][s] * swish(b1[s] + sum_k w1[s][k] * sq[n][k]))
// for the pooled [N][C] vector.  The reference runs it as two 1x1 convs on a 1x1 image plus two activation modules; as conv
// launches that is 6 kernels forward and 12 backward per block (weight packs, split-K slabs and their reducers for matrices of a
// few hundred KB), 32 blocks per EfficientNet-B4 step -- launch time, not work.  Here: 2 kernels forward, 3 backward, every sum in
// a fixed order (deterministic), fp32 throughout.  N <= 16 samples, S <= 256 squeezed channels, N*S <= 2048.
#include "common.h"

namespace {

constexpr int SE_MAXN = 16;

// The two contractions over the C channels (h = W1 sq, dL/dswish(h) = W2^T dz) are split into 64-channel chunks: one thread per
// (sample, squeezed channel) pair and chunk sums its 64 products, the consumer adds the <= C/64 partials in a fixed order.  (One
// wave per squeezed channel streaming all C channels: 22-43 us per launch on a 28-workgroup grid, latency-bound; per-channel
// threads with a wave butterfly per (n, s): 53-60 us, shuffle-bound.)
constexpr int SE_CK = 64;

// sum_k part[k][i] in a fixed order, eight loads in flight (a 16-wide guarded form was slower: 18 -> 21 us per consumer launch)
__device__ __forceinline__ float sum_parts(const float* __restrict__ part, int nck, int NS, int i) {
  float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= nck; k += 8)
#pragma unroll
    for (int u = 0; u < 8; ++u) z[u] += part[(size_t)(k + u) * NS + i];
  for (; k < nck; ++k) z[0] += part[(size_t)k * NS + i];
  return ((z[0] + z[1]) + (z[2] + z[3])) + ((z[4] + z[5]) + (z[6] + z[7]));
}

// part[k][n][s] = sum_{c in chunk k} w1[s][c] * sq[n][c]
__global__ __launch_bounds__(256) void se_hidden_part_kernel(const float* __restrict__ sq, int ldq, int N, int C, const float* __restrict__ w1,
                                                             int S, float* __restrict__ part) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // n * S + s
  if (i >= N * S) return;
  const int n = i / S, s_ = i - n * S;
  const int c0 = blockIdx.y * SE_CK, c1 = c0 + SE_CK < C ? c0 + SE_CK : C;
  const float* wr = w1 + (size_t)s_ * C;
  const float* qr = sq + (size_t)n * ldq;
  float acc = 0.f;
  if (c1 - c0 == SE_CK) {
#pragma unroll 16
    for (int c = 0; c < SE_CK; ++c) acc = fmaf(wr[c0 + c], qr[c0 + c], acc);
  } else {
    for (int c = c0; c < c1; ++c) acc = fmaf(wr[c], qr[c], acc);
  }
  part[(size_t)blockIdx.y * N * S + i] = acc;
}

// part[k][n][s] = sum_{c in chunk k} w2[c][s] * dz[n][c]      (threads of consecutive s read consecutive addresses)
__global__ __launch_bounds__(256) void se_dh_part_kernel(const float* __restrict__ dz, int N, int C, const float* __restrict__ w2, int S,
                                                         float* __restrict__ part) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * S) return;
  const int n = i / S, s_ = i - n * S;
  const int c0 = blockIdx.y * SE_CK, c1 = c0 + SE_CK < C ? c0 + SE_CK : C;
  const float* zr = dz + (size_t)n * C;
  float acc = 0.f;
  if (c1 - c0 == SE_CK) {
#pragma unroll 16
    for (int c = 0; c < SE_CK; ++c) acc = fmaf(w2[(size_t)(c0 + c) * S + s_], zr[c0 + c], acc);
  } else {
    for (int c = c0; c < c1; ++c) acc = fmaf(w2[(size_t)c * S + s_], zr[c], acc);
  }
  part[(size_t)blockIdx.y * N * S + i] = acc;
}

// h_pre[n][s] = b1[s] + sum_k part[k][n][s] (workgroup 0 stores it for the backward);
// gate[n][c] = sigmoid(b2[c] + sum_s w2[c][s] * swish(h_pre[n][s]))
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ b1, int N, int S,
                                                      const float* __restrict__ w2, const float* __restrict__ b2, int C,
                                                      float* __restrict__ h_pre, float* __restrict__ gate, int ldg) {
  extern __shared__ float hs[];                          // [N][S]
  for (int i = threadIdx.x; i < N * S; i += 256) {
    const float z = (b1 ? b1[i % S] : 0.f) + sum_parts(part, nblk, N * S, i);
    if (blockIdx.x == 0) h_pre[i] = z;
    hs[i] = z * ssg_sigmoid_fast(z);
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float acc[SE_MAXN];
  const float bv = b2 ? b2[c] : 0.f;
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n) acc[n] = bv;
  const float* wr = w2 + (size_t)c * S;
  if ((S & 3) == 0) {
    // rows are 16-byte aligned; 16 row quads (64 squeezed channels) are requested together, so a row of S = 112 costs two
    // round trips instead of one per quad pair (the loop was a chain of load latencies: 16.5 us per launch)
    for (int s0 = 0; s0 < S; s0 += 64) {
      f32x4 wv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) wv[q] = s0 + 4 * q < S ? *(const f32x4*)(wr + s0 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (s0 + 4 * q < S) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int n = 0; n < SE_MAXN; ++n)
              if (n < N) acc[n] = fmaf(wv[q][e], hs[n * S + s0 + 4 * q + e], acc[n]);
        }
    }
  } else {
#pragma unroll 4
    for (int s = 0; s < S; ++s) {
      const float wv = wr[s];
#pragma unroll
      for (int n = 0; n < SE_MAXN; ++n)
        if (n < N) acc[n] = fmaf(wv, hs[n * S + s], acc[n]);
    }
  }
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n)
    if (n < N) gate[(size_t)n * ldg + c] = ssg_sigmoid_fast(acc[n]);
}

// per channel c: dz[n][c] = dgate[n][c] * g (1 - g) (stored for se_dh_part_kernel); db2[c] = sum_n dz;
// dw2[c][s] = sum_n dz[n][c] * swish(h_pre[n][s])
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
  if ((S & 3) == 0) {
    for (int s = 0; s < S; s += 4) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int n = 0; n < SE_MAXN; ++n)
          if (n < N) t[e] = fmaf(d[n], hs[n * S + s + e], t[e]);
      *(f32x4*)(wr + s) = t;
    }
  } else {
    for (int s = 0; s < S; ++s) {
      float t = 0.f;
#pragma unroll
      for (int n = 0; n < SE_MAXN; ++n)
        if (n < N) t = fmaf(d[n], hs[n * S + s], t);
      wr[s] = t;
    }
  }
}

// dhp[n][s] = (sum_b part[b][n][s]) * swish'(h_pre[n][s]); db1[s] = sum_n dhp (workgroup 0);
// per channel c: dw1[s][c] = sum_n dhp[n][s] * sq[n][c];  dsq[n][c] = sum_s w1[s][c] * dhp[n][s]
__global__ __launch_bounds__(256) void se_bwd_in_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ h_pre,
                                                        const float* __restrict__ sq, int ldq, const float* __restrict__ w1, int N, int S, int C,
                                                        float* __restrict__ dsq, int lds_, float* __restrict__ dw1, float* __restrict__ db1) {
  extern __shared__ float hs[];                          // dhp [N][S]
  for (int i = threadIdx.x; i < N * S; i += 256) {
    hs[i] = sum_parts(part, nblk, N * S, i) * ssg_swish_grad(h_pre[i]);
  }
  __syncthreads();
  if (blockIdx.x == 0 && db1)
    for (int s = threadIdx.x; s < S; s += 256) {
      float t = 0.f;
      for (int n = 0; n < N; ++n) t += hs[n * S + s];
      db1[s] = t;
    }
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float q_[SE_MAXN], acc[SE_MAXN];
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n) { q_[n] = n < N ? sq[(size_t)n * ldq + c] : 0.f; acc[n] = 0.f; }
  for (int s0 = 0; s0 < S; s0 += 16) {                   // 16 (coalesced) weight loads in flight, then 16 stores
    float wv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) wv[q] = s0 + q < S ? w1[(size_t)(s0 + q) * C + c] : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (s0 + q < S) {
        float t = 0.f;
#pragma unroll
        for (int n = 0; n < SE_MAXN; ++n)
          if (n < N) { const float dh = hs[n * S + s0 + q]; t = fmaf(dh, q_[n], t); acc[n] = fmaf(wv[q], dh, acc[n]); }
        dw1[(size_t)(s0 + q) * C + c] = t;
      }
  }
#pragma unroll
  for (int n = 0; n < SE_MAXN; ++n)
    if (n < N) dsq[(size_t)n * lds_ + c] = acc[n];
}

bool se_shape_ok(int N, int C, int S) { return N >= 1 && N <= SE_MAXN && S >= 1 && S <= 256 && N * S <= 2048 && C >= 1; }

}  // namespace

extern "C" int ssg_se_gate_ok(int N, int C, int S) { return se_shape_ok(N, C, S) ? 1 : 0; }

// floats of scratch for either direction: one [N][S] partial per 64 channels, plus dz [N][C] in the backward
extern "C" int64_t ssg_se_gate_workspace_floats(int N, int C, int S) { return (int64_t)((C + SE_CK - 1) / SE_CK) * N * S + (int64_t)N * C; }

extern "C" int ssg_se_gate_fwd_f32(const float* sq, int ldq, int N, int C, const float* w1, const float* b1, const float* w2, const float* b2,
                                   int S, float* h_pre, float* gate, int ldg, float* tmp, void* stream) {
  SSG_REQUIRE(sq && w1 && w2 && h_pre && gate && tmp, SSG_EINVAL, "se_gate: null pointer");
  SSG_REQUIRE(se_shape_ok(N, C, S) && ldq >= C && ldg >= C, SSG_EINVAL, "se_gate: N=%d C=%d S=%d outside the kernel's range (N <= 16, S <= 256, N*S <= 2048)", N, C, S);
  SSG_REQUIRE((S & 3) != 0 || ssg_aligned16(w2), SSG_EALIGN, "se_gate: w2 rows are read 16 B at a time when S %% 4 == 0: w2 must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nck = (C + SE_CK - 1) / SE_CK;
  hipLaunchKernelGGL(se_hidden_part_kernel, dim3((unsigned)((N * S + 255) / 256), (unsigned)nck), dim3(256), 0, st, sq, ldq, N, C, w1, S, tmp);
  SSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(se_gate_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), (size_t)N * S * sizeof(float), st, (const float*)tmp, nck, b1, N, S, w2, b2, C, h_pre, gate, ldg);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_se_gate_bwd_f32(const float* dgate, int ldd, const float* gate, int ldg, const float* h_pre, const float* sq, int ldq, int N, int C,
                                   const float* w1, const float* w2, int S, float* dsq, int lds_, float* dw1, float* db1, float* dw2, float* db2,
                                   float* tmp, void* stream) {
  SSG_REQUIRE(dgate && gate && h_pre && sq && w1 && w2 && dsq && dw1 && dw2 && tmp, SSG_EINVAL, "se_gate_bwd: null pointer");
  SSG_REQUIRE(se_shape_ok(N, C, S) && ldd >= C && ldg >= C && ldq >= C && lds_ >= C, SSG_EINVAL, "se_gate_bwd: N=%d C=%d S=%d outside the kernel's range", N, C, S);
  SSG_REQUIRE((S & 3) != 0 || (ssg_aligned16(w2) && ssg_aligned16(dw2)), SSG_EALIGN, "se_gate_bwd: w2 / dw2 rows are accessed 16 B at a time when S %% 4 == 0: both must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nck = (C + SE_CK - 1) / SE_CK;
  float* part = tmp;                                     // [nck][N][S]
  float* dz = tmp + (size_t)nck * N * S;                 // [N][C]
  const size_t lds_bytes = (size_t)N * S * sizeof(float);
  const unsigned cblk = (unsigned)((C + 255) / 256);
  hipLaunchKernelGGL(se_bwd_out_kernel, dim3(cblk), dim3(256), lds_bytes, st, dgate, ldd, gate, ldg, h_pre, N, S, C, dz, dw2, db2);
  SSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(se_dh_part_kernel, dim3((unsigned)((N * S + 255) / 256), (unsigned)nck), dim3(256), 0, st, (const float*)dz, N, C, w2, S, part);
  SSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(se_bwd_in_kernel, dim3(cblk), dim3(256), lds_bytes, st, (const float*)part, nck, h_pre, sq, ldq, w1, N, S, C, dsq, lds_, dw1, db1);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
