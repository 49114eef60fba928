// Fused segmentation loss + metrics for gfx950 (one HBM pass over logits and targets).
// ABI + reference citations: include/ssunet_hip.h (ssg_seg_loss_*, ssg_bce_logits_const_*).
//
// Forward: every block owns a pixel range of ONE sample and reduces 10 quantities in fp64:
//   0 sum p*t   1 sum p   2 sum t          (per-sample soft-Dice terms, losses.py:291)
//   3 sum bce_elem   4 sum (x-t)^2         (StableBCE / MSE numerators)
//   5 |pred & tgt|   6 |pred | tgt|        (hard IoU on channels >= mc0, metrics.py:13-21)
//   7 sum p*t  8 sum p  9 sum t  on channels >= mc0 (soft Dice metric, metrics.py:32-35)
// A one-block finalize kernel adds block partials in order and evaluates the scalars on
// the device, so the training loop needs no host sync for loss / IoU / Dice.
#include "common.h"

namespace {

constexpr int NQ = 10;
constexpr int LOSS_BLOCK = 256;

struct LossGeom { int bps; long long pix_per_block; };   // blocks per sample

__host__ LossGeom loss_geom(int N, long long S) {
  LossGeom g;
  long long bps = 2048 / (N > 0 ? N : 1);
  if (bps < 1) bps = 1;
  const long long maxb = (S + LOSS_BLOCK - 1) / LOSS_BLOCK;
  if (bps > maxb) bps = maxb;
  g.pix_per_block = (S + bps - 1) / bps;
  g.bps = (int)((S + g.pix_per_block - 1) / g.pix_per_block);
  return g;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(LOSS_BLOCK) void seg_loss_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ t,
                                                                      int ldt, long long S, int C, int mc0, long long ppb,
                                                                      double* __restrict__ part) {
  __shared__ double red[LOSS_BLOCK / 64][NQ];
  const int n = blockIdx.y;
  const long long p0 = (long long)blockIdx.x * ppb;
  long long p1 = p0 + ppb; if (p1 > S) p1 = S;
  float a[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) a[k] = 0.f;
  double d[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) d[k] = 0.0;
  int cnt = 0;
  for (long long p = p0 + threadIdx.x; p < p1; p += LOSS_BLOCK) {
    const float* xp = x + ((size_t)n * S + p) * ldx;
    const float* tp = t + ((size_t)n * S + p) * ldt;
    for (int c = 0; c < C; ++c) {
      const float xv = xp[c], tv = tp[c];
      const float pv = sigmoidf_(xv);
      a[0] += pv * tv; a[1] += pv; a[2] += tv;
      a[3] += (xv < 0.f ? 0.f : xv) - xv * tv + logf(1.f + expf(-fabsf(xv)));
      const float df = xv - tv; a[4] += df * df;
      if (c >= mc0) {
        const bool po = pv > 0.5f, to = tv > 0.5f;
        a[5] += (po && to) ? 1.f : 0.f; a[6] += (po || to) ? 1.f : 0.f;
        a[7] += pv * tv; a[8] += pv; a[9] += tv;
      }
    }
    if (++cnt == 64) {   // flush float partials to fp64 regularly
#pragma unroll
      for (int k = 0; k < NQ; ++k) { d[k] += a[k]; a[k] = 0.f; }
      cnt = 0;
    }
  }
#pragma unroll
  for (int k = 0; k < NQ; ++k) d[k] += a[k];
  // wave reduce (64 lanes), then across the 4 waves
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    double v = d[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    d[k] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NQ; ++k) red[wave][k] = d[k];
  __syncthreads();
  if (threadIdx.x < NQ) {
    double v = 0;
    for (int w = 0; w < LOSS_BLOCK / 64; ++w) v += red[w][threadIdx.x];
    part[((size_t)n * gridDim.x + blockIdx.x) * NQ + threadIdx.x] = v;
  }
}

// Single block: thread (n, k) adds the per-block partials of quantity k of sample n in block order (samples in
// chunks of 64), then one thread combines the samples in sample order -- the summation order of a serial loop.
constexpr int FINAL_CHUNK = 64;
__global__ __launch_bounds__(1024) void seg_loss_final_kernel(const double* __restrict__ part, int N, int bps, long long S, int C, int mc0,
                                                            float* __restrict__ res, double* __restrict__ stats) {
  __shared__ double q[FINAL_CHUNK][NQ];
  __shared__ double tot[NQ];
  __shared__ double dice_sum;
  if (threadIdx.x < NQ) tot[threadIdx.x] = 0;
  if (threadIdx.x == 0) dice_sum = 0;
  __syncthreads();
  for (int n0 = 0; n0 < N; n0 += FINAL_CHUNK) {
    const int nn = N - n0 < FINAL_CHUNK ? N - n0 : FINAL_CHUNK;
    for (int t = threadIdx.x; t < nn * NQ; t += blockDim.x) {
      const int n = t / NQ, k = t - n * NQ;
      double a = 0;
      for (int b = 0; b < bps; ++b) a += part[((size_t)(n0 + n) * bps + b) * NQ + k];
      q[n][k] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int n = 0; n < nn; ++n) {
        stats[(n0 + n) * 3 + 0] = q[n][0]; stats[(n0 + n) * 3 + 1] = q[n][1]; stats[(n0 + n) * 3 + 2] = q[n][2];
        dice_sum += (2.0 * q[n][0] + 1e-5) / (q[n][1] + q[n][2] + 1e-5);
        for (int k = 3; k < NQ; ++k) tot[k] += q[n][k];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double numel = (double)N * (double)S * (double)C;
    const double bce = tot[3] / numel;
    const double mse = tot[4] / numel;
    const double dice = 1.0 - dice_sum / N;
    const bool finite = isfinite(bce);
    res[0] = (float)(finite ? 0.5 * bce + dice : 2.0 * dice);
    res[1] = (float)mse;
    res[2] = (float)bce;
    res[3] = (float)dice;
    res[4] = (float)((tot[5] + 1e-5) / (tot[6] + 1e-5));
    res[5] = (float)((2.0 * tot[7] + 1e-5) / (tot[8] + tot[9] + 1e-5));
    res[6] = finite ? 1.f : 0.f;
    res[7] = 0.f;
    for (int k = 0; k < 5; ++k) stats[3 * N + k] = tot[5 + k];     // metric partial sums (for cross-rank reduction)
  }
}

__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ t, int ldt,
                                                           int N, long long S, int C, const float* __restrict__ res,
                                                           const double* __restrict__ stats, const float* __restrict__ g_seg,
                                                           const float* __restrict__ g_mse, const float* __restrict__ g_bce,
                                                           float* __restrict__ dx, int lddx) {
  const long long total = (long long)N * S;
  const double numel = (double)N * (double)S * (double)C;
  const float gs = g_seg ? *g_seg : 0.f, gm = g_mse ? *g_mse : 0.f;
  const float gb = g_bce ? (float)((double)*g_bce / numel) : 0.f;       // res[2] = StableBCELoss alone (losses.py:130-136)
  const bool finite = res[6] != 0.f;
  const float k_bce = finite ? (float)(0.5 / numel) : 0.f;
  const float k_dice = finite ? 1.f : 2.f;
  const float k_mse = (float)(2.0 / numel);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / S);
    const double I = stats[n * 3 + 0], U = stats[n * 3 + 1] + stats[n * 3 + 2] + 1e-5;
    const float invU = (float)(1.0 / U), ratio = (float)((2.0 * I + 1e-5) / (U * U));
    const float* xp = x + (size_t)i * ldx;
    const float* tp = t + (size_t)i * ldt;
    float* dp = dx + (size_t)i * lddx;
    for (int c = 0; c < lddx; ++c) {
      float g = 0.f;
      if (c < C) {
        const float xv = xp[c], tv = tp[c];
        const float pv = sigmoidf_(xv);
        // d(1 - mean_n dice_n)/dp = -(1/N) * (2 t U - (2I + s)) / U^2
        const float ddice = -(2.f * tv * invU - ratio) / (float)N;
        g = gs * (k_bce * (pv - tv) + k_dice * ddice * pv * (1.f - pv)) + gm * k_mse * (xv - tv);
        if (gb != 0.f) g += gb * (pv - tv);
      }
      dp[c] = g;
    }
  }
}

__global__ void bce_const_fwd_kernel(const float* __restrict__ x, int n, int ldx, float label, float* __restrict__ loss) {
  __shared__ double red[256];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float xv = x[(size_t)i * ldx];
    // (1-y)*x + max(-x,0) + log1p(exp(-|x|))  (ATen binary_cross_entropy_with_logits)
    s += (double)((1.f - label) * xv + (xv < 0.f ? -xv : 0.f) + log1pf(expf(-fabsf(xv))));
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *loss = (float)(red[0] / n);
}
__global__ void bce_const_bwd_kernel(const float* __restrict__ x, int n, int ldx, float label, const float* __restrict__ g,
                                     float* __restrict__ dx, int lddx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    dx[(size_t)i * lddx] = (*g) * (sigmoidf_(x[(size_t)i * ldx]) - label) / (float)n;
    for (int c = 1; c < lddx; ++c) dx[(size_t)i * lddx + c] = 0.f;      // pad columns stay zero
  }
}

}  // namespace

extern "C" int64_t ssg_seg_loss_workspace_bytes(int N, int64_t S, int C) {
  (void)C;
  const LossGeom g = loss_geom(N, S);
  return (int64_t)N * g.bps * NQ * (int64_t)sizeof(double);
}

extern "C" int ssg_seg_loss_fwd_f32(const float* x, int ldx, const float* t, int ldt, int N, int64_t S, int C, int mc0, float* res,
                                    double* stats, void* ws, void* stream) {
  SSG_REQUIRE(x && t && res && stats && ws && N > 0 && S > 0 && C > 0 && ldx >= C && ldt >= C && mc0 >= 0, SSG_EINVAL, "seg_loss: bad args");
  const LossGeom g = loss_geom(N, S);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(seg_loss_partial_kernel, dim3((unsigned)g.bps, (unsigned)N), dim3(LOSS_BLOCK), 0, st, x, ldx, t, ldt, (long long)S, C, mc0,
                     g.pix_per_block, (double*)ws);
  SSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(seg_loss_final_kernel, dim3(1), dim3(1024), 0, st, (const double*)ws, N, g.bps, (long long)S, C, mc0, res, stats);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_seg_loss_bwd_f32(const float* x, int ldx, const float* t, int ldt, int N, int64_t S, int C, const float* res,
                                    const double* stats, const float* g_seg, const float* g_mse, const float* g_bce, float* dx, int lddx,
                                    void* stream) {
  SSG_REQUIRE(x && t && res && stats && dx && N > 0 && S > 0 && C > 0 && lddx >= C, SSG_EINVAL, "seg_loss_bwd: bad args");
  long long grid = ((long long)N * S + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(seg_loss_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, ldx, t, ldt, N, (long long)S, C, res,
                     stats, g_seg, g_mse, g_bce, dx, lddx);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_bce_logits_const_fwd_f32(const float* x, int n, int ldx, float label, float* loss, void* stream) {
  SSG_REQUIRE(x && loss && n > 0 && ldx > 0, SSG_EINVAL, "bce_const: bad args");
  hipLaunchKernelGGL(bce_const_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, ldx, label, loss);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_bce_logits_const_bwd_f32(const float* x, int n, int ldx, float label, const float* g, float* dx, int lddx, void* stream) {
  SSG_REQUIRE(x && g && dx && n > 0 && ldx > 0 && lddx > 0, SSG_EINVAL, "bce_const_bwd: bad args");
  hipLaunchKernelGGL(bce_const_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, ldx, label, g, dx, lddx);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
