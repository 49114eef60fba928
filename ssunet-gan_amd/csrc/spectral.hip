// Spectral normalisation (unwired per-op row A12; spec: the reference's scripts/spectral_norm.py:38-88,
// a vendored torch.nn.utils.spectral_norm): power iteration on W_mat = weight.view(Cout, -1),
//   v <- normalize(W^T u), u <- normalize(W v)   (n_power_iterations times, in place, eps 1e-12)
//   sigma = u . (W v),  W_sn = W / sigma.
// Two GEMVs over W per iteration: HBM-bound on W (<= a few MB), fp64 accumulation, deterministic.
// ABI: include/ssunet_hip.h (ssg_spectral_norm_fwd_f32 / _bwd_f32).
#include "common.h"

namespace {

// t[c] = sum_r W[r][c] * u[r]   (threads over columns: coalesced rows)
__global__ __launch_bounds__(256) void gemv_t_kernel(const float* __restrict__ W, int rows, int cols, const float* __restrict__ u, double* __restrict__ t) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  double s = 0;
  for (int r = 0; r < rows; ++r) s += (double)W[(size_t)r * cols + c] * (double)u[r];
  t[c] = s;
}
// s[r] = sum_c W[r][c] * v[c]   (one block per row)
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ W, int rows, int cols, const float* __restrict__ v, double* __restrict__ s) {
  __shared__ double red[256];
  const int r = blockIdx.x;
  double a = 0;
  for (int c = threadIdx.x; c < cols; c += 256) a += (double)W[(size_t)r * cols + c] * (double)v[c];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) s[r] = red[0];
}
// out = in / max(||in||, eps); optionally sigma = out . in
__global__ __launch_bounds__(256) void normalize_kernel(const double* __restrict__ in, int n, double eps, float* __restrict__ out, float* __restrict__ sigma) {
  __shared__ double red[256];
  double a = 0;
  for (int i = threadIdx.x; i < n; i += 256) a += in[i] * in[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const double nrm = sqrt(red[0]);
  const double d = nrm > eps ? nrm : eps;
  __syncthreads();
  double dot = 0;
  for (int i = threadIdx.x; i < n; i += 256) { const float o = (float)(in[i] / d); out[i] = o; dot += (double)o * in[i]; }
  if (sigma) {
    red[threadIdx.x] = dot;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *sigma = (float)red[0];
  }
}
// sigma = u . s  (no update of u: eval mode / zero iterations)
__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ u, const double* __restrict__ s, int n, float* __restrict__ sigma) {
  __shared__ double red[256];
  double a = 0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)u[i] * s[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *sigma = (float)red[0];
}
__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ W, long long n, const float* __restrict__ sigma, float* __restrict__ out) {
  const float inv = 1.f / *sigma;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = W[i] * inv;
}
// partial sums of dWsn . W (fp64), one per block
__global__ __launch_bounds__(256) void dotall_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, double* __restrict__ part) {
  __shared__ double red[256];
  double s = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += (double)a[i] * (double)b[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
// dW = (dWsn - (sum(dWsn . W) / sigma^2) * u v^T ... ) : with W_sn = W/sigma and sigma = u^T W v,
//   dW[r][c] = dWsn[r][c]/sigma - (D / sigma^2) * u[r] v[c],  D = sum dWsn . W
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ dWsn, int rows, int cols, const float* __restrict__ u,
                                                     const float* __restrict__ v, const float* __restrict__ sigma, const double* __restrict__ part,
                                                     int nparts, float* __restrict__ dW) {
  __shared__ double Dsh;
  if (threadIdx.x == 0) { double d = 0; for (int i = 0; i < nparts; ++i) d += part[i]; Dsh = d; }
  __syncthreads();
  const double sg = (double)*sigma;
  const float k = (float)(Dsh / (sg * sg)), inv = (float)(1.0 / sg);
  const long long n = (long long)rows * cols;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    dW[i] = dWsn[i] * inv - k * u[r] * v[c];
  }
}

}  // namespace

extern "C" int64_t ssg_spectral_norm_workspace_bytes(int rows, int cols) {
  return (int64_t)(rows + cols + 512) * (int64_t)sizeof(double);
}

extern "C" int ssg_spectral_norm_fwd_f32(const float* W, int rows, int cols, float* u, float* v, int n_power_iterations, double eps,
                                         float* W_out, float* sigma, void* ws, void* stream) {
  SSG_REQUIRE(W && u && v && sigma && ws && rows > 0 && cols > 0 && n_power_iterations >= 0, SSG_EINVAL, "spectral_norm: bad args");
  hipStream_t st = (hipStream_t)stream;
  double* t = (double*)ws;          // [cols]
  double* s = t + cols;             // [rows]
  for (int it = 0; it < n_power_iterations; ++it) {
    hipLaunchKernelGGL(gemv_t_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st, W, rows, cols, u, t);
    hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(256), 0, st, t, cols, eps, v, (float*)nullptr);
    hipLaunchKernelGGL(gemv_kernel, dim3((unsigned)rows), dim3(256), 0, st, W, rows, cols, v, s);
    hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(256), 0, st, s, rows, eps, u, it + 1 == n_power_iterations ? sigma : (float*)nullptr);
  }
  if (n_power_iterations == 0) {
    hipLaunchKernelGGL(gemv_kernel, dim3((unsigned)rows), dim3(256), 0, st, W, rows, cols, v, s);
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(256), 0, st, u, s, rows, sigma);
  }
  if (W_out) {                       // W_out == NULL: sigma only (the conv applies 1/sigma while packing W: ssg_pack_weights_scaled_f32)
    const long long n = (long long)rows * cols;
    long long g = (n + 255) / 256; if (g > 2048) g = 2048;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)g), dim3(256), 0, st, W, n, sigma, W_out);
  }
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_spectral_norm_bwd_f32(const float* dWsn, const float* W, int rows, int cols, const float* u, const float* v,
                                         const float* sigma, float* dW, void* ws, void* stream) {
  SSG_REQUIRE(dWsn && W && u && v && sigma && dW && ws && rows > 0 && cols > 0, SSG_EINVAL, "spectral_norm_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)rows * cols;
  int nparts = (int)((n + 255) / 256); if (nparts > 512) nparts = 512;
  hipLaunchKernelGGL(dotall_partial_kernel, dim3((unsigned)nparts), dim3(256), 0, st, dWsn, W, n, (double*)ws);
  long long g = (n + 255) / 256; if (g > 2048) g = 2048;
  hipLaunchKernelGGL(sn_bwd_kernel, dim3((unsigned)g), dim3(256), 0, st, dWsn, rows, cols, u, v, sigma, (const double*)ws, nparts, dW);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
