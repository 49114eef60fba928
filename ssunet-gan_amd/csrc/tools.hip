// Calibration kernels (not on the hot path): measured fp32-MFMA issue peak and HBM copy rate of
// THIS device, so roofline fractions can be read against what the box sustains as well as the spec.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f + blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a += 1e-7f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// the same loop on the bf16 pipe: v_mfma_f32_32x32x16_bf16 (32768 FLOP each), what the split-operand kernels multiply on
__global__ __launch_bounds__(256) void mfma_peak_bf16_kernel(float* out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8_t a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + (threadIdx.x + e) * 0.0078125f); b[e] = (__bf16)(0.5f + (blockIdx.x % 64 + e) * 0.00390625f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    asm volatile("" : "+v"(a));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ... and with operands that toggle: 4 + 4 fragments of caller data (random bf16), cycled.  The register-only loops above run
// on constants, which cost the matrix pipe little power; this one shows the rate the chip sustains on real operands.
__global__ __launch_bounds__(256) void mfma_peak_bf16_data_kernel(float* out, int iters, const bf16x8_t* __restrict__ data) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8_t a[4], b[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { a[u] = data[(u * 256 + threadIdx.x) & 4095]; b[u] = data[(1024 + u * 256 + threadIdx.x + blockIdx.x) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[i], acc[i], 0, 0, 0);
    asm volatile("" : "+v"(a[0]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ... and on the v_mfma_f32_16x16x32_bf16 shape (the round-4 k32 kernels): the same output tile per wave (16 accumulators of 4), the
// same FLOPs per iteration.  On random operands this shape sustains ~1.15x the FLOP/s of 32x32x16 at the clock the chip then holds.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void mfma_peak_bf16_data16_kernel(float* out, int iters, const bf16x8_t* __restrict__ data) {
  f32x4_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t a[4], b[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { a[u] = data[(u * 256 + threadIdx.x) & 4095]; b[u] = data[(1024 + u * 256 + threadIdx.x + blockIdx.x) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u * 2 + (i >> 3)) & 3], b[i & 3], acc[i], 0, 0, 0);
    asm volatile("" : "+v"(a[0]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// One float4 per thread, non-temporal both ways: the form of MI355X_MICROARCH.md's copy ceiling (6.3-6.6 TB/s on 1-GiB buffers; the
// grid-stride loop over 4096 workgroups that this probe used through round 3 reads 4.5-4.8 and under-states what the box sustains).
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long nq) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < nq) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace

// Launches `blocks` workgroups of 4 waves, each wave issuing iters*16 independent-accumulator
// 32x32x2 fp32 MFMAs (4096 FLOP each).  FLOPs = blocks * 4 * iters * 16 * 4096.
extern "C" int ssg_tool_mfma_peak_f32(float* scratch, int blocks, int iters, void* stream) {
  SSG_REQUIRE(scratch && blocks > 0 && iters > 0, SSG_EINVAL, "mfma_peak: bad args");
  hipLaunchKernelGGL(mfma_peak_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, scratch, iters);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

// Same launch shape on the bf16 pipe.  FLOPs = blocks * 4 * iters * 16 * 32768.
extern "C" int ssg_tool_mfma_peak_bf16(float* scratch, int blocks, int iters, void* stream) {
  SSG_REQUIRE(scratch && blocks > 0 && iters > 0, SSG_EINVAL, "mfma_peak_bf16: bad args");
  hipLaunchKernelGGL(mfma_peak_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, scratch, iters);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

// `data`: 4096 x 16 B of bf16 operands (64 KiB).
extern "C" int ssg_tool_mfma_peak_bf16_data(float* scratch, int blocks, int iters, const void* data, void* stream) {
  SSG_REQUIRE(scratch && data && blocks > 0 && iters > 0, SSG_EINVAL, "mfma_peak_bf16_data: bad args");
  hipLaunchKernelGGL(mfma_peak_bf16_data_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, scratch, iters, (const bf16x8_t*)data);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

// FLOPs = blocks * 4 * iters * 32 * 16384 (= the count of ssg_tool_mfma_peak_bf16_data for the same arguments)
extern "C" int ssg_tool_mfma_peak_bf16_data16(float* scratch, int blocks, int iters, const void* data, void* stream) {
  SSG_REQUIRE(scratch && data && blocks > 0 && iters > 0, SSG_EINVAL, "mfma_peak_bf16_data16: bad args");
  hipLaunchKernelGGL(mfma_peak_bf16_data16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, scratch, iters, (const bf16x8_t*)data);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_tool_copy_f32(const float* src, float* dst, int64_t n, void* stream) {
  SSG_REQUIRE(src && dst && n > 0 && n % 4 == 0 && n / 4 / 256 < (1ll << 31), SSG_EINVAL, "copy: bad args");
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, (long long)(n / 4));
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
