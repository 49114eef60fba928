#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 36 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(float* d, int waves_per_simd) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * waves_per_simd), block(256);
  hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid.x * 4 * iters * 36 * 512;
  printf("nacc %d waves/simd %d: %.3f ms %.1f TF\n", NACC, waves_per_simd, ms, flops / ms * 1e-9);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<1>(d, 1); run<1>(d, 4); run<4>(d, 1); run<4>(d, 4); run<9>(d, 1); run<9>(d, 4); run<36>(d, 1); run<36>(d, 2);
  return 0;
}
