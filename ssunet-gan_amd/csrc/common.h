// Shared helpers for the gfx950 hot-path library (internal; the ABI is include/ssunet_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ssunet_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Storage types of the HBM-bound kernels: fp32, or bf16 (BASELINE config 4: bf16 tensors in HBM, fp32 arithmetic in
// registers, fp64 statistics).  A kernel templated on T reads and writes 4 channels per lane through these two overloads
// (16 B of fp32 or 8 B of bf16); everything between them is fp32 and identical for both.
typedef __bf16 ssg_bf16;
typedef __bf16 ssg_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 ld4(const ssg_bf16* p) { return __builtin_convertvector(*(const ssg_bf16x4*)p, f32x4); }
#ifndef SSG_NT_STORES
#define SSG_NT_STORES 0      // 1: streaming stores of the element-wise kernels are non-temporal (A/B build switch)
#endif
#if SSG_NT_STORES
__device__ __forceinline__ void st4(float* p, f32x4 v) { __builtin_nontemporal_store(v, (f32x4*)p); }
__device__ __forceinline__ void st4(ssg_bf16* p, f32x4 v) { __builtin_nontemporal_store(__builtin_convertvector(v, ssg_bf16x4), (ssg_bf16x4*)p); }
#else
__device__ __forceinline__ void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void st4(ssg_bf16* p, f32x4 v) { *(ssg_bf16x4*)p = __builtin_convertvector(v, ssg_bf16x4); }
#endif

// 16 bytes per lane for either type: SSG_Q<T> channel quads per memory access (1 for fp32, 2 for bf16).  The element-wise
// kernels that are latency-bound at 8 B per lane (one load in flight per thread) use these: a bf16 thread then moves as many
// bytes per instruction as an fp32 one and works on 8 channels.  Needs C % (4 * SSG_Q<T>) == 0 and 16-byte aligned rows.
// Per-thread accumulation type of the reducing kernels: fp64 for fp32 tensors (ATen's CPU kernels accumulate float tensors in
// double, and var = E[x^2] - mean^2 cancels: DESIGN.md 3.2); fp32 for bf16 tensors, whose own rounding (2^-9) is four orders
// above fp32 accumulation error over the <= few hundred terms a thread sums -- the cross-thread / cross-block stages stay
// fp64 for both.  (fp64 VALU runs at half rate: with half the bytes per element the bf16 instantiations were VALU-bound.)
template <typename T> struct SsgAcc { typedef double type; };
template <> struct SsgAcc<__bf16> { typedef float type; };
typedef __bf16 ssg_bf16x8 __attribute__((ext_vector_type(8)));
template <typename T> struct SsgQ { static constexpr int value = 1; };
template <> struct SsgQ<ssg_bf16> { static constexpr int value = 2; };
__device__ __forceinline__ void ldq(const float* p, f32x4 (&v)[1]) { v[0] = *(const f32x4*)p; }
__device__ __forceinline__ void ldq(const ssg_bf16* p, f32x4 (&v)[2]) {
  const ssg_bf16x8 r = *(const ssg_bf16x8*)p;
  v[0] = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
  v[1] = f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
}
__device__ __forceinline__ void stq(float* p, const f32x4 (&v)[1]) { st4(p, v[0]); }
__device__ __forceinline__ void stq(ssg_bf16* p, const f32x4 (&v)[2]) {
  ssg_bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = (ssg_bf16)v[0][e]; r[4 + e] = (ssg_bf16)v[1][e]; }
#if SSG_NT_STORES
  __builtin_nontemporal_store(r, (ssg_bf16x8*)p);
#else
  *(ssg_bf16x8*)p = r;
#endif
}

void ssg_set_error(const char* fmt, ...);

#define SSG_REQUIRE(cond, code, ...)            \
  do {                                          \
    if (!(cond)) {                              \
      ssg_set_error(__VA_ARGS__);               \
      return (code);                            \
    }                                           \
  } while (0)

#define SSG_LAUNCH_CHECK()                                                  \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      ssg_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,          \
                    hipGetErrorString(e__));                                \
      return (int)e__;                                                      \
    }                                                                       \
  } while (0)

static inline bool ssg_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline int64_t ssg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// sigmoid on the hardware transcendentals (v_exp_f32 + v_rcp_f32, ~1 ulp each; |relative error| < 1e-6 for |x| < 10):
// the library expf + IEEE division made the swish-fused batch-norm kernels VALU-bound (2.5 TB/s on 0.9-GB tensors).
__device__ __forceinline__ float ssg_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__device__ __forceinline__ float ssg_act(float v, int act, float slope) {
  // NaN-propagating forms, as ATen's relu (clamp_min) and leaky_relu
  if (act == SSG_ACT_RELU) return v < 0.f ? 0.f : v;
  if (act == SSG_ACT_LRELU) return v > 0.f ? v : v * slope;
  if (act == SSG_ACT_SWISH) return v * ssg_sigmoid_fast(v);              // x * sigmoid(x) (efficientnet_pytorch/utils.py:37-48)
  return v;
}

// d act(z) / dz for the activations whose derivative is a function of the pre-activation z
__device__ __forceinline__ float ssg_swish_grad(float z) {
  const float s = ssg_sigmoid_fast(z);
  return s * (1.f + z * (1.f - s));                                     // utils.py:45-48
}

// Element-wise kernels: every block owns a CONTIGUOUS chunk of the index range and its threads stride 256 inside it, on a grid of
// ~total / (256 * items) blocks.  The grid-stride loop over 4096 blocks that these kernels used through round 3 strides 16 MB per
// iteration (a 1-GiB copy: 4.5-4.8 TB/s); contiguous chunks read 5.4-5.9, one item per thread 6.3 (tools/mfma_lab.hip, `copy`).
__device__ __forceinline__ long long ssg_chunk_len(long long total) { return (((total + gridDim.x - 1) / gridDim.x) + 255) / 256 * 256; }
// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), the result in every lane of the row: four data-parallel-primitive moves on
// the vector unit (quad swaps, half-row mirror, row mirror).  __shfl_xor compiles to ds_bpermute_b32 -- an LDS-crossbar operation per step:
// the 128 of them in the statistics epilogue of conv_halo_k32_kernel were 7-10 % of a Cin = 64 tile.
__device__ __forceinline__ float ssg_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // row_mirror
  return v;
}
// Workgroup ids are dealt to the 8 XCDs round-robin; map id -> position so that XCD k owns the k-th contiguous band of [0, n) and
// walks it in dispatch order (neighbouring positions then share one L2).
__device__ __forceinline__ unsigned ssg_xcd_band(unsigned id, unsigned n) {
  const unsigned per = n >> 3, rem = n & 7u, k = id & 7u;
  return k * per + (k < rem ? k : rem) + (id >> 3);
}
__device__ __forceinline__ long long ssg_chunk_begin(long long total) { return (long long)blockIdx.x * ssg_chunk_len(total); }
__device__ __forceinline__ long long ssg_chunk_end(long long total) {
  const long long e = ((long long)blockIdx.x + 1) * ssg_chunk_len(total);
  return e < total ? e : total;
}
#define SSG_CHUNK_LOOP(i, total) \
  for (long long i = ssg_chunk_begin(total) + threadIdx.x, ssg_end__ = ssg_chunk_end(total); i < ssg_end__; i += 256)
static inline int ssg_elem_grid(long long total, int items) {
  long long g = (total + 256ll * items - 1) / (256ll * items);
  if (g > 0x7fffffffll) g = 0x7fffffffll;
  if (g < 1) g = 1;
  return (int)g;
}
