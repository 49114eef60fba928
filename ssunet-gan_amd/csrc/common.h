// Shared helpers for the gfx950 hot-path library (internal; the ABI is include/ssunet_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ssunet_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

__device__ __forceinline__ float ssg_act(float v, int act, float slope) {
  // NaN-propagating forms, as ATen's relu (clamp_min) and leaky_relu
  if (act == SSG_ACT_RELU) return v < 0.f ? 0.f : v;
  if (act == SSG_ACT_LRELU) return v > 0.f ? v : v * slope;
  return v;
}
