// Operand split shared by the bf16-pipe kernels (conv_igemm_halo_x3.hip, conv_igemm_dma_x3.hip): x = p1 + p2 + p3 with bf16
// terms (round-to-nearest-even conversions, exact residuals); weight rows are pre-split into [3 planes][16 channels] per K-step.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int XROW = 96;                  // bytes per weight row per K-step: 3 planes x 16 bf16, dense
// 16-B slot (plane p, k-half h) of row r sits at position (2p + h) ^ ((r >> 3) & 1): with the 96-B pitch the slot index of a row
// is 6r mod 16, which repeats every 8 rows -- swapping the two halves of a plane in every other group of 8 rows puts those on
// the odd slots, and the row-per-lane ds_read_b128 of a column fragment hits 16 distinct slots in every 16-lane service group
// (enumerated for both groups {0-3,12-15,20-27}, {4-11,16-19,28-31}; SQ_LDS_BANK_CONFLICT = 0 measured).

__device__ __forceinline__ void split3(const f32x4& u, const f32x4& v, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = e < 4 ? u[e] : v[e - 4];
    const __bf16 h = (__bf16)x;
    const float r = x - (float)h;
    const __bf16 m = (__bf16)r;
    const float r2 = r - (float)m;
    p1[e] = h; p2[e] = m; p3[e] = (__bf16)r2;
  }
}

// the same split for one quad of values
__device__ __forceinline__ void split3_4(const f32x4& u, bf16x4& p1, bf16x4& p2, bf16x4& p3) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = u[e];
    const __bf16 h = (__bf16)x;
    const float r = x - (float)h;
    const __bf16 m = (__bf16)r;
    const float r2 = r - (float)m;
    p1[e] = h; p2[e] = m; p3[e] = (__bf16)r2;
  }
}
