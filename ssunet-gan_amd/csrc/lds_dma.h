// global -> LDS direct loads (gfx950 `global_load_lds_dwordx4`) shared by the four LDS-DMA kernels (internal).
// One call moves 16 B per lane; the LDS image of a wave-instruction is lane-linear (dest = base + lane * 16 B), so a
// 1-KiB "piece" is addressed by its wave-uniform base and every permutation lives on the SOURCE address.
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) void ssg_lds_void;
typedef const __attribute__((address_space(1))) void ssg_gbl_void;

__device__ __forceinline__ void dma16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((ssg_gbl_void*)src, (ssg_lds_void*)lds_dst, 16, 0, 0);
}

// s_waitcnt vmcnt(N) with an immediate; the "memory" clobber keeps LDS reads below it
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
