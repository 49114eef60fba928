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

// Every LDS read this wave has issued has returned.  REQUIRED before the workgroup barrier of a DMA ring step: the barrier is
// what lets another wave's LDS-DMA overwrite the stage those reads target (stage (s+2) % 3 == (s-1) % 3), and s_barrier does
// not wait for lgkmcnt -- a wave can arrive with its last fragment reads of step s-1 still in the LDS queue.  Without this the
// round-2 kernels produced a wrong 16-channel x 64-pixel patch in roughly one launch in ten when launched on an idle chip
// (tools/micro_halo_dbg.py), never in back-to-back launches.
__device__ __forceinline__ void wait_lds_reads() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
