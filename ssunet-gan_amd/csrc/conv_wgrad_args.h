// Kernel-argument block shared by the two weight-gradient kernels (internal).
#pragma once
#include "common.h"

struct WgArgs {
  const float* in1; const float* in2; const float* dout; float* ws;
  int C1, C2, ld1, ld2, N, H, W, Cout, ldd, GH, GW, in_sy, in_sx, ntaps;
  unsigned long long tap_bits;
  int M;                 // ntaps * Cin
  long long Ptot;        // N*GH*GW
  int steps_per_split;   // K-steps (16 pixels each) per z-slice
  int xcd_swizzle;       // conv_wgrad_dma.hip: sharers of an operand on one XCD (set by the launcher)
  const float* in_scale; const float* in_shift; int in_act; float in_slope;   // ssg_wgrad_desc.in_scale (conv_wgrad_k32.hip)
};


// conv_wgrad_dma.hip: LDS-DMA pipeline for Cout > 32 (variant 0 = <128,128>, 1 = <128,64>)
int ssg_wgrad_dma_launch(const WgArgs& a, int variant, dim3 grid, hipStream_t st, bool split);

// conv_wgrad_halo.hip: 3x3 stride-1 weight gradient with an LDS-resident pixel window
// (variant 0 = 9 taps x 32 channels x 128 output channels per workgroup, 1 = 9 x 64 x 64)
int ssg_wgrad_halo_cb(int variant);
bool ssg_wgrad_halo_ok(const ssg_wgrad_desc* d, int variant);
int ssg_wgrad_halo_launch(const WgArgs& a, int variant, dim3 grid, hipStream_t st, bool split);   // split: operands as three bf16 terms on the bf16 pipe

// conv_wgrad_k32.hip (round 4): split operands on v_mfma_f32_16x16x32_bf16, 9 x 64 x 64 tiles of 512 threads, rolling 3-row window
bool ssg_wgrad_k32_ok(const ssg_wgrad_desc* d);
long long ssg_wgrad_k32_steps(const ssg_wgrad_desc* d);
int ssg_wgrad_k32_flush();                                 // rows per in-register flush (0: none -- slabs then stay <= 128 rows)
int ssg_wgrad_k32_launch(const WgArgs& a, dim3 grid, hipStream_t st);
