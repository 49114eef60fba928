// Kernel-argument block shared by the two implicit-GEMM conv kernels (internal).
#pragma once
#include "common.h"

struct ConvArgs {
  const float* in1; const float* in2;
  const float* w; const float* bias; const float* res; float* out;
  int C1, C2, ld1, ld2;
  int N, H, W;
  int Kp, kmode;
  int ldr, Cout, ldo;
  int GH, GW, OH, OW;
  int in_sy, in_sx, out_sy, out_sx, out_oy, out_ox;
  int ntaps;
  unsigned long long tap_bits;   // 6 bits per tap: (dy+2) | (dx+2)<<3
  int act; float slope;
  int tiles_x, tiles_y, nsteps;
  int ntiles_n, xcd_swizzle;     // conv_igemm_dma.hip: Cout tiles (fastest workgroup index), XCD-contiguous tile map
};


// conv_igemm_dma.hip: LDS-DMA pipeline for kmode 0, Cout > 32 (variant 0 = <128,128>, 1 = <256,64>)
int ssg_conv_igemm_dma_launch(const ConvArgs& a, int variant, hipStream_t st);
int ssg_conv_dma_variant(const ConvArgs& a, int variant);    // 0 = <128,128>, 1 = <256,64>, 2 = <128,64> (short K)

// conv_igemm_halo.hip: LDS-resident halo tile for the 9 taps of a 3x3 window (variant 0 = <128,128>, 1 = <256,64>)
bool ssg_conv_halo_ok(const ConvArgs& a);
int ssg_conv_igemm_halo_launch(const ConvArgs& a, int variant, hipStream_t st);
int ssg_conv_halo_variant(const ConvArgs& a, int variant);   // 0 = <128,128>, 1 = <256,64>, 2 = <128,64>
