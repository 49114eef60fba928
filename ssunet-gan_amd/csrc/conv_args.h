// Kernel-argument block shared by the two implicit-GEMM conv kernels (internal).
#pragma once
#include "common.h"

struct ConvArgs {
  const float* in1; const float* in2;
  const float* w; const float* bias; const float* res; float* out;
  double* bnpart;                // optional [M-tile][2][Cout] per-tile (sum, sum of squares) of the outputs (acc + bias), or NULL
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
  float* ws; int ksplit;         // conv_igemm_halo.hip: split-K slabs [ksplit][N*GH*GW][pad4(Cout)] (ksplit <= 1: off)
  int parity;                    // ssg_conv_desc.parity_merge
  const float* in_scale; const float* in_shift; int in_act; float in_slope;   // ssg_conv_desc.in_scale (conv_igemm_halo_k32.hip, conv_slow.h)
  const float* bwd_x; int bwd_ldx; const float* bwd_scale; const float* bwd_shift; const float* bwd_mean; int bwd_act; float bwd_slope;   // ssg_conv_desc.bwd_x (conv_igemm_halo_k32.hip)
  const float* w32;              // split-operand launches: the fp32 packed weights (ssg_conv_desc.w) beside the split pack in `w` (conv_slow.h)
};


// conv_igemm_dma.hip: LDS-DMA pipeline for kmode 0, Cout > 32 (variant 0 = <128,128>, 1 = <256,64>)
int ssg_conv_igemm_dma_launch(const ConvArgs& a, int variant, hipStream_t st);
int ssg_conv_dma_variant(const ConvArgs& a, int variant);    // 0 = <128,128>, 1 = <256,64>, 2 = <128,64> (short K)

// conv_igemm_halo.hip: LDS-resident halo tile for the 9 taps of a 3x3 window (variant 0 = <128,128>, 1 = <256,64>)
bool ssg_conv_halo_ok(const ConvArgs& a);
int ssg_conv_igemm_halo_launch(const ConvArgs& a, int variant, hipStream_t st);
int ssg_conv_halo_variant(const ConvArgs& a, int variant);   // 0 = <128,128>, 1 = <256,64>, 2 = <128,64>, 3 / 4 = <128,128> / <128,64> on 8x16-pixel tiles
void ssg_conv_halo_tile(int halo_variant, int* th, int* tw, int* bn);
int ssg_conv_halo_ksplit(const ConvArgs& a, int variant);    // split-K slabs the launch would use given a workspace (1 = none)

// conv_igemm_halo_x3.hip: the same tiles with operands split into three bf16 terms on the bf16 matrix pipe (a.w = split-packed weights)
bool ssg_conv_halo_x3_ok(const ConvArgs& a, int variant);
int ssg_conv_halo_x3_bn(const ConvArgs& a, int variant);      // column tile (64 / 128) the weights must be split-packed for
int ssg_conv_igemm_halo_x3_launch(const ConvArgs& a, int variant, hipStream_t st);
bool ssg_conv_halo_x3_parity_ok(const ConvArgs& a);          // the four parity classes of a 3x3 stride-2 input gradient as one launch
int ssg_conv_igemm_halo_x3_parity_launch(const ConvArgs& a, hipStream_t st);
// conv_igemm_halo_k32.hip (round 4): 32-channel chunks on v_mfma_f32_16x16x32_bf16; split-pack format code 1128 / 1064, or 0
int ssg_conv_halo_k32_fmt(const ConvArgs& a);
void ssg_conv_halo_k32_tile(int fmt, int* th, int* tw);
bool ssg_conv_halo_k32_in_affine_ok(const ConvArgs& a, int fmt);
bool ssg_conv_halo_k32_bwd_stats_ok(const ConvArgs& a, int fmt);
int ssg_conv_igemm_halo_k32_launch(const ConvArgs& a, int fmt, hipStream_t st);
int ssg_pack_split_k32_launch(const float* w_packed, int R, int Kp, int BN, void* out, hipStream_t st);
// conv_igemm_dma_x3.hip: the LDS-DMA pipeline (1x1, stride 2, parity classes) with split operands; column tile 128 / 64 or 0 = not eligible
int ssg_conv_dma_x3_bn(const ConvArgs& a);
int ssg_conv_igemm_dma_x3_launch(const ConvArgs& a, hipStream_t st);

// Batch-norm statistics in the conv epilogue (halo and DMA kernels): every lane adds up its output column over the rows it
// holds (<= 32 values) as fp32 deviations from a pivot, converted to fp64 sums of the values once per column (see the
// kernels: fp32 sums of the raw values lose var = E[x^2] - mean^2 when |mean| >> std); the fp64 partials are folded over the
// two lane halves and the WAVES_M waves, and one row [2][Cout-slice] per workgroup goes to a.bnpart;
// ssg_bn_stats_from_partials_f32 adds the rows in order.
// One column fragment j of one wave: (fp32 deviation sums s1, s2 around pivot `piv` over `nv` rows) -> fp64 sums of the values,
// folded over the two lane halves, into the LDS scratch [WAVES_M][2][BN].  Call between ssg_bnpart_begin and ssg_bnpart_finish.
template <int BN, int WTN>
__device__ __forceinline__ void ssg_bnpart_put(float* lds_f, int j, float s1, float s2, float piv, int nv, int wm, int wn, int half, int l31) {
  double* red = (double*)lds_f;
  const double c = (double)piv, n = (double)nv;
  double d1 = (double)s1 + n * c;
  double d2 = (double)s2 + 2.0 * c * (double)s1 + n * c * c;
  d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32);
  if (half == 0) {
    red[(wm * 2 + 0) * BN + wn * WTN + j * 32 + l31] = d1;
    red[(wm * 2 + 1) * BN + wn * WTN + j * 32 + l31] = d2;
  }
}
__device__ __forceinline__ void ssg_bnpart_begin() { __syncthreads(); }      // the K loop's LDS tiles are dead for every wave
template <int NI, int WAVES_M, int BN, int WTN>
__device__ __forceinline__ void ssg_bnpart_finish(const ConvArgs& a, float* lds_f, int mtile, int n0, int wm, int wn, int half, int l31) {
  double* red = (double*)lds_f;
  __syncthreads();
  if (wm == 0 && half == 0) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = wn * WTN + j * 32 + l31;
      const int co = n0 + col;
      if (co < a.Cout) {
        double t1 = 0, t2 = 0;
#pragma unroll
        for (int k = 0; k < WAVES_M; ++k) { t1 += red[(k * 2 + 0) * BN + col]; t2 += red[(k * 2 + 1) * BN + col]; }
        double* dst = a.bnpart + (size_t)mtile * 2 * a.Cout;
        dst[co] = t1; dst[a.Cout + co] = t2;
      }
    }
  }
}
