// Weight-gradient implicit GEMM, LDS-DMA pipeline (gfx950).  Same contract and slab layout as
// wgrad_kernel in conv_wgrad.hip; selected for Cout > 32.
//   * [16 pixels][rows] LDS images are exactly the DMA's lane-linear layout (a 1-KiB piece = 2
//     pixels x 128 rows, or 4 pixels x 64 rows), read back with conflict-free ds_read_b32, so the
//     register staging, its ds_write pass and the padding disappear;
//   * 3 stages, two K-steps (2 x 16 pixels) in flight across the one barrier per step;
//   * shifted/out-of-image taps and the pixel tail read a zero page instead of being predicated;
//   * the per-lane pixel coordinate advances incrementally (no divisions in the loop).
#include "common.h"
#include "lds_dma.h"
#include "conv_wgrad_args.h"

namespace {

__device__ __attribute__((aligned(64))) float ssg_zero_page_w[64];


constexpr int BKP = 16;
constexpr int NSTAGE = 3;

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(const WgArgs a) {
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int A_PC = BKP * BM / 256 / 4;       // pieces per wave per step (BM=128: 2)
  constexpr int B_PC = BKP * BN / 256 / 4;       // BN=128: 2, BN=64: 1
  constexpr int STAGE = BKP * (BM + BN);
  static_assert(B_PC >= 1, "BN >= 64");

  __shared__ __attribute__((aligned(1024))) float lds[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int half = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int Cin = a.C1 + a.C2;

  const long long step0 = (long long)blockIdx.z * a.steps_per_split;
  long long nst = (a.Ptot + BKP - 1) / BKP - step0;
  if (nst > a.steps_per_split) nst = a.steps_per_split;
  const int nsteps = nst > 0 ? (int)nst : 0;
  const int GHW = a.GH * a.GW;

  // ---- A pieces: lane -> (pixel within step, row quad); row quad fixes (tap, channel)
  const float* a_base[A_PC]; int a_ld[A_PC]; int a_dy[A_PC], a_dx[A_PC];
  int a_n[A_PC], a_gy[A_PC], a_gx[A_PC];          // running pixel coordinate of this lane's pixel
  bool a_rowok[A_PC];
#pragma unroll
  for (int j = 0; j < A_PC; ++j) {
    const int idx = (wave * A_PC + j) * 64 + lane;            // float4 index inside the [16][BM] image
    const int px = idx / (BM / 4), rq = idx % (BM / 4);
    const int row = m0 + 4 * rq;
    a_rowok[j] = row < a.M;
    int t = 0, c = 0;
    if (a_rowok[j]) { t = row / Cin; c = row - t * Cin; }
    const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
    a_dy[j] = (tb & 7) - 2; a_dx[j] = (tb >> 3) - 2;
    if (c < a.C1) { a_base[j] = a.in1 + c; a_ld[j] = a.ld1; } else { a_base[j] = a.in2 + (c - a.C1); a_ld[j] = a.ld2; }
    const long long P = step0 * BKP + px;
    a_n[j] = (int)(P / GHW); const int rem = (int)(P - (long long)a_n[j] * GHW);
    a_gy[j] = rem / a.GW; a_gx[j] = rem - a_gy[j] * a.GW;
  }
  // ---- B pieces: dout rows are contiguous, lane -> (pixel within step, column quad)
  long long b_off[B_PC]; bool b_colok[B_PC]; long long b_P[B_PC];
#pragma unroll
  for (int j = 0; j < B_PC; ++j) {
    const int idx = (wave * B_PC + j) * 64 + lane;
    const int px = idx / (BN / 4), cq = idx % (BN / 4);
    b_colok[j] = n0 + 4 * cq < a.Cout;
    b_P[j] = step0 * BKP + px;
    b_off[j] = b_P[j] * a.ldd + n0 + 4 * cq;
  }
  const float* zero = ssg_zero_page_w;

  auto issue = [&](int s) {
    float* st = lds + (s % NSTAGE) * STAGE;
#pragma unroll
    for (int j = 0; j < A_PC; ++j) {
      const int iy = a_gy[j] * a.in_sy + a_dy[j], ix = a_gx[j] * a.in_sx + a_dx[j];
      const bool ok = a_rowok[j] && a_n[j] < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const float* p = ok ? a_base[j] + ((size_t)(a_n[j] * a.H + iy) * a.W + ix) * a_ld[j] : zero;
      dma16(p, st + (wave * A_PC + j) * 256);
      // advance this lane's pixel by one K-step (16 pixels)
      a_gx[j] += BKP;
      while (a_gx[j] >= a.GW) { a_gx[j] -= a.GW; if (++a_gy[j] >= a.GH) { a_gy[j] = 0; ++a_n[j]; } }
    }
#pragma unroll
    for (int j = 0; j < B_PC; ++j) {
      const float* p = (b_colok[j] && b_P[j] < a.Ptot) ? a.dout + b_off[j] : zero;
      dma16(p, st + BKP * BM + (wave * B_PC + j) * 256);
      b_P[j] += BKP; b_off[j] += (long long)BKP * a.ldd;
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nsteps > 0) issue(0);
  if (nsteps > 1) issue(1);
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_PC + B_PC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 2 < nsteps) issue(s + 2);
    const float* st = lds + (s % NSTAGE) * STAGE;
    const float* Ab = st + half * BM + wm * WTM + l31;
    const float* Bb = st + BKP * BM + half * BN + wn * WTN + l31;
#pragma unroll
    for (int kk = 0; kk < BKP / 2; ++kk) {
      float fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = Ab[2 * kk * BM + i * 32];
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = Bb[2 * kk * BN + j * 32];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }

  float* slab = a.ws + (size_t)blockIdx.z * a.M * a.Cout;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int co = n0 + wn * WTN + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < a.M && co < a.Cout) slab[(size_t)row * a.Cout + co] = acc[i][j][r];
      }
    }
}

}  // namespace

int ssg_wgrad_dma_launch(const WgArgs& a, int variant, dim3 grid, hipStream_t st) {
  if (variant == 0) hipLaunchKernelGGL((wgrad_dma_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wgrad_dma_kernel<128, 64, 2, 2>), grid, dim3(256), 0, st, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
