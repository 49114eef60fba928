// Weight-gradient implicit GEMM, LDS-DMA pipeline (gfx950).  Same contract and slab layout as
// wgrad_kernel in conv_wgrad.hip; selected for Cout > 32.
//   * [16 pixels][rows] LDS images are exactly the DMA's lane-linear layout (a 1-KiB piece = 2
//     pixels x 128 rows, or 4 pixels x 64 rows), read back with conflict-free ds_read_b32, so the
//     register staging, its ds_write pass and the padding disappear;
//   * 3 stages, two K-steps (2 x 16 pixels) in flight across the one barrier per step;
//   * shifted/out-of-image taps and the pixel tail read a zero page instead of being predicated;
//   * the per-lane pixel coordinate advances incrementally (no divisions in the loop).
// X3 = true: the same pipeline multiplying on the bf16 pipe with both fp32 operands split into three bf16 terms as they
// leave LDS (mfma_split.h; arithmetic and error as in conv_igemm_halo_x3.hip).  A K-step of 16 pixels is ONE
// v_mfma_f32_32x32x16_bf16 per term pair: a lane gathers its row's 8 pixels (8 ds_read_b32, as many as the fp32 form
// reads), splits them (~45 VALU per fragment) and issues 6 MFMAs per (M, N) fragment pair where the fp32 form issues 8.
#include "common.h"
#include "lds_dma.h"
#include "conv_wgrad_args.h"
#include "conv_slow.h"
#include "mfma_split.h"

namespace {

__device__ __attribute__((aligned(64))) float ssg_zero_page_w[64];


constexpr int BKP = 16;
constexpr int NSTAGE = 3;

template <int BM, int BN, int WAVES_M, int WAVES_N, bool X3>
__device__ __forceinline__ void wgrad_dma_body(const WgArgs& a) {
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
  // Workgroup -> (row block, column block, K slice).  The row blocks of one (column block, K slice) stream the SAME dy pixels
  // and the column blocks of one row block the same input pixels; linear workgroup ids go round-robin over the 8 XCDs, each
  // with its own L2, so the natural x-fastest order puts the sharers on eight different L2s and every one of them fetches the
  // shared operand from HBM (stride-2 3x3 layers: 9 row blocks -> dy read 9 times; PMC round 2: 3.3-4.6 TB/s).  Remap so that
  // the row blocks of a group follow one another on ONE XCD (groups dealt to the XCDs in turn); a tail of < 8 groups keeps
  // the natural order.
  int bx = blockIdx.x, g = blockIdx.y + gridDim.y * blockIdx.z;
  if (a.xcd_swizzle) {
    const int mt = gridDim.x, G = gridDim.y * gridDim.z;
    const int L = bx + mt * g;
    if (L < (G >> 3) * 8 * mt) {
      const int q = L >> 3;
      bx = q % mt; g = (q / mt) * 8 + (L & 7);
    }
  }
  const int by = g % (int)gridDim.y, bz = g / (int)gridDim.y;
  const int m0 = bx * BM, n0 = by * BN;
  const int Cin = a.C1 + a.C2;

  const long long step0 = (long long)bz * a.steps_per_split;
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
    if constexpr (X3) {
      // lane (row l31, k-half h) holds pixels 8h .. 8h + 7 of the step
      const float* Ab = st + 8 * half * BM + wm * WTM + l31;
      const float* Bb = st + BKP * BM + 8 * half * BN + wn * WTN + l31;
      bf16x8 a1[MI], a2[MI], a3[MI], b1[NI], b2[NI], b3[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        f32x4 u, v;
#pragma unroll
        for (int k = 0; k < 4; ++k) { u[k] = Ab[k * BM + i * 32]; v[k] = Ab[(4 + k) * BM + i * 32]; }
        split3(u, v, a1[i], a2[i], a3[i]);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        f32x4 u, v;
#pragma unroll
        for (int k = 0; k < 4; ++k) { u[k] = Bb[k * BN + j * 32]; v[k] = Bb[(4 + k) * BN + j * 32]; }
        split3(u, v, b1[j], b2[j], b3[j]);
      }
#define SSG_X3_TERM(A, B)                                                                           \
  _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                   \
  _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                   \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B[j], acc[i][j], 0, 0, 0);
      SSG_X3_TERM(a3, b1) SSG_X3_TERM(a2, b2) SSG_X3_TERM(a1, b3)
      SSG_X3_TERM(a2, b1) SSG_X3_TERM(a1, b2)
      SSG_X3_TERM(a1, b1)
#undef SSG_X3_TERM
    } else {
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
  }

  if constexpr (X3) {                                    // non-finite operands: conv_slow.h
    wait_lds_reads();
    bool bad = false;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) bad |= ssg_nonfinite16(acc[i][j]);
    if (__builtin_amdgcn_readfirstlane(__syncthreads_or(bad))) {     // scalar condition: a uniform branch, the accumulators are dead inside it
      const WgArgs& as = *ssg_reload_args<WgArgs>();
      const long long P0 = step0 * BKP, P1e = (step0 + nsteps) * BKP;
      const long long P1 = P1e < as.Ptot ? P1e : as.Ptot;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          ssg_slow_refill16(acc[i][j], lds + tid, 256, [&](int r) {
            const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int t = row / Cin;
            return ssg_wgrad_slow_value_flat(as, t, row - t * Cin, n0 + wn * WTN + j * 32 + l31, P0, P1);
          });
    }
  }
  float* slab = a.ws + (size_t)bz * a.M * a.Cout;
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

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(const WgArgs a) { wgrad_dma_body<BM, BN, WAVES_M, WAVES_N, false>(a); }

template <int BM, int BN>
__global__ __launch_bounds__(256) void wgrad_dma_x3_kernel(const WgArgs a) { wgrad_dma_body<BM, BN, 2, 2, true>(a); }

}  // namespace

int ssg_wgrad_dma_launch(const WgArgs& a0, int variant, dim3 grid, hipStream_t st, bool split) {
  WgArgs a = a0;
  static const int swz = [] { const char* e = getenv("SSG_XCD_SWIZZLE"); return e ? atoi(e) : 1; }();
  a.xcd_swizzle = swz;
  if (split) {
    if (variant == 0) hipLaunchKernelGGL((wgrad_dma_x3_kernel<128, 128>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad_dma_x3_kernel<128, 64>), grid, dim3(256), 0, st, a);
  } else {
    if (variant == 0) hipLaunchKernelGGL((wgrad_dma_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad_dma_kernel<128, 64, 2, 2>), grid, dim3(256), 0, st, a);
  }
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
