// Implicit-GEMM convolution, LDS-DMA pipeline (gfx950).  Same contract as conv_igemm.hip's kernel
// (ssg_conv2d_igemm_f32, kmode 0 only: Cin % 16 == 0); selected for the dense layers.
//
// What changes against the register-staged kernel:
//   * A and B tiles go global -> LDS directly with `global_load_lds_dwordx4` (one 1-KiB piece =
//     16 rows x 64 B per wave-instruction): no staging VGPRs, no ds_write pass, no vmcnt(0)
//     before a write pass, ~6 VALU per piece instead of ~35 for the predicated register loads.
//   * the LDS image is the DMA's lane-linear one ([row][16 floats], 64-B rows, no padding);
//     bank conflicts of the one-row-per-lane ds_read_b128 are removed by an XOR swizzle applied
//     on the SOURCE address (16-B position p of row r holds channel quad p ^ ((r>>2)&3)) and
//     mirrored on the read.
//   * out-of-image taps and rows beyond Cout read a zero page (a __device__ array), so the DMA
//     needs no predication.
//   * 3 LDS stages, two K-steps in flight across the single barrier per step:
//       wait vmcnt(pieces of 1 step) ; s_barrier ; issue step s+2 ; 32 MFMAs on step s.
//     The barrier both publishes step s (every wave has waited for its own pieces) and proves
//     stage (s+2)%3 == (s-1)%3 is no longer read.
#include "common.h"
#include "lds_dma.h"
#include "conv_args.h"

#ifndef SSG_EXPERIMENT
#define SSG_EXPERIMENT 0
#endif

namespace {

__device__ __attribute__((aligned(64))) float ssg_zero_page[64];


#ifndef SSG_DMA_STAGES
#define SSG_DMA_STAGES 3
#endif
constexpr int NSTAGE = SSG_DMA_STAGES;      // LDS stages; NSTAGE-1 K-steps are in flight across each barrier
constexpr int AHEAD = NSTAGE - 1;

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_dma_kernel(const ConvArgs a) {
  constexpr int TH = BM / 16;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int A_PC = BM / 64;          // A pieces (16 rows each) per wave per K-step
  constexpr int B_PC = BN / 64;          // B pieces per wave per K-step
  constexpr int STAGE = (BM + BN) * 16;  // floats per stage
  static_assert(B_PC >= 1, "BN >= 64");

  extern __shared__ __attribute__((aligned(1024))) float lds[];      // NSTAGE * STAGE floats

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // Workgroup -> tile map.  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs
  // (one L2 each); id -> (id % 8) * (nwg / 8) + id / 8 hands every XCD one contiguous run of tiles, and
  // the Cout tile is the fastest index, so the workgroups that are resident together on an XCD share
  // their input rows (halo and Cout re-reads hit its L2).  The ragged tail keeps the identity map.
  int bid = blockIdx.x;
  if (a.xcd_swizzle) {
    const int per = (int)gridDim.x >> 3;
    if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
  }
  const int nyt = a.ntiles_n;
  const int n0 = (bid % nyt) * BN; bid /= nyt;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int n = bid / a.tiles_y;

  // ---- per-lane DMA source state.  Piece j of this wave covers tile rows [(wave*PC + j)*16, +16);
  // lane l writes row r = base + (l>>2), 16-B position p = l&3, i.e. channel quad q = p ^ ((r>>2)&3).
  const int lr = lane >> 2, lp = lane & 3;
  int a_iy0[A_PC], a_ix0[A_PC];
  int a_q[A_PC];
#pragma unroll
  for (int j = 0; j < A_PC; ++j) {
    const int r = (wave * A_PC + j) * 16 + lr;
    const int gy = ty * TH + (r >> 4), gx = tx * 16 + (r & 15);
    const bool ok = (gy < a.GH) && (gx < a.GW);
    a_iy0[j] = ok ? gy * a.in_sy : -100000;
    a_ix0[j] = gx * a.in_sx;
    a_q[j] = 4 * (lp ^ ((r >> 2) & 3));
  }
  const float* b_src[B_PC];
#pragma unroll
  for (int j = 0; j < B_PC; ++j) {
    const int r = (wave * B_PC + j) * 16 + lr;
    const int q = 4 * (lp ^ ((r >> 2) & 3));
    b_src[j] = (n0 + r < a.Cout) ? a.w + (size_t)(n0 + r) * a.Kp + q : nullptr;
  }
  const float* zero = ssg_zero_page;

  auto issue = [&](int s) {
    float* st = lds + (s % NSTAGE) * STAGE;
    const int chunk = s / a.ntaps;
    const int t = s - chunk * a.ntaps;
    const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
    const int dy = (tb & 7) - 2, dx = (tb >> 3) - 2;
    const int c0 = chunk * 16;
    const float* src; int ld, cc;
    if (c0 < a.C1) { src = a.in1; ld = a.ld1; cc = c0; } else { src = a.in2; ld = a.ld2; cc = c0 - a.C1; }
#if SSG_EXPERIMENT != 3
#pragma unroll
    for (int j = 0; j < A_PC; ++j) {
      const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const float* p = ok ? src + ((size_t)(n * a.H + iy) * a.W + ix) * ld + cc + a_q[j] : zero;
      dma16(p, st + (wave * A_PC + j) * 256);
    }
#endif
#if SSG_EXPERIMENT != 4
#pragma unroll
    for (int j = 0; j < B_PC; ++j) {
      const float* p = b_src[j] ? b_src[j] + (size_t)s * 16 : zero;
      dma16(p, st + BM * 16 + (wave * B_PC + j) * 256);
    }
#endif
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int half = lane >> 5, l31 = lane & 31;
  const int sw = (l31 >> 2) & 3;
  // float offsets of the four channel quads of this lane's row inside a 16-float LDS row
  const int qoff0 = 4 * ((0 + half) ^ sw), qoff1 = 4 * ((2 + half) ^ sw);

  const int nsteps = a.nsteps;
#pragma unroll
  for (int p = 0; p < AHEAD; ++p)
    if (p < nsteps) issue(p);

  for (int s = 0; s < nsteps; ++s) {
    // steps s+1 .. s+AHEAD-1 may stay in flight; near the tail fewer are outstanding
    const int rem = nsteps - 1 - s;
#if SSG_EXPERIMENT == 3 || SSG_EXPERIMENT == 4
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (false)
#endif
    if (rem >= AHEAD - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * (A_PC + B_PC)) : "memory");
    else if (AHEAD >= 3 && rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_PC + B_PC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + AHEAD < nsteps) issue(s + AHEAD);
    const float* st = lds + (s % NSTAGE) * STAGE;
    const float* Ab = st + (wm * WTM + l31) * 16;
    const float* Bb = st + BM * 16 + (wn * WTN + l31) * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int qo = h == 0 ? qoff0 : qoff1;
      f32x4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *(const f32x4*)(Ab + i * 32 * 16 + qo);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = *(const f32x4*)(Bb + j * 32 * 16 + qo);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue (identical to conv_igemm.hip): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool want_bn = a.bnpart != nullptr;
  if (want_bn) ssg_bnpart_begin();
  // Column sums for the batch-norm statistics.  var = E[x^2] - mean^2 cancels, so plain fp32 partials lose it once |mean| >> std
  // (ADVICE r2).  Each lane sums DEVIATIONS from a pivot (its first output of the column) in fp32 -- their rounding error is
  // relative to |v - pivot|, not |v| -- and converts to sums of v in fp64 once per column: S1 = s1 + n*c, S2 = s2 + 2*c*s1 + n*c^2.
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = n0 + wn * WTN + j * 32 + l31;
    const bool cok = co < a.Cout;
    const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
    float s1 = 0.f, s2 = 0.f; int nv = 0;
    const float piv = acc[0][j][0] + bv;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int gy = ty * TH + (p >> 4), gx = tx * 16 + (p & 15);
        if (gy < a.GH && gx < a.GW) {
          const size_t pix = ((size_t)(n * a.OH + gy * a.out_sy + a.out_oy) * a.OW + gx * a.out_sx + a.out_ox);
          float v = acc[i][j][r] + bv;
          if (want_bn) { const float dv = v - piv; s1 += dv; s2 += dv * dv; ++nv; }
          if (cok) {
            if (a.res) v += a.res[pix * a.ldr + co];
            if (a.act == SSG_ACT_RELU) v = v < 0.f ? 0.f : v;
            else if (a.act == SSG_ACT_LRELU) v = v > 0.f ? v : v * a.slope;
            a.out[pix * a.ldo + co] = v;
          } else if (co < ((a.Cout + 3) & ~3)) {
            a.out[pix * a.ldo + co] = 0.f;
          }
        }
      }
    }
    if (want_bn) ssg_bnpart_put<BN, WTN>(lds, j, s1, s2, piv, nv, wm, wn, half, l31);
  }
  if (want_bn) ssg_bnpart_finish<NI, WAVES_M, BN, WTN>(a, lds, (n * a.tiles_y + ty) * a.tiles_x + tx, n0, wm, wn, half, l31);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  constexpr int TH = BM / 16;
  a.tiles_x = (a.GW + 15) / 16;
  a.tiles_y = (a.GH + TH - 1) / TH;
  static const int swz = [] { const char* e = getenv("SSG_XCD_SWIZZLE"); return e ? atoi(e) : 1; }();
  a.xcd_swizzle = swz;
  a.ntiles_n = (a.Cout + BN - 1) / BN;
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N * a.ntiles_n));
  constexpr int lds_bytes = NSTAGE * (BM + BN) * 16 * (int)sizeof(float);
  if (lds_bytes > 64 * 1024) {
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<BM, BN, WAVES_M, WAVES_N>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (attr != hipSuccess) { ssg_set_error("conv dma: LDS attribute: %s", hipGetErrorString(attr)); return (int)attr; }
  }
  hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, BN, WAVES_M, WAVES_N>), grid, dim3(256), lds_bytes, st, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

}  // namespace

// Called by ssg_conv2d_igemm_f32 (conv_igemm.hip) for kmode 0, Cout > 32.
// (A <256,128> tile with 128 x 64 per wave was measured: 284 VGPRs -> one wave per SIMD, 10-50% slower.)
// 0 = <128,128>, 1 = <256,64>, 2 = <128,64>.  Short K loops (1x1 convs, the 1/2/2/4-tap parity launches of a
// stride-2 input gradient): a tile is mostly prologue and epilogue, and four small workgroups per CU overlap those
// better than two or three large ones (measured: <= 16 steps -1.4 ms/step, 36 or 72 no better).
int ssg_conv_dma_variant(const ConvArgs& a, int variant) {
  static const int small_k = [] { const char* e = getenv("SSG_DMA_SMALLK"); return e ? atoi(e) : 16; }();
  if (small_k && a.nsteps <= small_k) return 2;
  return variant == 0 ? 0 : 1;
}

int ssg_conv_igemm_dma_launch(const ConvArgs& a, int variant, hipStream_t st) {
  switch (ssg_conv_dma_variant(a, variant)) {
    case 0: return launch<128, 128, 2, 2>(a, st);
    case 2: return launch<128, 64, 2, 2>(a, st);
    default: return launch<256, 64, 4, 1>(a, st);
  }
}
