// Weight gradient of 3x3 stride-1 convolutions with an LDS-resident pixel window (gfx950).  Same
// contract and slab layout as wgrad_dma_kernel (conv_wgrad_dma.hip); selected when the taps are the 9
// positions of a 3x3 window, the grid equals the image and the channel counts are multiples of CB.
//
// wgrad_dma_kernel's M tile is 128 rows of (tap, channel): every tap brings its own shifted copy of
// the 16 pixels of a K-step.  Here the M tile is [9 taps] x [CB channels]: one K-step (16 pixels of
// an image row) needs the 3 x 18 pixel window around it ONCE, and tap (dy,dx) is an LDS address offset.
//   * per step and per M row the A traffic falls 2.7x (54 vs 144 pixel rows), the B traffic and the
//     number of barriers per FLOP 2.25x (288 = 9 x 32 M rows per 32 channels instead of 128);
//   * each wave owns 32 channels x 9 taps x 32 output channels: 9 accumulators, per pixel pair one
//     B read + nine A reads (ds_read_b32, lanes = consecutive channels: conflict free) for nine MFMAs;
//   * A window image [54 pixels][CB] and B image [16 pixels][BN] are the DMA's lane-linear layouts;
//     3 stages, two K-steps in flight across the one barrier per step; out-of-image window pixels and
//     the ragged end of an image row read a zero page.
#include "common.h"
#include "lds_dma.h"
#include "conv_wgrad_args.h"
#include "conv_slow.h"

namespace {

__device__ __attribute__((aligned(64))) float ssg_zero_page_wh[64];


constexpr int BKP = 16;          // pixels per K-step (one run inside an image row)
constexpr int WW = BKP + 2;      // window width
constexpr int WPX = 3 * WW;      // window pixels
constexpr int NSTAGE = 3;

template <int CB, int BN>
__global__ __launch_bounds__(256) void wgrad_halo_kernel(const WgArgs a) {
  constexpr int WAVES_N = BN / 32, WAVES_M = 4 / WAVES_N;
  static_assert(WAVES_M * 32 == CB, "one 32-channel fragment per wave row");
  constexpr int AP = (WPX * CB + 255) / 256;     // 1-KiB pieces of the A window
  constexpr int APW = (AP + 3) / 4;              // per wave (the image is padded to 4*APW pieces)
  constexpr int B_PC = BKP * BN / 256 / 4;
  static_assert(B_PC >= 1, "BN >= 64");
  constexpr int ASZ = 4 * APW * 256;             // floats
  constexpr int STAGE = ASZ + BKP * BN;

  extern __shared__ __attribute__((aligned(1024))) float lds[];      // NSTAGE * STAGE floats

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int half = lane >> 5, l31 = lane & 31;
  const int c0 = blockIdx.x * CB, n0 = blockIdx.y * BN;
  const int Cin = a.C1 + a.C2;
  const int XB = (a.GW + BKP - 1) / BKP;

  const int step0 = (int)blockIdx.z * a.steps_per_split;
  const int total = a.N * a.GH * XB;
  int nsteps = total - step0;
  if (nsteps > a.steps_per_split) nsteps = a.steps_per_split;
  if (nsteps < 0) nsteps = 0;

  // wave-uniform coordinate of the next step to issue.  Steps walk DOWN a 16-pixel-wide column strip
  // (gy fastest, then the strip, then the image): consecutive steps share two of their three window rows,
  // so the 3x row re-read of the window is served by L2 instead of HBM (measured before: 2.5x the
  // algorithmic bytes with x-fastest order).
  int is_gy = step0 % a.GH, is_col = step0 / a.GH;
  int is_xb = is_col % XB, is_n = is_col / XB;

  // ---- A window pieces: lane -> (window pixel, channel quad)
  const float* a_src; int a_ld;
  if (c0 < a.C1) { a_src = a.in1 + c0; a_ld = a.ld1; } else { a_src = a.in2 + (c0 - a.C1); a_ld = a.ld2; }
  int a_wy[APW], a_wx[APW], a_cq[APW]; bool a_ok[APW];
#pragma unroll
  for (int k = 0; k < APW; ++k) {
    const int idx = (wave + 4 * k) * 64 + lane;          // float4 index in the [WPX][CB/4] image
    const int wp = idx / (CB / 4);
    a_cq[k] = 4 * (idx % (CB / 4));
    a_ok[k] = wp < WPX;
    a_wy[k] = wp / WW - 1; a_wx[k] = wp % WW - 1;
  }
  // ---- B pieces: lane -> (pixel within step, column quad)
  int b_px[B_PC], b_cq[B_PC]; bool b_colok[B_PC];
#pragma unroll
  for (int j = 0; j < B_PC; ++j) {
    const int idx = (wave * B_PC + j) * 64 + lane;
    b_px[j] = idx / (BN / 4); b_cq[j] = 4 * (idx % (BN / 4));
    b_colok[j] = n0 + b_cq[j] < a.Cout;
  }
  const float* zero = ssg_zero_page_wh;

  auto issue = [&](int s) {
    float* st = lds + (s % NSTAGE) * STAGE;
    const int gx0 = is_xb * BKP;
#pragma unroll
    for (int k = 0; k < APW; ++k) {
      const int iy = is_gy + a_wy[k], ix = gx0 + a_wx[k];
      const bool ok = a_ok[k] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const float* p = ok ? a_src + ((size_t)(is_n * a.H + iy) * a.W + ix) * a_ld + a_cq[k] : zero;
      dma16(p, st + (wave + 4 * k) * 256);
    }
#pragma unroll
    for (int j = 0; j < B_PC; ++j) {
      const int gx = gx0 + b_px[j];
      const float* p = (b_colok[j] && gx < a.GW) ? a.dout + ((size_t)(is_n * a.GH + is_gy) * a.GW + gx) * a.ldd + n0 + b_cq[j] : zero;
      dma16(p, st + ASZ + (wave * B_PC + j) * 256);
    }
    if (++is_gy == a.GH) { is_gy = 0; if (++is_xb == XB) { is_xb = 0; ++is_n; } }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // tap t = (dy+1)*3 + (dx+1) (checked on the host) reads window pixel (dy+1)*WW + (px + dx + 1): the LDS
  // offsets are compile-time constants, so pairs of reads fuse into ds_read2 and need no address arithmetic

  if (nsteps > 0) issue(0);
  if (nsteps > 1) issue(1);
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APW + B_PC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 2 < nsteps) issue(s + 2);
    const float* st = lds + (s % NSTAGE) * STAGE;
    const float* Aw = st + half * CB + wm * 32 + l31;
    const float* Bb = st + ASZ + half * BN + wn * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < BKP / 2; ++kk) {
      const float fb = Bb[2 * kk * BN];
      float fa[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) fa[t] = Aw[((t / 3) * WW + (t % 3) + 2 * kk) * CB];
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb, acc[t], 0, 0, 0);
    }
  }

  // slab layout of conv_wgrad.hip: [split][row = t*Cin + c][Cout]
  float* slab = a.ws + (size_t)blockIdx.z * a.M * a.Cout;
  const int co = n0 + wn * 32 + l31;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (c < Cin && co < a.Cout) slab[((size_t)t * Cin + c) * a.Cout + co] = acc[t][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same weight gradient with both fp32 operands split into three bf16 terms on the bf16 matrix pipe (see
// conv_igemm_halo_x3.hip for the arithmetic: x = x1 + x2 + x3, the six products of order >= 2^-16, fp32 accumulation).
// Same tiles, same LDS images, same DMA ring and slab layout as wgrad_halo_kernel; what changes is the inner product:
//   * one v_mfma_f32_32x32x16_bf16 covers the 16 pixels of a K-step: lane (row = channel, k-half h) needs 8 CONSECUTIVE
//     pixels of its channel, 8h..8h+7 shifted by the tap's dx.  Per window row a lane reads the 10 pixels 8h..8h+9 once
//     (ds_read_b32, lanes = consecutive channels: conflict free) and splits them once; the three taps of the row are the
//     three 8-pixel sub-windows.  The bf16 terms are produced already PACKED by v_cvt_pk_bf16_f32 in both pair alignments
//     (even pairs (0,1)...(8,9) serve dx = -1 / +1, odd pairs (1,2)...(7,8) serve dx = 0): no repacking.
//   * per step and wave: 38 ds_read_b32, ~210 VALU, 54 MFMAs (1728 cycles; the fp32 kernel: 72 MFMAs, 4608 cycles).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x2 cvt2(f32x2 v) { return __builtin_convertvector(v, bf16x2); }                 // v_cvt_pk_bf16_f32 (RNE)
__device__ __forceinline__ f32x2 up2(bf16x2 v) { return __builtin_convertvector(v, f32x2); }
__device__ __forceinline__ bf16x8 cat4(bf16x2 a, bf16x2 b, bf16x2 c, bf16x2 d) {
  const bf16x4 lo = __builtin_shufflevector(a, b, 0, 1, 2, 3), hi = __builtin_shufflevector(c, d, 0, 1, 2, 3);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// NV consecutive values (NV even) -> the three bf16 terms as even pairs E[k][i] = (2i, 2i+1) and, with ODD, odd pairs
// O[k][i] = (2i+1, 2i+2)
template <int NV, bool ODD>
__device__ __forceinline__ void split_pairs(const float (&v)[NV], bf16x2 (&E)[3][NV / 2], bf16x2 (&O)[3][NV / 2 - 1]) {
  f32x2 r[NV / 2];
#pragma unroll
  for (int i = 0; i < NV / 2; ++i) r[i] = f32x2{v[2 * i], v[2 * i + 1]};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) E[k][i] = cvt2(r[i]);
    if (ODD) {
#pragma unroll
      for (int i = 0; i < NV / 2 - 1; ++i) O[k][i] = cvt2(f32x2{r[i][1], r[i + 1][0]});
    }
    if (k < 2) {
#pragma unroll
      for (int i = 0; i < NV / 2; ++i) r[i] = r[i] - up2(E[k][i]);        // exact residual
    }
  }
}

template <int CB, int BN>
__global__ __launch_bounds__(256, 2) void wgrad_halo_x3_kernel(const WgArgs a) {
  constexpr int WAVES_N = BN / 32, WAVES_M = 4 / WAVES_N;
  static_assert(WAVES_M * 32 == CB, "one 32-channel fragment per wave row");
  constexpr int AP = (WPX * CB + 255) / 256;
  constexpr int APW = (AP + 3) / 4;
  constexpr int B_PC = BKP * BN / 256 / 4;
  static_assert(B_PC >= 1, "BN >= 64");
  constexpr int ASZ = 4 * APW * 256;
  constexpr int STAGE = ASZ + BKP * BN;

  extern __shared__ __attribute__((aligned(1024))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int half = lane >> 5, l31 = lane & 31;
  const int c0 = blockIdx.x * CB, n0 = blockIdx.y * BN;
  const int Cin = a.C1 + a.C2;
  const int XB = (a.GW + BKP - 1) / BKP;

  const int step0 = (int)blockIdx.z * a.steps_per_split;
  const int total = a.N * a.GH * XB;
  int nsteps = total - step0;
  if (nsteps > a.steps_per_split) nsteps = a.steps_per_split;
  if (nsteps < 0) nsteps = 0;

  int is_gy = step0 % a.GH, is_col = step0 / a.GH;
  int is_xb = is_col % XB, is_n = is_col / XB;

  const float* a_src; int a_ld;
  if (c0 < a.C1) { a_src = a.in1 + c0; a_ld = a.ld1; } else { a_src = a.in2 + (c0 - a.C1); a_ld = a.ld2; }
  int a_wy[APW], a_wx[APW], a_cq[APW]; bool a_ok[APW];
#pragma unroll
  for (int k = 0; k < APW; ++k) {
    const int idx = (wave + 4 * k) * 64 + lane;
    const int wp = idx / (CB / 4);
    a_cq[k] = 4 * (idx % (CB / 4));
    a_ok[k] = wp < WPX;
    a_wy[k] = wp / WW - 1; a_wx[k] = wp % WW - 1;
  }
  int b_px[B_PC], b_cq[B_PC]; bool b_colok[B_PC];
#pragma unroll
  for (int j = 0; j < B_PC; ++j) {
    const int idx = (wave * B_PC + j) * 64 + lane;
    b_px[j] = idx / (BN / 4); b_cq[j] = 4 * (idx % (BN / 4));
    b_colok[j] = n0 + b_cq[j] < a.Cout;
  }
  const float* zero = ssg_zero_page_wh;

  auto issue = [&](int s) {
    float* st = lds + (s % NSTAGE) * STAGE;
    const int gx0 = is_xb * BKP;
#pragma unroll
    for (int k = 0; k < APW; ++k) {
      const int iy = is_gy + a_wy[k], ix = gx0 + a_wx[k];
      const bool ok = a_ok[k] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const float* p = ok ? a_src + ((size_t)(is_n * a.H + iy) * a.W + ix) * a_ld + a_cq[k] : zero;
      dma16(p, st + (wave + 4 * k) * 256);
    }
#pragma unroll
    for (int j = 0; j < B_PC; ++j) {
      const int gx = gx0 + b_px[j];
      const float* p = (b_colok[j] && gx < a.GW) ? a.dout + ((size_t)(is_n * a.GH + is_gy) * a.GW + gx) * a.ldd + n0 + b_cq[j] : zero;
      dma16(p, st + ASZ + (wave * B_PC + j) * 256);
    }
    if (++is_gy == a.GH) { is_gy = 0; if (++is_xb == XB) { is_xb = 0; ++is_n; } }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (nsteps > 0) issue(0);
  if (nsteps > 1) issue(1);
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APW + B_PC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 2 < nsteps) issue(s + 2);
    const float* st = lds + (s % NSTAGE) * STAGE;
    const float* Aw = st + wm * 32 + l31 + 8 * half * CB;            // window pixel 8h of row 0, this lane's channel
    const float* Bb = st + ASZ + wn * 32 + l31 + 8 * half * BN;      // pixel 8h, this lane's output channel
    // dout fragment: 8 pixels -> 3 planes
    bf16x8 b1, b2, b3;
    {
      float bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = Bb[e * BN];
      bf16x2 E[3][4], O[3][3];
      split_pairs<8, false>(bv, E, O);
      b1 = cat4(E[0][0], E[0][1], E[0][2], E[0][3]); b2 = cat4(E[1][0], E[1][1], E[1][2], E[1][3]); b3 = cat4(E[2][0], E[2][1], E[2][2], E[2][3]);
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      float av[10];
#pragma unroll
      for (int e = 0; e < 10; ++e) av[e] = Aw[(dy * WW + e) * CB];
      bf16x2 E[3][5], O[3][4];
      split_pairs<10, true>(av, E, O);
      bf16x8 a1[3], a2[3], a3[3];                                    // [dx]: dx = 0 / 2 from the even pairs, dx = 1 from the odd pairs
      a1[0] = cat4(E[0][0], E[0][1], E[0][2], E[0][3]); a1[1] = cat4(O[0][0], O[0][1], O[0][2], O[0][3]); a1[2] = cat4(E[0][1], E[0][2], E[0][3], E[0][4]);
      a2[0] = cat4(E[1][0], E[1][1], E[1][2], E[1][3]); a2[1] = cat4(O[1][0], O[1][1], O[1][2], O[1][3]); a2[2] = cat4(E[1][1], E[1][2], E[1][3], E[1][4]);
      a3[0] = cat4(E[2][0], E[2][1], E[2][2], E[2][3]); a3[1] = cat4(O[2][0], O[2][1], O[2][2], O[2][3]); a3[2] = cat4(E[2][1], E[2][2], E[2][3], E[2][4]);
#define SSG_WX3(A, B)                                                                                            \
  _Pragma("unroll") for (int dx = 0; dx < 3; ++dx)                                                              \
      acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[dx], B, acc[dy * 3 + dx], 0, 0, 0);
      SSG_WX3(a3, b1) SSG_WX3(a2, b2) SSG_WX3(a1, b3) SSG_WX3(a2, b1) SSG_WX3(a1, b2) SSG_WX3(a1, b1)
#undef SSG_WX3
    }
  }
  float* slab = a.ws + (size_t)blockIdx.z * a.M * a.Cout;
  const int co = n0 + wn * 32 + l31;
  {                                                      // non-finite operands: conv_slow.h
    bool bad = false;
#pragma unroll
    for (int t = 0; t < 9; ++t) bad |= ssg_nonfinite16(acc[t]);
    if (__builtin_amdgcn_readfirstlane(__syncthreads_or(bad))) {
      // the slab values are written by the slow path itself (144 accumulators per lane: refilling them would spill)
      const WgArgs& as = *ssg_reload_args<WgArgs>();
#pragma unroll 1
      for (int e = 0; e < 144; ++e) {
        const int t = e >> 4, r = e & 15;
        const int c = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float v = ssg_wgrad_slow_value_strips(as, t, c, co, step0, step0 + nsteps, BKP);
        if (c < Cin && co < as.Cout) slab[((size_t)t * Cin + c) * as.Cout + co] = v;
      }
      return;
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (c < Cin && co < a.Cout) slab[((size_t)t * Cin + c) * a.Cout + co] = acc[t][r];
    }
}

template <int CB, int BN>
int launch(const WgArgs& a, dim3 grid, hipStream_t st, bool split) {
  constexpr int AP = (WPX * CB + 255) / 256, APW = (AP + 3) / 4;
  constexpr int lds_bytes = NSTAGE * (4 * APW * 256 + BKP * BN) * (int)sizeof(float);
  static_assert(lds_bytes <= 64 * 1024, "LDS budget");
  if (split) hipLaunchKernelGGL((wgrad_halo_x3_kernel<CB, BN>), grid, dim3(256), lds_bytes, st, a);
  else hipLaunchKernelGGL((wgrad_halo_kernel<CB, BN>), grid, dim3(256), lds_bytes, st, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

}  // namespace

// variant 0: 32 channels x 128 output channels per workgroup (Cout > 64); 1: 64 x 64
int ssg_wgrad_halo_cb(int variant) { return variant == 0 ? 32 : 64; }

bool ssg_wgrad_halo_ok(const ssg_wgrad_desc* d, int variant) {
  if (d->ntaps != 9 || d->in_sy != 1 || d->in_sx != 1 || d->GH != d->H || d->GW != d->W) return false;
  unsigned seen = 0;
  for (int t = 0; t < 9; ++t) {
    if (d->dy[t] != t / 3 - 1 || d->dx[t] != t % 3 - 1) return false;      // row-major window order
    seen |= 1u << t;
  }
  if (seen != 0x1ffu) return false;
  const int cb = ssg_wgrad_halo_cb(variant);
  if (d->C1 % cb || d->C2 % cb) return false;
  if ((long long)d->N * d->GH * ((d->GW + BKP - 1) / BKP) >= (1ll << 31)) return false;
  return true;
}

int ssg_wgrad_halo_launch(const WgArgs& a, int variant, dim3 grid, hipStream_t st, bool split) {
  if (variant == 0) return launch<32, 128>(a, grid, st, split);
  return launch<64, 64>(a, grid, st, split);
}
