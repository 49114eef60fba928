// Thin convolutions (one side <= 4 channels) on v_mfma_f32_4x4x1_16B_f32 (gfx950).
//
//   D_b[i][j] += A[4b+i] * B[4b+j]   (16 independent 4x4 blocks b; D_b[i][j] sits in lane 4b+j, register i)
//
// thin-Cin  (in has 4 channels, any Cout):  A = w[co = co0 + lane][tap][c]  (36 registers per lane, loaded once),
//     B = in[pixel j + tap][c] broadcast over b -> after 36 MFMAs lane 4b+j holds out[pixel j][co0 + 4b .. +3]:
//     one 16-B store per lane, four complete 256-B pixel rows per wave.
// thin-Cout (Cout <= 8, Cin % 64 == 0):     the 16 blocks split K: block b owns channels 4b..4b+3 of a 64-channel
//     chunk.  A = w[co = i][tap][chunk*64 + 4b + e], B = in[pixel j + tap][chunk*64 + 4b + e] (one 16-B load per
//     lane per tap, four complete 256-B rows per wave); the 16 partial sums per output are folded by a
//     data-halving butterfly over the lane bits of b (fixed order -> bitwise reproducible).
// Both stream the wide tensor exactly once through 16-B buffer loads whose out-of-image taps are dropped by
// the descriptor's range check (offset 0xffffffff -> 0.0f): no branches, no selects, no LDS.
// Same descriptor contract as ssg_conv2d_igemm_f32 (bias, residual, activation, pad lanes written as 0).
#include "common.h"
#include "conv_thin.h"
#ifndef SSG_T4_EXP
#define SSG_T4_EXP 0      // 1 = no MFMAs, 2 = no output stores: ablation builds of thin4_cin; 3 = no MFMAs, 4 = cache-resident loads:
                          // of thin4_cout (tools/micro_thin_exp.py), never shipped
#endif
#include <stdlib.h>

namespace {

struct T4Args {
  const float* in; const float* w; const float* bias; const float* res; float* out;
  int C, ld, N, H, W, Kp, kmode, ldr, Cout, ldo, ntaps;
  int tapidx[9];             // tap index at window position (dy+1)*3 + (dx+1), or -1
  int act; float slope;
  int groups, waves_per_group, total_units, strips, ybands;
  int nt_store;              // thin-Cin: non-temporal output stores (output larger than the caches)
  float* gsave; int ldg;     // thin32 SPADE mode: where gamma (+ bias) goes (the backward needs it), pixel stride
};

constexpr int RH_CIN = 16;       // rows per work unit (thin-Cin): a 4-pixel-wide strip marched downwards
constexpr int RH_COUT = 8;       // rows per work unit (thin-Cout) = accumulator sets folded by one butterfly
constexpr unsigned OOB = 0xffffffffu;

__device__ __forceinline__ f32x4 ldbuf4(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  if (act == SSG_ACT_RELU) return v < 0.f ? 0.f : v;
  if (act == SSG_ACT_LRELU) return v > 0.f ? v : v * slope;
  return v;
}

// Both kernels march a 4-pixel-wide strip down the image with a ring of KS+1 input rows in registers:
// every input row is loaded once per strip (KS 16-B loads per lane: the pixel's own column and its
// neighbours), the load of row y+R+1 is in flight while row y is multiplied.

// ------------------------------------------------------------------ thin-Cin: C == 4
// PF = rows of input in flight ahead of the row being multiplied.  vmcnt retires in order and counts stores: with PF = 1 the
// wait for the one prefetched row also waits for the 1-KiB output store issued just before it, i.e. every row pays a full
// store round trip (the waves sat in s_waitcnt 55-60 % of their cycles, 2.5 TB/s).  With PF rows in flight the counted wait
// leaves the PF-1 youngest row loads AND the stores between them outstanding.  12 VGPRs per extra row (KS = 3): PF = 3 keeps
// the kernel at 3 waves per SIMD (164 <= 168 registers).
// HAS_RES: the residual is a compile-time property of the kernel.  As a runtime `if (a.res)` around a plain global load,
// hipcc branched around the load and put `s_waitcnt vmcnt(0)` in front of its use IN EVERY ROW, taken or not: that drained
// the just-issued output store and every prefetched row each iteration (the kernel ran at 2.3 TB/s whatever PF was).  The
// residual now comes through the buffer descriptor path (OOB offset instead of a branch), one row ahead, so the compiler
// emits counted waits only.
template <int KS, int PF, bool HAS_RES>
__global__ __launch_bounds__(256) void thin4_cin_kernel(const T4Args a) {
  constexpr int R = KS / 2, NT = KS * KS, RING = KS + PF;
  const int lane = threadIdx.x & 63, j = lane & 3, b = lane >> 2;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int cg = wid % a.groups, wg = wid / a.groups;
  if (wg >= a.waves_per_group) return;
  const unsigned npix = (unsigned)(a.N * a.H * a.W);
  const auto in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)(npix * (unsigned)a.ld * 4u), 0x00020000);
  const unsigned ldb = (unsigned)a.ld * 4u;
  const auto res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_RES ? a.res : a.in), 0,
                                                        (int)(npix * (unsigned)(HAS_RES ? a.ldr : a.ld) * 4u), 0x00020000);

  // A operand: this lane's output channel, all taps x 4 channels (kmode 1: k = t*4 + c)
  const int co = cg * 64 + lane;
  f32x4 wv[NT];
#pragma unroll
  for (int p = 0; p < NT; ++p) {
    const int t = a.tapidx[KS == 3 ? p : 4];
    wv[p] = (co < a.Cout && t >= 0) ? *(const f32x4*)(a.w + (size_t)co * a.Kp + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int cb = cg * 64 + 4 * b;                    // first of this lane's 4 output channels
  const bool st_ok = cb < ((a.Cout + 3) & ~3);
  f32x4 bv = {0.f, 0.f, 0.f, 0.f}, cmask;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    cmask[i] = (cb + i < a.Cout) ? 1.f : 0.f;                      // pad lanes [Cout, pad4) are written as 0
    if (a.bias) bv[i] = (cb + i < a.Cout) ? a.bias[cb + i] : 0.f;
  }
  // activation without branches: v < 0 ? (relu ? 0 : v * neg_slope) : v   (neg_slope = 1 for "none")
  const bool is_relu = a.act == SSG_ACT_RELU;
  const float neg_slope = a.act == SSG_ACT_LRELU ? a.slope : 1.f;
  f32x4 pend = {0.f, 0.f, 0.f, 0.f};
  float* pend_ptr = nullptr;

  for (int u = wg; u < a.total_units; u += a.waves_per_group) {
    const int xs = u % a.strips; const int r0 = u / a.strips;
    const int yb = r0 % a.ybands, n = r0 / a.ybands;
    const int x = xs * 4 + j;
    const int y0 = yb * RH_CIN;
    const int y1 = y0 + RH_CIN < a.H ? y0 + RH_CIN : a.H;
    unsigned coloff[KS];
#pragma unroll
    for (int e = 0; e < KS; ++e) {
      const int ix = x + e - R;
      coloff[e] = (unsigned)ix < (unsigned)a.W ? (unsigned)ix * ldb : OOB;
    }
    const unsigned imgoff = (unsigned)(n * a.H) * (unsigned)a.W * ldb;
    auto load_row = [&](f32x4* dst, int iy) {
      const bool rok = (unsigned)iy < (unsigned)a.H;
      const unsigned ro = imgoff + (unsigned)iy * (unsigned)a.W * ldb;
#pragma unroll
      for (int e = 0; e < KS; ++e) dst[e] = ldbuf4(in_rs, (rok && coloff[e] != OOB) ? ro + coloff[e] : OOB);
    };
    f32x4 v[RING][KS];
#pragma unroll
    for (int q = 0; q < KS + PF - 1; ++q) load_row(v[q], y0 - R + q);
    // residual of output row yy (this lane's pixel column, its 4 channels): one row ahead, ring-indexed like v
    f32x4 rvr[HAS_RES ? RING : 1];
    auto load_res = [&](int yy) -> f32x4 {
      const bool ok = x < a.W && st_ok && yy < y1;
      return ldbuf4(res_rs, ok ? ((unsigned)((n * a.H + yy) * a.W + x) * (unsigned)a.ldr + (unsigned)cb) * 4u : OOB);
    };
    if (HAS_RES) rvr[0] = load_res(y0);
    for (int y = y0; y < y1; y += RING) {
#pragma unroll
      for (int r = 0; r < RING; ++r) {
        if (y + r < y1) {
          // The previous row's result is stored here, a whole MFMA phase before the next s_waitcnt has to
          // cover it (gfx9 counts stores in vmcnt: a store issued right before the wait exposes its latency).
#if SSG_T4_EXP == 2
          if (pend_ptr && pend[0] == 123.456f) *(f32x4*)pend_ptr = pend;          // ablation: no output stores
#else
          // outputs far larger than the caches (the 1-GB gamma / beta buffers of the 512^2 level) are stored non-temporally:
          // 0.474 -> 0.402 ms on 16 x 512^2, 4 -> 64 (tools/micro_thin_exp.py)
          if (pend_ptr) { if (a.nt_store) __builtin_nontemporal_store(pend, (f32x4*)pend_ptr); else *(f32x4*)pend_ptr = pend; }
#endif
          // row (y + r) + R + PF goes into the slot of row (y + r) - R - 1, whose last reader was the previous iteration
          load_row(v[(r + KS + PF - 1) % RING], y + r - R + KS + PF - 1);
          const size_t pix = (size_t)(n * a.H + y + r) * a.W + x;
          f32x4 rv = {0.f, 0.f, 0.f, 0.f};
          if (HAS_RES) { rvr[(r + 1) % RING] = load_res(y + r + 1); rv = rvr[r]; }
          // two accumulator chains (even / odd k) keep dependent MFMAs apart
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#if SSG_T4_EXP == 1
          acc0 = v[r % RING][0] + v[(r + KS - 1) % RING][KS - 1];                  // ablation: no MFMAs (rows stay live)
#else
#pragma unroll
          for (int q = 0; q < KS; ++q)
#pragma unroll
            for (int e = 0; e < KS; ++e)
#pragma unroll
              for (int c = 0; c < 4; c += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[q * KS + e][c], v[(r + q) % RING][e][c], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[q * KS + e][c + 1], v[(r + q) % RING][e][c + 1], acc1, 0, 0, 0);
              }
#endif
          const f32x4 t = acc0 + acc1 + bv + rv * cmask;
#pragma unroll
          for (int i = 0; i < 4; ++i) pend[i] = (t[i] < 0.f ? (is_relu ? 0.f : t[i] * neg_slope) : t[i]) * cmask[i];
          pend_ptr = (x < a.W && st_ok) ? a.out + pix * a.ldo + cb : nullptr;
        }
      }
    }
  }
  if (pend_ptr) *(f32x4*)pend_ptr = pend;
}

// ------------------------------------------------------------------ thin-Cin on the 32x32x2 MFMA: C == 4, 3x3 window, Cout >= 32
// The 4x4x1 kernel above issues 36 MFMAs + ~90 other instructions per 1 KiB of output and is bound by instruction issue
// (tools/micro_thin_exp.py: 0.30 ms of its 0.40 ms at 16 x 512^2, 4 -> 64, remain with the stores compiled out).  K = 9 taps x
// 4 channels = 36 is exactly 18 K-steps of v_mfma_f32_32x32x2_f32, so the same FLOPs take 8x fewer MFMA instructions as a
// plain GEMM: M = 32 consecutive pixels of one image row (lane&31), N = 2 x 32 output channels, and the lane half (lane>>5)
// picks the channel pair {2h, 2h+1} of a tap -- one 8-byte load per (input row, dx) and lane, every input row loaded once per
// strip and kept in a register ring for the three output rows that use it (as above).  A K-step is (tap, channel j of the
// pair): A = in[y+dy][x+dx][2h+j], B = w[co][tap][2h+j] (36 registers per lane, loaded once).  The 32x32 accumulator tile
// leaves each lane with 16 pixels of ONE channel: 32 dword stores per output row whose lanes 0-31 / 32-63 each cover one
// whole 128-byte line (32 consecutive channels of a pixel).  All memory operations go through buffer descriptors (OOB
// offset instead of branches).
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 ldbuf2(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
}

constexpr int RH_K36 = 16;       // output rows per work unit

// MODE 0: out = act(conv + bias); 1: + residual; 2: SPADE (normalization.py:117-120) -- `w` holds [gamma rows 0..C-1 | beta rows
// C..2C-1] (a.Cout = 2C), the wave's two N-fragments are gamma and beta of the SAME 32 channels, `res` is the tensor being
// modulated: out = x * (1 + gamma) + beta, and gamma goes to a.gsave.  The [N,2C,H,W] gamma|beta tensor and the separate
// modulate pass (3 reads + 1 write of C channels) never exist.
// CIN = 4: the lane half h owns the channel pair {2h, 2h+1} of a tap (8-byte loads, 18 K-steps); CIN = 8 (SPADE at the 256^2
// level, nhidden = 8): the quad {4h .. 4h+3} (16-byte loads, 36 K-steps, 72 weight registers).
template <int MODE, int PF, bool NT, int CIN = 4>
__global__ __launch_bounds__(256) void thin32_cin_kernel(const T4Args a) {
  constexpr bool HAS_RES = MODE >= 1, SPADE = MODE == 2;
  constexpr int CH = CIN / 2;               // channels per lane half and tap
  typedef float vrow_t __attribute__((ext_vector_type(CH)));
  constexpr int AUX = NT ? 2 : 0;           // non-temporal stores for outputs far larger than the caches
  constexpr int RING = 3 + PF;
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int cg = wid % a.groups, wg = wid / a.groups;
  if (wg >= a.waves_per_group) return;
  const unsigned npix = (unsigned)(a.N * a.H * a.W);
  const auto in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)(npix * (unsigned)a.ld * 4u), 0x00020000);
  const auto out_rs = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(npix * (unsigned)a.ldo * 4u), 0x00020000);
  const auto res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_RES ? a.res : a.in), 0,
                                                        (int)(npix * (unsigned)(HAS_RES ? a.ldr : a.ld) * 4u), 0x00020000);
  const auto g_rs = __builtin_amdgcn_make_buffer_rsrc(SPADE ? a.gsave : a.out, 0, (int)(npix * (unsigned)(SPADE ? a.ldg : a.ldo) * 4u), 0x00020000);
  const unsigned ldb = (unsigned)a.ld * 4u, ldob = (unsigned)a.ldo * 4u, ldrb = (unsigned)a.ldr * 4u, ldgb = (unsigned)a.ldg * 4u;
  const int Cmod = a.Cout >> 1;            // SPADE: channels of x / out / gamma

  // B operand: K-step s = 2*p + j (p = window position, j = channel of the lane half's pair) for the lane's output channel
  float wB[2][9 * CH], bv[2], keep[2];       // keep = 0 for the pad lanes [Cout, pad4(Cout)): written as 0
  unsigned vo_out[2], vo_res[2];         // per-lane part of the byte offset (pixel 4*h of the strip, this lane's channel) or OOB
  unsigned vo_g = OOB;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int cm = cg * 32 + l31;          // SPADE: the modulated channel of this lane
    const int co = SPADE ? (cm < Cmod ? f * Cmod + cm : a.Cout) : cg * 64 + f * 32 + l31;
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      const int t = a.tapidx[p];
      const bool ok = co < a.Cout && t >= 0;
#pragma unroll
      for (int j = 0; j < CH; ++j) wB[f][CH * p + j] = ok ? a.w[(size_t)co * a.Kp + t * CIN + CH * h + j] : 0.f;
    }
    bv[f] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    keep[f] = co < a.Cout ? 1.f : 0.f;
    const bool cok = SPADE ? cm < Cmod : co < ((a.Cout + 3) & ~3);
    const int cch = SPADE ? cm : co;       // channel inside out / res rows
    vo_out[f] = cok ? 4u * (unsigned)h * ldob + (unsigned)cch * 4u : OOB;
    vo_res[f] = cok ? 4u * (unsigned)h * ldrb + (unsigned)cch * 4u : OOB;
    if (SPADE) vo_g = cok ? 4u * (unsigned)h * ldgb + (unsigned)cch * 4u : OOB;
  }
  const bool is_relu = a.act == SSG_ACT_RELU;
  const float neg_slope = a.act == SSG_ACT_LRELU ? a.slope : 1.f;

  for (int u = wg; u < a.total_units; u += a.waves_per_group) {
    const int xs = u % a.strips; const int r0 = u / a.strips;
    const int yb = r0 % a.ybands, n = r0 / a.ybands;
    const int x0 = xs * 32, x = x0 + l31;
    const int y0 = yb * RH_K36;
    const int y1 = y0 + RH_K36 < a.H ? y0 + RH_K36 : a.H;
    unsigned coloff[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int ix = x + e - 1;
      coloff[e] = (unsigned)ix < (unsigned)a.W ? (unsigned)ix * ldb + (unsigned)(4 * CH) * (unsigned)h : OOB;
    }
    const unsigned imgoff = (unsigned)(n * a.H) * (unsigned)a.W * ldb;
    auto load_row = [&](vrow_t* dst, int iy) {
      const bool rok = (unsigned)iy < (unsigned)a.H;
      const unsigned ro = imgoff + (unsigned)iy * (unsigned)a.W * ldb;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const unsigned off = (rok && coloff[e] != OOB) ? ro + coloff[e] : OOB;
        if constexpr (CIN == 4) dst[e] = ldbuf2(in_rs, off); else dst[e] = ldbuf4(in_rs, off);
      }
    };
    // accumulator register q holds pixel column (q&3) + 8*(q>>2) + 4*h of the strip; bit q of `inw`: that column is inside the image
    unsigned inw = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) inw |= (x0 + (q & 3) + 8 * (q >> 2) + 4 * h < a.W ? 1u : 0u) << q;
    const bool full = x0 + 32 <= a.W;
    vrow_t v[RING][3];
#pragma unroll
    for (int q = 0; q < 2 + PF; ++q) load_row(v[q], y0 - 1 + q);
    auto mma_row = [&](f32x16 (&acc)[2], int r) {        // output row whose window rows sit in ring slots r, r+1, r+2
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[f][q] = 0.f;
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
          for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int f = 0; f < 2; ++f)
              acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[(r + q) % RING][e][j], wB[f][CH * (q * 3 + e) + j], acc[f], 0, 0, 0);
    };
    auto store_row_full = [&](const f32x16 (&acc)[2], int yy) {       // whole strip inside the image: per-lane offset is
      const unsigned rowpix = (unsigned)((n * a.H + yy) * a.W + x0);  // loop-invariant, the rest rides the scalar offset
      float rv[2][16];
      const unsigned (&vo)[2] = vo_out; const unsigned (&vr)[2] = vo_res;
      if (HAS_RES) {
#pragma unroll
        for (int f = 0; f < (SPADE ? 1 : 2); ++f)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            rv[f][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                res_rs, vr[f], (int)((rowpix + (unsigned)((q & 3) + 8 * (q >> 2))) * ldrb), 0));
      }
      if (SPADE) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float g = acc[0][q] + bv[0], bt = acc[1][q] + bv[1];
          const unsigned pq = rowpix + (unsigned)((q & 3) + 8 * (q >> 2));
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g), g_rs, vo_g, (int)(pq * ldgb), AUX);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaf(rv[0][q], 1.f + g, bt)), out_rs, vo[0], (int)(pq * ldob), AUX);
        }
        return;
      }
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float t = acc[f][q] + bv[f];
          if (HAS_RES) t += rv[f][q];
          t = (t < 0.f ? (is_relu ? 0.f : t * neg_slope) : t) * keep[f];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), out_rs, vo[f],
                                                (int)((rowpix + (unsigned)((q & 3) + 8 * (q >> 2))) * ldob), AUX);
        }
    };
    auto store_row_edge = [&](const f32x16 (&acc)[2], int yy) {       // a branch per access instead of 32 live offset registers
      const unsigned rowpix = (unsigned)((n * a.H + yy) * a.W + x0);
      if (yy >= y1) return;
      if (SPADE) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          if ((inw >> q) & 1u) {
            const unsigned pq = rowpix + (unsigned)((q & 3) + 8 * (q >> 2));
            const float g = acc[0][q] + bv[0], bt = acc[1][q] + bv[1];
            const float xv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(res_rs, vo_res[0], (int)(pq * ldrb), 0));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g), g_rs, vo_g, (int)(pq * ldgb), AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaf(xv, 1.f + g, bt)), out_rs, vo_out[0], (int)(pq * ldob), AUX);
          }
        }
        return;
      }
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          if ((inw >> q) & 1u) {
            float t = acc[f][q] + bv[f];
            if (HAS_RES) t += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                res_rs, vo_res[f], (int)((rowpix + (unsigned)((q & 3) + 8 * (q >> 2))) * ldrb), 0));
            t = (t < 0.f ? (is_relu ? 0.f : t * neg_slope) : t) * keep[f];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), out_rs, vo_out[f],
                                                  (int)((rowpix + (unsigned)((q & 3) + 8 * (q >> 2))) * ldob), AUX);
          }
        }
    };
    // (Issuing the MFMAs of row y+1 before the epilogue of row y, interleaved by sched_group_barrier, was measured: no gain
    //  -- 0.319 vs 0.309 ms at 16 x 512^2, 4 -> 64 -- and 220-284 registers; the rows are processed one after the other.)
    f32x16 acc[2];
    for (int y = y0; y < y1; y += RING) {
#pragma unroll
      for (int r = 0; r < RING; ++r) {
        if (y + r < y1) {
          // input row y+r+1+PF goes into the slot of row y+r-2, whose last reader (output row y+r-1) is done
          load_row(v[(r + 2 + PF) % RING], y + r + 1 + PF);
          mma_row(acc, r);
          if (full) store_row_full(acc, y + r);
          else store_row_edge(acc, y + r);
        }
      }
    }
  }
}

// ------------------------------------------------------------------ tiny: C == 4 AND Cout <= 8 (SPADE's 3 -> nhidden conv and its
// input gradient at the 512^2 / 256^2 levels, normalization.py:92): 134 MB of traffic and 0.6 GFMA at 16 x 512^2.  One
// thread per pixel on the VALU: 9 coalesced 16-byte loads (1 KiB per wave and tap), 36 * Cout FMAs whose weight operand is a
// scalar register (uniform address -> s_load), one or two 16-byte stores.  The 4x4x1 kernel above spends a whole wave on 4
// pixels for these shapes (0.55 ms per launch; this one: the traffic floor).
template <int CO4>
__global__ __launch_bounds__(256) void tiny4_kernel(const T4Args a) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long npix = (long long)a.N * a.H * a.W;
  if (p >= npix) return;
  const int x = (int)(p % a.W); const long long r0 = p / a.W;
  const int y = (int)(r0 % a.H);
  const auto in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((unsigned)npix * (unsigned)a.ld * 4u), 0x00020000);
  const unsigned ldb = (unsigned)a.ld * 4u;
  float acc[CO4 * 4];
#pragma unroll
  for (int co = 0; co < CO4 * 4; ++co) acc[co] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
  f32x4 v[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {                            // all loads first, none behind a branch
    const int iy = y + q / 3 - 1, ix = x + q % 3 - 1;
    const bool ok = a.tapidx[q] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    v[q] = ldbuf4(in_rs, ok ? (unsigned)(p + (long long)(q / 3 - 1) * a.W + (q % 3 - 1)) * ldb : OOB);
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int t = a.tapidx[q];
    const float* wt = a.w + (t < 0 ? 0 : t) * 4;
    const float on = t < 0 ? 0.f : 1.f;                    // a window position without a tap: its loads came back as 0 already
#pragma unroll
    for (int co = 0; co < CO4 * 4; ++co) {
      const float* wr = wt + (size_t)(co < a.Cout ? co : 0) * a.Kp;
      const float m = co < a.Cout ? on : 0.f;
      acc[co] = fmaf(v[q][3], wr[3] * m, fmaf(v[q][2], wr[2] * m, fmaf(v[q][1], wr[1] * m, fmaf(v[q][0], wr[0] * m, acc[co]))));
    }
  }
#pragma unroll
  for (int g = 0; g < CO4; ++g) {
    if (4 * g < ((a.Cout + 3) & ~3)) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = 4 * g + e;
        float t = acc[co];
        if (a.res && co < a.Cout) t += a.res[p * a.ldr + co];
        o[e] = co < a.Cout ? act_apply(t, a.act, a.slope) : 0.f;
      }
      *(f32x4*)(a.out + p * a.ldo + 4 * g) = o;
    }
  }
}

// Fold the 16 K-blocks of one output-channel group (8 rows x 4 pixels x 4 channels per lane) and store: lane bit 5 (b bit 3)
// halves the rows 8 -> 4, bit 4: 4 -> 2, bit 3: 2 -> 1, bit 2 (b bit 0) halves the 4 channels -> 2; the kept half is chosen
// by the lane's own bit, so the order of the additions is fixed.
__device__ __forceinline__ void t4_fold_store(const T4Args& a, const f32x4* acc, int lane, int g, int n, int x, int y0, int my_s, int my_c,
                                              float bias0g, float bias1g) {
    // fold the 16 K-blocks: lane bit 5 (b bit 3) halves the rows 8 -> 4, bit 4: 4 -> 2, bit 3: 2 -> 1,
    // bit 2 (b bit 0) halves the 4 channels -> 2.  Kept half is chosen by the lane's own bit.
    float h4[4][4];
    {
      const bool hi = (lane >> 5) & 1;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float keep = hi ? acc[s + 4][i] : acc[s][i];
          const float send = hi ? acc[s][i] : acc[s + 4][i];
          h4[s][i] = keep + __shfl_xor(send, 32);
        }
    }
    float h2[2][4];
    {
      const bool hi = (lane >> 4) & 1;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float keep = hi ? h4[s + 2][i] : h4[s][i];
          const float send = hi ? h4[s][i] : h4[s + 2][i];
          h2[s][i] = keep + __shfl_xor(send, 16);
        }
    }
    float h1[4];
    {
      const bool hi = (lane >> 3) & 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float keep = hi ? h2[1][i] : h2[0][i];
        const float send = hi ? h2[0][i] : h2[1][i];
        h1[i] = keep + __shfl_xor(send, 8);
      }
    }
    float o0, o1;
    {
      const bool hi = (lane >> 2) & 1;
      const float k0 = hi ? h1[2] : h1[0], k1 = hi ? h1[3] : h1[1];
      const float s0 = hi ? h1[0] : h1[2], s1 = hi ? h1[1] : h1[3];
      o0 = k0 + __shfl_xor(s0, 4);
      o1 = k1 + __shfl_xor(s1, 4);
    }
    // lane bits: bit5 -> s bit 2, bit4 -> s bit 1, bit3 -> s bit 0, bit2 -> channel pair  == (b >> 1, b & 1)
    const int y = y0 + my_s;
    if (x < a.W && y < a.H) {
      const size_t pix = (size_t)(n * a.H + y) * a.W + x;
      const int ch = 4 * g + my_c;
      float t0 = o0 + bias0g, t1 = o1 + bias1g;
      if (a.res) {
        if (ch < a.Cout) t0 += a.res[pix * a.ldr + ch];
        if (ch + 1 < a.Cout) t1 += a.res[pix * a.ldr + ch + 1];
      }
      float2 o;
      o.x = ch < a.Cout ? act_apply(t0, a.act, a.slope) : 0.f;
      o.y = ch + 1 < a.Cout ? act_apply(t1, a.act, a.slope) : 0.f;
      if (ch < ((a.Cout + 3) & ~3)) *(float2*)(a.out + pix * a.ldo + ch) = o;
    }
}

// ------------------------------------------------------------------ thin-Cout: Cout <= 4*NG, C % 64 == 0
// NG = 2 (Cout 5..8, e.g. the 128 -> 8 gamma/beta input gradients) runs two A-operand sets over the same loaded rows.
template <int KS, int NG>
__global__ __launch_bounds__(256) void thin4_cout_kernel(const T4Args a) {
  constexpr int R = KS / 2, NT = KS * KS, S = RH_COUT, RING = KS + 1;
  const int lane = threadIdx.x & 63, j = lane & 3, b = lane >> 2;
  const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wg >= a.waves_per_group) return;
  const unsigned npix = (unsigned)(a.N * a.H * a.W);
  const auto in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)(npix * (unsigned)a.ld * 4u), 0x00020000);
  const unsigned ldb = (unsigned)a.ld * 4u;
  const int nchunks = a.C >> 6;
  // A operand row: lane 4b+i supplies w[co = 4g + i]
  // after the butterfly lane (b, j) owns outputs (row s = b >> 1, pixel j, channels 4g + 2*(b&1), +1)
  const int my_s = b >> 1, my_c = 2 * (b & 1);
  float bias0[NG], bias1[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    bias0[g] = (a.bias && 4 * g + my_c < a.Cout) ? a.bias[4 * g + my_c] : 0.f;
    bias1[g] = (a.bias && 4 * g + my_c + 1 < a.Cout) ? a.bias[4 * g + my_c + 1] : 0.f;
  }

  for (int u = wg; u < a.total_units; u += a.waves_per_group) {
    const int xs = u % a.strips; const int r0 = u / a.strips;
    const int yb = r0 % a.ybands, n = r0 / a.ybands;
    const int x = xs * 4 + j;
    const int y0 = yb * S;
    unsigned coloff[KS];
#pragma unroll
    for (int e = 0; e < KS; ++e) {
      const int ix = x + e - R;
      coloff[e] = (unsigned)ix < (unsigned)a.W ? (unsigned)ix * ldb + 16u * b : OOB;
    }
    const unsigned imgoff = (unsigned)(n * a.H) * (unsigned)a.W * ldb;
    f32x4 accg[NG][S];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int s = 0; s < S; ++s) accg[g][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nchunks; ++ch) {
      const int c0 = ch * 64 + 4 * b;                // this lane's 4 reduced channels
      f32x4 wv[NG][NT];
#pragma unroll
      for (int p = 0; p < NT; ++p) {
        const int t = a.tapidx[KS == 3 ? p : 4];
        const int k = a.kmode == 0 ? (c0 >> 4) * a.ntaps * 16 + t * 16 + (c0 & 15) : t * a.C + c0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int co = 4 * g + j;
          wv[g][p] = (co < a.Cout && t >= 0) ? *(const f32x4*)(a.w + (size_t)co * a.Kp + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      auto load_row = [&](f32x4* dst, int iy) {
        const bool rok = (unsigned)iy < (unsigned)a.H;
        const unsigned ro = imgoff + (unsigned)iy * (unsigned)a.W * ldb + (unsigned)ch * 256u;
#if SSG_T4_EXP == 4                                        // ablation: every row load reads the same cache-resident 1 KiB
#pragma unroll
        for (int e = 0; e < KS; ++e) dst[e] = ldbuf4(in_rs, (rok && coloff[e] != OOB) ? (unsigned)(lane * 16) : OOB);
#elif SSG_T4_EXP == 5                                      // ablation: only the dx = 0 load is real, dx = +-1 read that KiB
#pragma unroll
        for (int e = 0; e < KS; ++e) dst[e] = ldbuf4(in_rs, (rok && coloff[e] != OOB) ? (e == R ? ro + coloff[e] : (unsigned)(lane * 16)) : OOB);
#else
#pragma unroll
        for (int e = 0; e < KS; ++e) dst[e] = ldbuf4(in_rs, (rok && coloff[e] != OOB) ? ro + coloff[e] : OOB);
#endif
      };
      f32x4 v[RING][KS];
#pragma unroll
      for (int q = 0; q < KS; ++q) load_row(v[q], y0 - R + q);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        load_row(v[(s + KS) % RING], y0 + s - R + KS);
#pragma unroll
        for (int q = 0; q < KS; ++q)
#pragma unroll
          for (int e = 0; e < KS; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int g = 0; g < NG; ++g) {
#if SSG_T4_EXP == 3                                        // ablation: no MFMAs (loaded rows stay live through one add per value)
                accg[g][s][c] += v[(s + q) % RING][e][c];
#else
                accg[g][s] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[g][q * KS + e][c], v[(s + q) % RING][e][c], accg[g][s], 0, 0, 0);
#endif
              }
      }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) t4_fold_store(a, accg[g], lane, g, n, x, y0, my_s, my_c, bias0[g], bias1[g]);
  }
}

// (A streaming form of this kernel -- a wave walking a column of bands with all 10 rows of a body in a register ring, every row
//  requested a whole body before its first use -- was built and measured in round 2: 0.580 vs 0.585 ms at 16 x 512^2, 64 -> 3.
//  Ablations (tools/micro_thin_cout.py): no MFMAs 0.520, every load from one cache-resident KiB 0.319, only the dx = 0 loads
//  real 0.480: the time is issue (0.32) PLUS memory (0.16 unique + 0.10 for the overlapping dx loads), not latency.)

bool window_taps(const ssg_conv_desc* d, int* tapidx) {
  for (int p = 0; p < 9; ++p) tapidx[p] = -1;
  bool only_center = true;
  for (int t = 0; t < d->ntaps; ++t) {
    const int dy = d->dy[t], dx = d->dx[t];
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
    if (tapidx[(dy + 1) * 3 + dx + 1] >= 0) return false;
    tapidx[(dy + 1) * 3 + dx + 1] = t;
    if (dy || dx) only_center = false;
  }
  (void)only_center;
  return true;
}

int thin4_on() {
  static const int on = [] { const char* e = getenv("SSG_THIN4"); return e ? atoi(e) : 3; }();
  return on;
}

}  // namespace

// 0 = no; 3 = thin-Cin (in has 4 channels); 4 = thin-Cout (Cout <= 8, Cin % 64 == 0).  SSG_THIN4 is a bit
// mask (1 = thin-Cin, 2 = thin-Cout; default both).
int ssg_thin4_conv_kind(const ssg_conv_desc* d) {
  if (d->C2 != 0 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1 || d->out_oy || d->out_ox) return 0;
  if (d->GH != d->H || d->GW != d->W || d->OH != d->H || d->OW != d->W) return 0;
  if ((long long)d->N * d->H * d->W * d->ld1 >= (1ll << 30)) return 0;          // 32-bit byte offsets
  if (((uintptr_t)d->in1 & 15) || d->ld1 % 4 || ((uintptr_t)d->w & 15) || d->Kp % 4) return 0;
  int tapidx[9];
  if (!window_taps(d, tapidx)) return 0;
  if ((thin4_on() & 1) && d->C1 == 4 && d->kmode == 1 && d->ldo % 4 == 0 && !((uintptr_t)d->out & 15) &&
      (!d->res || (d->ldr % 4 == 0 && !((uintptr_t)d->res & 15) && (long long)d->N * d->H * d->W * d->ldr < (1ll << 30)))) return 3;
  if ((thin4_on() & 2) && d->Cout <= 8 && d->C1 >= 64 && d->C1 % 64 == 0 && d->ldo % 2 == 0 && !((uintptr_t)d->out & 7)) return 4;
  return 0;
}

static bool routes_tiny(const ssg_conv_desc* d, int kind) {
  static const int tiny = [] { const char* e = getenv("SSG_TINY4"); return e ? atoi(e) : 1; }();       // 0: 4x4x1 kernel (A/B switch)
  return kind == 3 && tiny && d->Cout <= 8 && (long long)d->N * d->H * d->W * d->ld1 < (1ll << 30);
}

static bool routes_thin32(const ssg_conv_desc* d, int kind) {
  static const int k36 = [] { const char* e = getenv("SSG_THIN32"); return e ? atoi(e) : 1; }();       // 0: 4x4x1 kernel (A/B switch)
  if (kind != 3 || !k36 || d->Cout < 32 || (long long)d->N * d->H * d->W * d->ldo >= (1ll << 30)) return false;
  if (d->W < 32 || (long long)d->N * d->H * d->W < 65536) return false;          // 32-pixel strips: small images stay on the 4-pixel kernel
  for (int t = 0; t < d->ntaps; ++t) if (d->dy[t] || d->dx[t]) return true;       // a real window, not the 1x1 case
  return false;
}

// profiling label: 12 = thin4_cin (4x4x1), 13 = thin4_cout, 14 = tiny4 (VALU), 15 = thin32_cin (32x32x2)
int ssg_thin4_conv_id(const ssg_conv_desc* d, int kind) {
  if (routes_tiny(d, kind)) return 14;
  if (routes_thin32(d, kind)) return 15;
  return 9 + kind;
}

int ssg_thin4_conv_launch(const ssg_conv_desc* d, int kind, hipStream_t st) {
  T4Args a;
  a.in = d->in1; a.w = d->w; a.bias = d->bias; a.res = d->res; a.out = d->out;
  a.C = d->C1; a.ld = d->ld1; a.N = d->N; a.H = d->H; a.W = d->W; a.Kp = d->Kp; a.kmode = d->kmode;
  a.ldr = d->ldr; a.Cout = d->Cout; a.ldo = d->ldo; a.ntaps = d->ntaps;
  a.act = d->act; a.slope = d->slope;
  a.gsave = nullptr; a.ldg = 0;
  window_taps(d, a.tapidx);
  a.nt_store = (long long)d->N * d->H * d->W * d->ldo * 4 >= (256ll << 20);
  bool ks1 = true;
  for (int p = 0; p < 9; ++p) if (p != 4 && a.tapidx[p] >= 0) ks1 = false;
  const int rh = kind == 3 ? RH_CIN : RH_COUT;
  a.strips = (d->W + 3) / 4;
  a.ybands = (d->H + rh - 1) / rh;
  a.total_units = d->N * a.ybands * a.strips;
  a.groups = kind == 3 ? (d->Cout + 63) / 64 : 1;
  int wpg = 8192 / a.groups;                         // ~32 waves per CU in total
  if (wpg > a.total_units) wpg = a.total_units;
  a.waves_per_group = wpg;
  const dim3 grid((unsigned)((wpg * a.groups + 3) / 4)), block(256);
  if (routes_tiny(d, kind)) {
    const dim3 gridt((unsigned)(((long long)d->N * d->H * d->W + 255) / 256));
    if (d->Cout <= 4) hipLaunchKernelGGL(tiny4_kernel<1>, gridt, block, 0, st, a);
    else hipLaunchKernelGGL(tiny4_kernel<2>, gridt, block, 0, st, a);
    SSG_LAUNCH_CHECK();
    return SSG_OK;
  }
  if (routes_thin32(d, kind)) {
    a.strips = (d->W + 31) / 32;
    a.ybands = (d->H + RH_K36 - 1) / RH_K36;
    a.total_units = d->N * a.ybands * a.strips;
    int wpg2 = 8192 / a.groups;
    if (wpg2 > a.total_units) wpg2 = a.total_units;
    a.waves_per_group = wpg2;
    const dim3 grid2((unsigned)((wpg2 * a.groups + 3) / 4));
    static const int nt32 = [] { const char* e = getenv("SSG_THIN32_NT"); return e ? atoi(e) : 1; }();
    const bool nt = nt32 && a.nt_store;
    if (d->res) {
      if (nt) hipLaunchKernelGGL((thin32_cin_kernel<1, 1, true>), grid2, block, 0, st, a);
      else hipLaunchKernelGGL((thin32_cin_kernel<1, 1, false>), grid2, block, 0, st, a);
    } else {
      if (nt) hipLaunchKernelGGL((thin32_cin_kernel<0, 1, true>), grid2, block, 0, st, a);
      else hipLaunchKernelGGL((thin32_cin_kernel<0, 1, false>), grid2, block, 0, st, a);
    }
    SSG_LAUNCH_CHECK();
    return SSG_OK;
  }
  if (kind == 3) {
    static const int pf = [] { const char* e = getenv("SSG_THIN4_PF"); return e ? atoi(e) : 2; }();     // rows in flight (A/B switch)
    if (d->res) {
      if (ks1) hipLaunchKernelGGL((thin4_cin_kernel<1, 1, true>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((thin4_cin_kernel<3, 1, true>), grid, block, 0, st, a);
    } else if (ks1) hipLaunchKernelGGL((thin4_cin_kernel<1, 2, false>), grid, block, 0, st, a);
    else if (pf <= 1) hipLaunchKernelGGL((thin4_cin_kernel<3, 1, false>), grid, block, 0, st, a);
    else if (pf == 2) hipLaunchKernelGGL((thin4_cin_kernel<3, 2, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((thin4_cin_kernel<3, 4, false>), grid, block, 0, st, a);
  } else {
    if (d->Cout <= 4) {
      if (ks1) hipLaunchKernelGGL((thin4_cout_kernel<1, 1>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((thin4_cout_kernel<3, 1>), grid, block, 0, st, a);
    } else {
      if (ks1) hipLaunchKernelGGL((thin4_cout_kernel<1, 2>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((thin4_cout_kernel<3, 2>), grid, block, 0, st, a);
    }
  }
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

// ------------------------------------------------------------------ SPADE: gamma|beta conv + modulation in one kernel
// 4 = the 4-channel-input kernel, 8 = the 8-channel one, 0 = not handled
static int spade_fused_cin(const ssg_conv_desc* d) {
  static const int on = [] { const char* e = getenv("SSG_SPADE_FUSED"); return e ? atoi(e) : 1; }();     // 0: conv + modulate pass (A/B); 4: only the 4-channel form
  if (!on || !d || d->res || d->act != SSG_ACT_NONE || d->bnpart || (d->Cout & 7) || d->ntaps != 9 || d->kmode != 1) return 0;
  if (d->C2 != 0 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1 || d->out_oy || d->out_ox) return 0;
  if (d->GH != d->H || d->GW != d->W || d->OH != d->H || d->OW != d->W) return 0;
  if (((uintptr_t)d->in1 & 15) || d->ld1 % 4 || ((uintptr_t)d->w & 15) || d->Kp % 4 || d->ldo % 4 || ((uintptr_t)d->out & 15)) return 0;
  int tapidx[9];
  if (!window_taps(d, tapidx)) return 0;
  if (d->W < 32 || (long long)d->N * d->H * d->W < 65536) return 0;
  if ((long long)d->N * d->H * d->W * (d->Cout / 2) >= (1ll << 30) || (long long)d->N * d->H * d->W * d->ld1 >= (1ll << 30)) return 0;
  if (d->C1 == 4) return 4;
  if (d->C1 == 8 && on != 4) return 8;
  return 0;
}

static bool spade_fused_ok(const ssg_conv_desc* d) { return spade_fused_cin(d) != 0; }

extern "C" int ssg_spade_conv_modulate_ok(const ssg_conv_desc* d) { return spade_fused_ok(d) ? 1 : 0; }

extern "C" int ssg_spade_conv_modulate_f32(const ssg_conv_desc* d, const float* x, int ldx, float* gamma, int ldg, void* stream) {
  SSG_REQUIRE(d && x && gamma && d->in1 && d->w && d->out, SSG_EINVAL, "spade_conv_modulate: null pointer");
  SSG_REQUIRE(spade_fused_ok(d), SSG_EINVAL, "spade_conv_modulate: shape not handled by the fused kernel (ssg_spade_conv_modulate_ok == 0)");
  const int C = d->Cout / 2;
  SSG_REQUIRE(ldx >= C && ldg >= C && d->ldo >= C, SSG_EINVAL, "spade_conv_modulate: pixel strides < C");
  T4Args a;
  a.in = d->in1; a.w = d->w; a.bias = d->bias; a.res = x; a.out = d->out;
  a.C = d->C1; a.ld = d->ld1; a.N = d->N; a.H = d->H; a.W = d->W; a.Kp = d->Kp; a.kmode = d->kmode;
  a.ldr = ldx; a.Cout = d->Cout; a.ldo = d->ldo; a.ntaps = d->ntaps;
  a.act = SSG_ACT_NONE; a.slope = 0.f;
  a.gsave = gamma; a.ldg = ldg;
  window_taps(d, a.tapidx);
  a.nt_store = (long long)d->N * d->H * d->W * C * 4 >= (256ll << 20);
  a.groups = (C + 31) / 32;
  a.strips = (d->W + 31) / 32;
  a.ybands = (d->H + RH_K36 - 1) / RH_K36;
  a.total_units = d->N * a.ybands * a.strips;
  int wpg = 8192 / a.groups;
  if (wpg > a.total_units) wpg = a.total_units;
  a.waves_per_group = wpg;
  const dim3 grid((unsigned)((wpg * a.groups + 3) / 4)), block(256);
  if (spade_fused_cin(d) == 8) {
    if (a.nt_store) hipLaunchKernelGGL((thin32_cin_kernel<2, 1, true, 8>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((thin32_cin_kernel<2, 1, false, 8>), grid, block, 0, (hipStream_t)stream, a);
  } else if (a.nt_store) hipLaunchKernelGGL((thin32_cin_kernel<2, 1, true>), grid, block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((thin32_cin_kernel<2, 1, false>), grid, block, 0, (hipStream_t)stream, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
