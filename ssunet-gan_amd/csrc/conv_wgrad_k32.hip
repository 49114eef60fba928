// Weight gradient of 3x3 stride-1 convolutions, split-operand form on v_mfma_f32_16x16x32_bf16 (round 4).  Same contract and slab
// layout as wgrad_halo_x3_kernel (conv_wgrad_halo.hip): dW[tap][ci][co] = sum over pixels of x[p + tap][ci] * dy[p][co], fp32 tensors
// in HBM, both operands split into three bf16 terms (mfma_split.h), six products, fp32 accumulation, split-K slabs
// [split][tap * Cin + ci][Cout] reduced in order by wgrad_reduce_kernel.  What is new:
//   * the split happens ONCE per element, on the way from HBM into LDS (buffer loads -> registers -> split3 -> bf16 planes
//     [pixel][64 channels], 128-byte rows), not in every wave that consumes the element (wgrad_halo_x3: ~4 VALU per MFMA, the x
//     window split in all four waves); fragments leave LDS by ds_read_b64_tr_b16 (hardware transpose: per 16-lane group 4 pixels x
//     16 channels, channel-major), two per 16x16x32 operand.  A chunk XOR keyed by pixel bits 1 and 3 makes the eight 32-byte
//     pieces a 32-lane half touches fill one 256-byte bank window at ANY pixel alignment (the taps shift the window by one pixel);
//   * a K-step = 32 pixels of one image row, steps walk DOWN a 32-pixel column strip and the 3-row x window ROLLS through four
//     LDS row slots: a step brings one new x row (34 pixels x 64 channels) and one dy row (32 x 64) -- 17 KB of fp32 per 9 x 64 x 64 x 32
//     MACs = 140 FLOP/B from L2, against 29 KB (82 FLOP/B) for the window reloaded per step;
//   * 512 threads: M = 9 taps x 64 channels, N = 64 output channels; wave (wm, wn) = 16 input channels x 9 taps x 2 output-channel
//     fragments (18 accumulators), dy as the A operand (rows = output channels) so a lane stores 4 consecutive output channels;
//     66 transposed reads and 108 MFMAs per wave between two barriers; the 16x16x32 shape for the clock (tools/mfma_lab.hip).
// LDS: x 3 planes x 4 slots x 36 pixels x 128 B = 54 KB, dy 3 x 2 x 32 x 128 B = 24 KB.
#include "common.h"
#include "lds_dma.h"
#include "conv_wgrad_args.h"
#include "mfma_split.h"
#include "conv_slow.h"
#include <stdlib.h>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int KP = 32;                     // pixels per K-step
constexpr int WPX = 36;                    // pixel rows per x slot (34 used: window columns gx0 - 1 .. gx0 + 32)
constexpr int XSLOT = WPX * 128;           // bytes
constexpr int XPLANE = 4 * XSLOT;
constexpr int XIMG = 3 * XPLANE;
constexpr int DSLOT = KP * 128;
constexpr int DPLANE = 2 * DSLOT;
constexpr int DIMG = 3 * DPLANE;
constexpr int CB = 64, BN = 64;
#ifndef SSG_WK32_FLUSH
#define SSG_WK32_FLUSH 0                       // > 0: rows of MFMA accumulation per flush into vector-unit totals (A/B; costs 72 registers and ~10 %); 0: the slab
                                               // length bounds the accumulation chain instead (conv_wgrad.hip: at most 128 rows per slab)
#endif

// 16-byte chunk c (0..7) of pixel row p sits at chunk position c ^ sw(p)
__device__ __forceinline__ int sw(int p) { return ((((p >> 1) & 1) | (((p >> 3) & 1) << 1)) << 1); }

// XF: the x operand is act(x * in_scale[c] + in_shift[c]) (ssg_wgrad_desc.in_scale: a batch-norm apply that was never materialised),
// applied to a staged element on its way into LDS; a thread stages the same 8 channels for the whole launch, so its constants
// are 16 registers loaded once.  Window pixels outside the image stay zero (the padding of the ACTIVATED tensor).
template <bool XF>
__global__ __launch_bounds__(512, 1) void wgrad_k32_kernel(const WgArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  unsigned char* const ximg = lds;
  unsigned char* const dimg = lds + XIMG;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;               // 16-channel block of the 64, pair of output-channel fragments
  const int l15 = lane & 15, g = lane >> 4;
  const int c0 = blockIdx.x * CB, n0 = blockIdx.y * BN;
  const int Cin = a.C1 + a.C2;
  const int XB = (a.GW + KP - 1) / KP;

  // ---- this workgroup's range of K-steps, S = strip * GH + gy (strip = image * XB + column strip)
  const long long total = (long long)a.N * XB * a.GH;
  long long S0 = (long long)blockIdx.z * a.steps_per_split, S1 = S0 + a.steps_per_split;
  if (S1 > total) S1 = total;

  const unsigned OOB = 0xffffffffu;
  const bool first = c0 < a.C1;
  const float* xsrc = first ? a.in1 : a.in2;
  const int xld = first ? a.ld1 : a.ld2;
  const int xc0 = first ? c0 : c0 - a.C1;
  const unsigned npix = (unsigned)a.N * (unsigned)a.H * (unsigned)a.W;
  const auto x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsrc), 0, (int)(npix * (unsigned)xld * 4u), 0x00020000);
  const auto d_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dout), 0, (int)((unsigned)a.N * (unsigned)a.GH * (unsigned)a.GW * (unsigned)a.ldd * 4u), 0x00020000);

  // ---- staging items, 16-byte pieces in lane order: a wave instruction covers 4 pixels x the 256 bytes of their 64 channels (whole lines;
  // the first form -- a thread = one pixel's 8-channel chunk as two loads 16 bytes apart -- touched every line from two instructions, and
  // had threads 0..271 stage x, 256..511 dy: wave 4 did both).  Piece t of the new x row = (window pixel t >> 4, quarter t & 15) for every
  // thread, pieces 512..543 (window pixels 32, 33) once more for threads 0..31; piece t of the dy row for every thread.
  const int s_px = tid >> 4, s_q = tid & 15;
  const bool has_x2 = tid < 32;
  const int x_dst = s_px * 128 + (((s_q >> 1) ^ sw(s_px)) << 4) + (s_q & 1) * 8;
  const int x_dst2 = (32 + s_px) * 128 + (((s_q >> 1) ^ sw(32 + s_px)) << 4) + (s_q & 1) * 8;      // threads 0..31: s_px = 0, 1
  const int d_dst = x_dst;                               // same (pixel, quarter) position in the dy slot

  struct RawX { u32x4 a, b; };
  struct RawD { u32x4 a; };
  float* const xtab = (float*)(lds + XIMG + DIMG);        // XF: scale[64] | shift[64] of this workgroup's input channels
  if constexpr (XF) {                                    // C2 == 0 (ssg_conv2d_wgrad_in_affine_ok); visible after the first segment's barrier
    if (tid < 128) xtab[tid] = tid < 64 ? a.in_scale[c0 + tid] : a.in_shift[c0 + tid - 64];
  }
  auto x_inside = [&](int row, int gx0, int px) -> bool {        // is window pixel px of row `row` inside the image
    return (unsigned)row < (unsigned)a.H && (unsigned)(gx0 - 1 + px) < (unsigned)a.W;
  };
  auto load_x = [&](int n, int row, int gx0) -> RawX {   // window row `row` (may be -1 or H: zeros), columns gx0 - 1 .. gx0 + 32
    const unsigned rowbase = (unsigned)((n * a.H + row) * a.W + gx0 - 1);
    const unsigned cho = (unsigned)(xc0 + 4 * s_q) * 4u;
    const unsigned vo = x_inside(row, gx0, s_px) ? (rowbase + (unsigned)s_px) * (unsigned)xld * 4u + cho : OOB;
    const unsigned vo2 = (has_x2 && x_inside(row, gx0, 32 + s_px)) ? (rowbase + 32u + (unsigned)s_px) * (unsigned)xld * 4u + cho : OOB;
    RawX r;
    r.a = __builtin_amdgcn_raw_buffer_load_b128(x_rs, vo, 0, 0);
    r.b = __builtin_amdgcn_raw_buffer_load_b128(x_rs, vo2, 0, 0);
    return r;
  };
  auto load_d = [&](int n, int gy, int gx0) -> RawD {
    const int gx = gx0 + s_px;
    const unsigned vo = gx < a.GW ? (unsigned)((n * a.GH + gy) * a.GW + gx) * (unsigned)a.ldd * 4u + (unsigned)(n0 + 4 * s_q) * 4u : OOB;
    RawD r;
    r.a = __builtin_amdgcn_raw_buffer_load_b128(d_rs, vo, 0, 0);
    return r;
  };
  auto put = [&](unsigned char* base, int plane_bytes, f32x4 v) {
    bf16x4 p1, p2, p3;
    split3_4(v, p1, p2, p3);
    *(bf16x4*)base = p1; *(bf16x4*)(base + plane_bytes) = p2; *(bf16x4*)(base + 2 * plane_bytes) = p3;
  };
  auto xform = [&](f32x4 v, bool inside) -> f32x4 {
    if constexpr (XF) {
      v = v * *(const f32x4*)(xtab + 4 * s_q) + *(const f32x4*)(xtab + 64 + 4 * s_q);      // bn_apply_kernel's expression
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ssg_act(v[e], a.in_act, a.in_slope);
      if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return v;
  };
  auto store_x = [&](const RawX& r, int slot, int row, int gx0) {
    put(ximg + slot * XSLOT + x_dst, XPLANE, xform(__builtin_bit_cast(f32x4, r.a), XF ? x_inside(row, gx0, s_px) : true));
    if (has_x2) put(ximg + slot * XSLOT + x_dst2, XPLANE, xform(__builtin_bit_cast(f32x4, r.b), XF ? x_inside(row, gx0, 32 + s_px) : true));
  };
  auto store_d = [&](const RawD& r, int slot) {
    put(dimg + slot * DSLOT + d_dst, DPLANE, __builtin_bit_cast(f32x4, r.a));
  };

  // ---- transposed-read addresses.  Operand k-group g covers pixels 8g .. 8g+7 of the step; lane 4q+pp of the group supplies pixel
  // row q (t adds 4), channels 4pp .. 4pp+3 of the 16-channel block
  const int q4 = l15 >> 2, pp = l15 & 3;
  int xoff[3][2], doff[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int p = dx + 8 * g + q4 + 4 * t;             // window pixel: step pixel + 1 + (dx - 1)
      xoff[dx][t] = p * 128 + ((((2 * wm + (pp >> 1)) ^ sw(p))) << 4) + (pp & 1) * 8;
    }
    const int p = 8 * g + q4 + 4 * t;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      doff[j][t] = p * 128 + ((((2 * (2 * wn + j) + (pp >> 1)) ^ sw(p))) << 4) + (pp & 1) * 8;
  }

  // Accumulation chains: the rounding error of a slab grows with the number of pixels one accumulator sums.  Default: the planner keeps
  // a slab at <= 128 rows (4096 pixels, wgrad_halo_x3's slab at the bench sizes; rms error vs fp64 then equals that kernel's).  With
  // FLUSH > 0 acc takes FLUSH rows and is then added into tot by the vector unit (slabs may be long): measured 205 vs 228 TFLOP/s.
  constexpr int FLUSH = SSG_WK32_FLUSH;
  f32x4 acc[9][2], tot[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) { acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f}; tot[t][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  int since = 0;

  auto frag = [&](const unsigned char* base, int o0, int o1) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  long long S = S0;
  while (S < S1) {
    // ---- segment: rows [ga, gb) of one column strip
    const int strip = (int)(S / a.GH), ga = (int)(S - (long long)strip * a.GH);
    long long Se = (long long)(strip + 1) * a.GH;
    if (Se > S1) Se = S1;
    const int gb = ga + (int)(Se - S);
    const int n = strip / XB, gx0 = (strip - n * XB) * KP;

    // prologue: x rows ga-1, ga, ga+1 and dy row ga.  slot(r) = (r + 1) & 3.
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();                        // the previous segment's last reads are done
    asm volatile("" ::: "memory");
    {
      const RawX r0 = load_x(n, ga - 1, gx0), r1 = load_x(n, ga, gx0), r2 = load_x(n, ga + 1, gx0);
      const RawD r3 = load_d(n, ga, gx0);                // all in flight together
      store_x(r0, (ga + 0) & 3, ga - 1, gx0); store_x(r1, (ga + 1) & 3, ga, gx0);
      store_x(r2, (ga + 2) & 3, ga + 1, gx0); store_d(r3, ga & 1);
    }

    for (int gy = ga; gy < gb; ++gy) {
      wait_lds_reads();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const bool more = gy + 1 < gb;
      RawX nx; RawD nd;
#ifndef SSG_WK32_LOAD_MID
#define SSG_WK32_LOAD_MID 0                                // 1: the next rows' loads go out after the first tap row's MFMAs instead of behind the barrier (A/B build switch)
#endif
#if !SSG_WK32_LOAD_MID
      if (more) { nx = load_x(n, gy + 2, gx0); nd = load_d(n, gy + 1, gx0); }
#endif

      const unsigned char* db = dimg + (gy & 1) * DSLOT;
      bf16x8 df[2][3];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) df[j][q] = frag(db + q * DPLANE, doff[j][0], doff[j][1]);
#pragma unroll
      for (int dyi = 0; dyi < 3; ++dyi) {
#ifndef SSG_WK32_DYNPRIO
#define SSG_WK32_DYNPRIO 1                                 // 1: a wave's priority falls as it advances through a step (as conv_halo_k32_kernel; A/B build switch)
#endif
        if (SSG_WK32_DYNPRIO) { if (dyi == 0) __builtin_amdgcn_s_setprio(3); else if (dyi == 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
        const unsigned char* xb = ximg + ((gy + dyi) & 3) * XSLOT;        // row gy - 1 + dyi -> slot (row + 1) & 3
        bf16x8 xf[3][3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int q = 0; q < 3; ++q) xf[dx][q] = frag(xb + q * XPLANE, xoff[dx][0], xoff[dx][1]);
        // small terms first; A = dy (rows = output channels), B = x (columns = input channels)
#define SSG_WK_TERM(QD, QX)                                                                       \
  _Pragma("unroll") for (int dx = 0; dx < 3; ++dx)                                                \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
      acc[dyi * 3 + dx][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[j][QD], xf[dx][QX], acc[dyi * 3 + dx][j], 0, 0, 0);
        SSG_WK_TERM(2, 0) SSG_WK_TERM(1, 1) SSG_WK_TERM(0, 2)
        SSG_WK_TERM(1, 0) SSG_WK_TERM(0, 1)
        SSG_WK_TERM(0, 0)
#undef SSG_WK_TERM
#if SSG_WK32_LOAD_MID
        if (dyi == 0) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) { nx = load_x(n, gy + 2, gx0); nd = load_d(n, gy + 1, gx0); }
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
      }
      if (SSG_WK32_DYNPRIO) __builtin_amdgcn_s_setprio(0);
      if (more) {
        __builtin_amdgcn_sched_barrier(0);
        store_x(nx, (gy + 3) & 3, gy + 2, gx0);         // row gy + 2
        store_d(nd, (gy + 1) & 1);
      }
      if (FLUSH > 0 && ++since == FLUSH) {
        since = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j) { tot[t][j] += acc[t][j]; acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
    }
    S = Se;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) tot[t][j] += acc[t][j];

  wait_lds_reads();
  {                                                      // non-finite operands: conv_slow.h
    bool bad = false;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= ssg_nonfinite(tot[t][j][r]);
    if (__builtin_amdgcn_readfirstlane(__syncthreads_or(bad))) {     // scalar condition: a uniform branch, the accumulators are dead inside it
      const WgArgs& as = *ssg_reload_args<WgArgs>();
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          ssg_slow_refill4(tot[t][j], (float*)lds + tid, 512, [&](int r) {
            return ssg_wgrad_slow_value_strips<XF>(as, t, c0 + wm * 16 + l15, n0 + (2 * wn + j) * 16 + 4 * g + r, S0, S1, KP);
          });
    }
  }

  // ---- slab [split][row = tap * Cin + ci][Cout]: acc[tap][j][r] = (ci = c0 + wm*16 + l15, co = n0 + (2wn + j)*16 + 4g + r)
  float* slab = a.ws + (size_t)blockIdx.z * a.M * a.Cout;
  const int ci = c0 + wm * 16 + l15;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *(f32x4*)(slab + ((size_t)t * Cin + ci) * a.Cout + n0 + (2 * wn + j) * 16 + 4 * g) = tot[t][j];
}

}  // namespace

// eligible: the 3x3 row-major window at unit stride over the whole image, 64-channel multiples on both inputs and on the output,
// 16-byte-aligned rows, tensors within 32-bit byte offsets.  SSG_WGRAD_K32=0 switches it off (A/B).
static int g_wk32_mode = -1;
extern "C" int ssg_wgrad_set_k32_mode(int mode) {         // 0 = off, 1 = on (default; SSG_WGRAD_K32)
  SSG_REQUIRE(mode == 0 || mode == 1, SSG_EINVAL, "wgrad k32 mode %d", mode);
  g_wk32_mode = mode;
  return SSG_OK;
}

bool ssg_wgrad_k32_ok(const ssg_wgrad_desc* d) {
  if (g_wk32_mode < 0) { const char* e = getenv("SSG_WGRAD_K32"); g_wk32_mode = e ? atoi(e) : 1; }
  if (!g_wk32_mode || !(d->flags & 1)) return false;
  if (d->ntaps != 9 || d->in_sy != 1 || d->in_sx != 1 || d->GH != d->H || d->GW != d->W) return false;
  for (int t = 0; t < 9; ++t)
    if (d->dy[t] != t / 3 - 1 || d->dx[t] != t % 3 - 1) return false;
  if (d->C1 % 64 || d->C2 % 64 || d->Cout % 64 || d->GW < 17) return false;
  if ((d->ld1 & 3) || (d->C2 && (d->ld2 & 3)) || (d->ldd & 3)) return false;
  const unsigned long long xb = (unsigned long long)d->N * d->H * d->W * (unsigned long long)(d->ld1 > d->ld2 ? d->ld1 : d->ld2) * 4ull;
  const unsigned long long db = (unsigned long long)d->N * d->GH * d->GW * (unsigned long long)d->ldd * 4ull;
  return xb <= 0xfffffff0ull && db <= 0xfffffff0ull;
}

int ssg_wgrad_k32_flush() { return SSG_WK32_FLUSH; }

// K-steps of the k32 kernel: one per image row of each 32-pixel column strip
long long ssg_wgrad_k32_steps(const ssg_wgrad_desc* d) { return (long long)d->N * ((d->GW + KP - 1) / KP) * d->GH; }

int ssg_wgrad_k32_launch(const WgArgs& a, dim3 grid, hipStream_t st) {
  constexpr int lds_bytes = XIMG + DIMG;
  static const hipError_t attr = hipFuncSetAttribute((const void*)wgrad_k32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  static const hipError_t attr_xf = hipFuncSetAttribute((const void*)wgrad_k32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes + 512);
  if (attr != hipSuccess || attr_xf != hipSuccess) { ssg_set_error("wgrad k32: LDS attribute: %s", hipGetErrorString(attr != hipSuccess ? attr : attr_xf)); return (int)(attr != hipSuccess ? attr : attr_xf); }
  if (a.in_scale) hipLaunchKernelGGL(wgrad_k32_kernel<true>, grid, dim3(512), lds_bytes + 512, st, a);
  else hipLaunchKernelGGL(wgrad_k32_kernel<false>, grid, dim3(512), lds_bytes, st, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
