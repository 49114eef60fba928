// Implicit-GEMM 3x3 convolution with an LDS-resident halo tile (gfx950).  Same contract as
// conv_igemm_dma.hip (kmode 0, unit input stride, Cout > 32); selected when the taps are the 9
// positions of a 3x3 window -- every stride-1 3x3 forward conv and its input gradient.
//
// conv_igemm_dma.hip re-gathers the A tile (pixels x 16 channels) from global memory for each of the
// 9 taps: 9 x BM rows per 16-channel chunk, the part of its loop that costs the most (ablation:
// DESIGN.md).  Here the pixel tile is TH x 32 and its (TH+2) x 34 halo is brought into LDS ONCE per chunk;
// tap (dy,dx) is just a row offset (dy*34 + dx) on the ds_read address: global->LDS traffic for A drops
// 9x -> (TH+2)*34/(TH*32) = 1.33x (TH = 8) / 1.59x (TH = 4), and the DMA instruction count with it.
// (32-pixel-wide tile rows: the 32 lanes of an MFMA M-fragment read 32 CONSECUTIVE halo rows, which keeps
// every ds_read_b128 lane group on 16 distinct rows mod 16; a 16-wide tile would put lanes 16-31 at +18.)
//
//   * A: two halo buffers (chunk parity).  The pieces of chunk c+1 are issued one per wave per step
//     during the first steps of chunk c.
//   * B: the 3-stage ring of conv_igemm_dma.hip (weights of step s+2 issued at step s).
//   * the 9 steps of a chunk are unrolled, so every s_waitcnt vmcnt(N) is an immediate: at step s the
//     wave needs B(s); issued after it are B(s+1) and possibly one A piece.  Dummy pieces (zero page)
//     keep the counts uniform at the tail.
//   * LDS image, XOR swizzle (16-B position p of row r holds channel quad p ^ ((r>>2)&3)), zero page
//     for out-of-image pixels and the epilogue are those of conv_igemm_dma.hip.
#include "common.h"
#include "lds_dma.h"
#include "conv_args.h"
#include "conv_halo_epilogue.h"

namespace {

__device__ __attribute__((aligned(64))) float ssg_zero_page_h[64];

#ifndef SSG_HALO_EXP
#define SSG_HALO_EXP 0     // ablation builds (tools/micro_halo_exp.py), never shipped: 1 = one barrier per chunk instead of per
#endif                     // step, 2 = no weight DMA, 3 = both, 4 = s_setprio around the MFMA block, 5/6/7 = A / B / both DMA sources = zero page, 8 = contiguous B pieces



#ifdef SSG_CLOCK_PROBE
// diagnostic build only (tools/clock_probe.py): per workgroup (shader cycles, 100-MHz ticks) spent in the main loop
__device__ unsigned long long* ssg_probe_buf = nullptr;
#endif

// TWL = log2 of the tile width.  32-wide tiles keep the halo rows densely packed (pitch 34); the 16-wide tile (images
// 16 pixels wide or less: a 32-wide tile would multiply half its rows by nothing) pads the halo pitch to 32 so that lanes
// 16-31 of an M-fragment (the second image row) sit 32 halo rows after lanes 0-15 -- the same rows mod 16 as in the
// dense case, i.e. the same conflict-free ds_read_b128 pattern.
// Split-K (a.ksplit > 1): workgroup `slab` reduces the 16-channel chunks [slab*cps, (slab+1)*cps) only and writes its raw
// partial sums to a.ws[slab][pixel of the launch grid][Cout4]; conv_splitk_reduce_kernel adds the slabs in order and
// applies bias / residual / activation.  For launches whose tile count leaves most of the chip idle (the 16x16 and 32x32
// levels, the Cout <= 64 input gradients of SPADE's gamma|beta conv, batch-1 inference).
template <int BM, int BN, int WAVES_M, int WAVES_N, int TWL, bool SPLIT>
__device__ __forceinline__ void halo_body(const ConvArgs& a) {
  constexpr int TW = 1 << TWL, TH = BM / TW;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int HW = TWL == 5 ? TW + 2 : 32, HR = (TH + 2) * HW;   // halo pitch / rows (one row = one pixel x 16 channels)
  constexpr int AP = (HR + 15) / 16;                     // 1-KiB pieces per halo tile
  constexpr int APW = (AP + 3) / 4;                      // pieces per wave (dummy-padded), one per step
  constexpr int B_PC = BN / 64;                          // B pieces per wave per step
  constexpr int ABUF = AP * 256;                         // floats per halo buffer
  constexpr int BSTG = BN * 16;                          // floats per B stage
  static_assert(APW <= 7, "A pieces must be issued before the last two steps of a chunk");

  extern __shared__ __attribute__((aligned(1024))) float lds[];     // 2 * ABUF + 3 * BSTG floats
  float* const ldsB = lds + 2 * ABUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  int bid = blockIdx.x;
  if (a.xcd_swizzle) {
    const int per = (int)gridDim.x >> 3;
    if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
  }
  const int nyt = a.ntiles_n;
  const int n0 = (bid % nyt) * BN; bid /= nyt;
  int slab = 0;
  if (SPLIT) { slab = bid % a.ksplit; bid /= a.ksplit; }
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int n = bid / a.tiles_y;

  // ---- per-lane DMA source state
  const int lr = lane >> 2, lp = lane & 3;
  int a_pix[APW];                   // pixel index of this lane's halo row of piece k, or -1 (zero page)
  int a_q[APW];
#pragma unroll
  for (int k = 0; k < APW; ++k) {
    const int g = wave + 4 * k;                          // piece index: waves interleave
    const int r = g * 16 + lr;
    const int hy = r / HW, hx = r - hy * HW;
    const int iy = ty * TH + hy - 1, ix = tx * TW + hx - 1;
    const bool ok = g < AP && r < HR && hx < TW + 2 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    a_pix[k] = ok ? (n * a.H + iy) * a.W + ix : -1;
    a_q[k] = 4 * (lp ^ ((r >> 2) & 3));
  }
  const float* b_src[B_PC];
#pragma unroll
  for (int j = 0; j < B_PC; ++j) {
    const int r = (wave * B_PC + j) * 16 + lr;
    const int q = 4 * (lp ^ ((r >> 2) & 3));
    b_src[j] = (n0 + r < a.Cout) ? a.w + (size_t)(n0 + r) * a.Kp + q : nullptr;
  }
  const float* zero = ssg_zero_page_h;
  const int cps = SPLIT ? (((a.C1 + a.C2) >> 4) + a.ksplit - 1) / a.ksplit : 0;      // chunks per slab
  const int chunk0 = SPLIT ? slab * cps : 0;
  const int nchunks = SPLIT ? min((a.C1 + a.C2) >> 4, chunk0 + cps) : (a.C1 + a.C2) >> 4;   // end of this workgroup's chunk range
  const int nsteps = nchunks * 9;

  // piece k of this wave for chunk `chunk` (a dummy zero-page piece past the last chunk or past AP)
  auto issue_a = [&](int chunk, int k) {
    const int g = wave + 4 * k;
    if (g >= AP) { return; }                             // wave-uniform: this wave has no k-th piece
    float* dst = lds + (chunk & 1) * ABUF + g * 256;
    const int c0 = chunk * 16;
    const float* src; int ld, cc;
    if (c0 < a.C1) { src = a.in1; ld = a.ld1; cc = c0; } else { src = a.in2; ld = a.ld2; cc = c0 - a.C1; }
    const float* p = (chunk < nchunks && a_pix[k] >= 0) ? src + (size_t)a_pix[k] * ld + cc + a_q[k] : zero;
#if SSG_HALO_EXP == 5 || SSG_HALO_EXP == 7
    p = zero + (lane & 3) * 4;                           // every A piece from the (cache-resident) zero page
#endif
    dma16(p, dst);
  };
  auto issue_b = [&](int s) {
    float* st = ldsB + (s % 3) * BSTG;
#pragma unroll
    for (int j = 0; j < B_PC; ++j) {
      const float* p = (b_src[j] && s < nsteps) ? b_src[j] + (size_t)s * 16 : zero;
#if SSG_HALO_EXP == 6 || SSG_HALO_EXP == 7
      p = zero + (lane & 3) * 4;                         // every B piece from the zero page
#endif
#if SSG_HALO_EXP == 8                                    // B pieces from 1-KiB contiguous runs (what a step-major pack would give)
      if (s < nsteps) p = a.w + ((size_t)(n0 / BN) * nsteps + s) * BSTG + (wave * B_PC + j) * 256 + lane * 4;
#endif
      dma16(p, st + (wave * B_PC + j) * 256);
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int half = lane >> 5, l31 = lane & 31;
  // halo row of this lane's pixel for M-fragment i at tap (0,0)
  int rb[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int p = wm * WTM + i * 32 + l31;
    rb[i] = ((p >> TWL) + 1) * HW + (p & (TW - 1)) + 1;
  }
  const int swb = (l31 >> 2) & 3;
  const int bq0 = 4 * ((0 + half) ^ swb), bq1 = 4 * ((2 + half) ^ swb);

  // ---- prologue: halo of chunk 0, weights of steps 0 and 1
#pragma unroll
  for (int k = 0; k < APW; ++k) issue_a(chunk0, k);
  issue_b(chunk0 * 9);
  issue_b(chunk0 * 9 + 1);

  // whether THIS wave issues an A piece at tap-step t (wave-uniform, but not compile-time for the last k)
  // -> make the count compile-time: waves without a k-th piece issue nothing and the wait is sized per wave
  // class below (has_last = the wave owns a piece with index k = APW-1).
  const bool has_last = wave + 4 * (APW - 1) < AP;

#ifdef SSG_CLOCK_PROBE
  const unsigned long long pt0 = __builtin_amdgcn_s_memtime(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int chunk = chunk0; chunk < nchunks; ++chunk) {
    const float* Abuf = lds + (chunk & 1) * ABUF;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int s = chunk * 9 + t;
      // outstanding after B(s): B(s+1) and the A piece of the previous step (if that step had one)
      constexpr int dummy = 0; (void)dummy;
      const int tp = (t + 8) % 9;                        // previous tap-step
      if (tp < APW - 1) wait_vmcnt<B_PC + 1>();
      else if (tp == APW - 1) { if (has_last) wait_vmcnt<B_PC + 1>(); else wait_vmcnt<B_PC>(); }
      else wait_vmcnt<B_PC>();
#if SSG_HALO_EXP == 1 || SSG_HALO_EXP == 3
      if (t == 0) __builtin_amdgcn_s_barrier();
#else
      wait_lds_reads();                                  // lds_dma.h: the barrier hands stage (s + 2) % 3 to another wave's DMA
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
      if (t < APW) issue_a(chunk + 1, t);
#if SSG_HALO_EXP == 2 || SSG_HALO_EXP == 3
      if (t >= APW) dma16(zero, ldsB + (s % 3) * BSTG + wave * 256);     // keeps the vmcnt arithmetic alive with one piece
#else
      issue_b(s + 2);
#endif

      const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
      const int toff = ((tb & 7) - 2) * HW + ((tb >> 3) - 2);          // dy*HW + dx
      const float* Bb = ldsB + (s % 3) * BSTG + (wn * WTN + l31) * 16;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int r = rb[i] + toff;
          const int q = 4 * (((2 * h) + half) ^ ((r >> 2) & 3));
          fa[i] = *(const f32x4*)(Abuf + r * 16 + q);
        }
        const int bq = h == 0 ? bq0 : bq1;
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[j] = *(const f32x4*)(Bb + j * 32 * 16 + bq);
#if SSG_HALO_EXP == 4
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
#if SSG_HALO_EXP == 4
        __builtin_amdgcn_s_setprio(0);
#endif
      }
    }
  }
  // drain the dummy pieces before the workgroup's LDS can be handed to another workgroup
  wait_vmcnt<0>();
#ifdef SSG_CLOCK_PROBE
  if (ssg_probe_buf && tid == 0) {
    ssg_probe_buf[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - pt0;
    ssg_probe_buf[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - pr0;
  }
#endif

  ssg_halo_epilogue<BM, BN, WAVES_M, WAVES_N, TWL, SPLIT>(a, acc, lds, n, ty, tx, n0, slab, wm, wn, half, l31);
}

// second launch bound = waves per SIMD of the 3 (4 for <128,64>) workgroups per CU the LDS footprint is sized for: the pivoted
// statistics of the epilogue cost 4 registers and the allocator would otherwise trade a resident workgroup for them
// (<128,128> sat at 167 of 168)
template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, (BN == 128 || BM == 256) ? 3 : 4) void conv_igemm_halo_kernel(const ConvArgs a) { halo_body<BM, BN, WAVES_M, WAVES_N, 5, false>(a); }

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_halo16_kernel(const ConvArgs a) { halo_body<BM, BN, WAVES_M, WAVES_N, 4, false>(a); }

template <int BM, int BN, int WAVES_M, int WAVES_N, int TWL>
__global__ __launch_bounds__(256) void conv_igemm_halo_splitk_kernel(const ConvArgs a) { halo_body<BM, BN, WAVES_M, WAVES_N, TWL, true>(a); }

// out = act(sum over slabs (in slab order) + bias + res): second stage of a split-K launch
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvArgs a) {
  const int ldw = (a.Cout + 3) & ~3, q4 = ldw >> 2;
  const long long npix = (long long)a.N * a.GH * a.GW, total = npix * q4;
  const size_t slab_stride = (size_t)npix * ldw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long gp = i / q4; const int c0 = 4 * (int)(i - gp * q4);
    const float* src = a.ws + (size_t)gp * ldw + c0;
    f32x4 v = *(const f32x4*)src;
    for (int k = 1; k < a.ksplit; ++k) v += *(const f32x4*)(src + k * slab_stride);
    const int gx = (int)(gp % a.GW); const long long t = gp / a.GW;
    const int gy = (int)(t % a.GH), n = (int)(t / a.GH);
    const size_t pix = ((size_t)(n * a.OH + gy * a.out_sy + a.out_oy) * a.OW + gx * a.out_sx + a.out_ox);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = c0 + e;
      float x = v[e];
      if (co < a.Cout) {
        if (a.bias) x += a.bias[co];
        if (a.res) x += a.res[pix * a.ldr + co];
        if (a.act == SSG_ACT_RELU) x = x < 0.f ? 0.f : x;
        else if (a.act == SSG_ACT_LRELU) x = x > 0.f ? x : x * a.slope;
      } else x = 0.f;
      o[e] = x;
    }
    *(f32x4*)(a.out + pix * a.ldo + c0) = o;
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int TWL>
int launch(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  constexpr int TW = 1 << TWL, TH = BM / TW;
  constexpr int AP = ((TH + 2) * (TWL == 5 ? TW + 2 : 32) + 15) / 16;
  a.tiles_x = (a.GW + TW - 1) / TW;
  a.tiles_y = (a.GH + TH - 1) / TH;
  static const int swz = [] { const char* e = getenv("SSG_XCD_SWIZZLE"); return e ? atoi(e) : 1; }();
  a.xcd_swizzle = swz;
  a.ntiles_n = (a.Cout + BN - 1) / BN;
  if (a.ksplit < 1) a.ksplit = 1;
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N * a.ntiles_n * a.ksplit));
  constexpr int lds_bytes = (2 * AP * 256 + 3 * BN * 16) * (int)sizeof(float);
  static_assert(lds_bytes <= 64 * 1024, "LDS budget");
  if (a.ksplit > 1) {
    if constexpr (BM == 128 && !(BN == 128 && TWL == 5)) hipLaunchKernelGGL((conv_igemm_halo_splitk_kernel<BM, BN, WAVES_M, WAVES_N, TWL>), grid, dim3(256), lds_bytes, st, a);
    else { ssg_set_error("conv halo: this tile has no split-K instantiation"); return SSG_EINVAL; }
  } else if constexpr (TWL == 5) hipLaunchKernelGGL((conv_igemm_halo_kernel<BM, BN, WAVES_M, WAVES_N>), grid, dim3(256), lds_bytes, st, a);
  else hipLaunchKernelGGL((conv_igemm_halo16_kernel<BM, BN, WAVES_M, WAVES_N>), grid, dim3(256), lds_bytes, st, a);
  SSG_LAUNCH_CHECK();
  if (a.ksplit > 1) {
    long long blocks = ((long long)a.N * a.GH * a.GW * (((a.Cout + 3) & ~3) >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    SSG_LAUNCH_CHECK();
  }
  return SSG_OK;
}

}  // namespace

#ifdef SSG_CLOCK_PROBE
extern "C" int ssg_debug_set_probe_buffer(void* p) {
  unsigned long long* v = (unsigned long long*)p;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(ssg_probe_buf), &v, sizeof(v));
}
#endif

// the 9 taps must be the 9 positions of the 3x3 window (any order), unit input stride
bool ssg_conv_halo_ok(const ConvArgs& a) {
  if (a.ntaps != 9 || a.in_sy != 1 || a.in_sx != 1 || a.kmode != 0) return false;
  unsigned seen = 0;
  for (int t = 0; t < 9; ++t) {
    const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
    const int dy = (tb & 7) - 2, dx = (tb >> 3) - 2;
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
    seen |= 1u << ((dy + 1) * 3 + dx + 1);
  }
  return seen == 0x1ffu;
}

int ssg_conv_igemm_halo_launch(const ConvArgs& a, int variant, hipStream_t st) {
  switch (ssg_conv_halo_variant(a, variant)) {
    case 0: return launch<128, 128, 2, 2, 5>(a, st);
    case 2: return launch<128, 64, 2, 2, 5>(a, st);
    case 3: return launch<128, 128, 2, 2, 4>(a, st);
    case 4: return launch<128, 64, 2, 2, 4>(a, st);
    default: return launch<256, 64, 4, 1, 5>(a, st);
  }
}

// pixel-tile size of a variant (rows of the batch-norm partial buffer, split-K workspace geometry)
void ssg_conv_halo_tile(int v, int* th, int* tw, int* bn) {
  *tw = v >= 3 ? 16 : 32;
  *th = (v == 1 ? 256 : 128) / *tw;
  *bn = (v == 0 || v == 3) ? 128 : 64;
}

// 0 = <128,128> (Cout > 64), 1 = <256,64>, 2 = <128,64>: for Cout <= 64 with a short K loop (Cin <= 128: 36-72
// steps per tile) four small workgroups per CU hide each other's prologue and epilogue better than two big ones
// (measured at 16x512^2: Cin=64 108 -> 122 TFLOP/s, Cin=128 123 -> 128, Cin=192 132 -> 130)
// 3 / 4 = <128,128> / <128,64> with 8 x 16-pixel tiles, for images at most 16 pixels wide.
int ssg_conv_halo_variant(const ConvArgs& a, int variant) {
  static const int w16 = [] { const char* e = getenv("SSG_HALO_W16"); return e ? atoi(e) : 1; }();
  if (w16 && a.GW <= 16) return (variant == 0) ? 3 : 4;
  // Cout > 64 with Cin = 64 on the largest grids also prefers two <128,64> column tiles (110 -> 120 at 16x512^2)
  if (variant == 0) {            // fewer than 3 workgroups per CU with 128x128 tiles: halve the tile (32x32 level)
    const long long wgs = (long long)a.N * ((a.GH + 3) / 4) * ((a.GW + 31) / 32) * ((a.Cout + 127) / 128);
    if (wgs < 768) return 2;
  }
  if (variant == 0) return ((a.C1 + a.C2) <= 64 && (long long)a.N * a.GH * a.GW >= (1ll << 22)) ? 2 : 0;
  if ((a.C1 + a.C2) <= 128) return 2;
  // Cout <= 64 with a long K loop: 256-pixel tiles, unless they leave CUs without a workgroup (the 32x32 / 64x64 levels)
  return (long long)a.N * ((a.GH + 7) / 8) * ((a.GW + 31) / 32) < 512 ? 2 : 1;
}

// Split-K factor for a launch: as many slabs as bring the grid to ~3 workgroups per CU, each slab keeping at least 4
// chunks (36 K-steps); 1 when the grid already fills the chip or no workspace was given.  SSG_HALO_SPLITK=0 switches it off.
int ssg_conv_halo_ksplit(const ConvArgs& a, int variant) {
  static const int on = [] { const char* e = getenv("SSG_HALO_SPLITK"); return e ? atoi(e) : 1; }();
  if (!on || a.bnpart) return 1;
  int th, tw, bn;
  const int hv = ssg_conv_halo_variant(a, variant);
  if (hv == 0 || hv == 1) return 1;                         // <128,128> / <256,64> on 32-wide tiles are only picked for grids that fill the chip
  ssg_conv_halo_tile(hv, &th, &tw, &bn);
  const long long wgs = (long long)a.N * ((a.GH + th - 1) / th) * ((a.GW + tw - 1) / tw) * ((a.Cout + bn - 1) / bn);
  const int chunks = (a.C1 + a.C2) >> 4;
  if (wgs >= 512 || chunks < 8) return 1;
  long long k = (768 + wgs - 1) / wgs;
  if (k > chunks / 4) k = chunks / 4;
  if (k > 16) k = 16;
  if (k < 2) return 1;
  const int cps = (chunks + (int)k - 1) / (int)k;           // no empty slab
  return (chunks + cps - 1) / cps;
}
