// Weight-gradient implicit GEMM for gfx950: dw[(t,c)][co] = sum_pixels in[pix+tap t][c] * dout[pix][co]
// fp32 MFMA 32x32x2.  ABI: include/ssunet_hip.h (ssg_conv2d_wgrad_f32, ssg_pack_weights_f32).
//
// GEMM view: M = taps x input channels (row = t*Cin + c), N = output channels, K = pixels.
// Both operands are channel-contiguous in NHWC, which is exactly the MFMA's "row on the
// lane" direction here: LDS tiles are [16 pixels][rows], and a wave's ds_read_b32 of one
// pixel touches 32 consecutive floats (conflict-free); lanes 32-63 take the next pixel (k=1).
// The pixel reduction is split over gridDim.z; every split writes its own [M][Cout] slab and
// an ordered second-stage kernel sums the slabs and scatters into the OIHW gradient, so the
// result is bitwise reproducible (no float atomics).
#include "common.h"
#include "conv_thin.h"
#include "conv_wgrad_args.h"
#include <stdlib.h>

namespace {

constexpr int BKP = 16;

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgArgs a) {
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int AQ = BM / 4, BQ = BN / 4;                // float4 quads per pixel row
  constexpr int A_LD = BKP * AQ / 256, B_LD = (BKP * BQ + 255) / 256;
  constexpr int A_PSTEP = 256 / AQ, B_PSTEP = 256 / BQ;
  static_assert(A_LD >= 1, "tile");

  __shared__ __attribute__((aligned(16))) float lds[2 * BKP * (LDA + LDB)];
  float* As = lds;
  float* Bs = lds + 2 * BKP * LDA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int half = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int Cin = a.C1 + a.C2;

  // A: this thread's row quad (fixed) -> tap + channel
  const int arq = tid % AQ, apx = tid / AQ;
  const int arow = m0 + 4 * arq;
  const bool arow_ok = arow < a.M;
  int at = 0, ac = 0;
  if (arow_ok) { at = arow / Cin; ac = arow - at * Cin; }
  const int tb = (int)((a.tap_bits >> (6 * at)) & 63ull);
  const int ady = (tb & 7) - 2, adx = (tb >> 3) - 2;
  const float* asrc; int ald, acc_;
  if (ac < a.C1) { asrc = a.in1; ald = a.ld1; acc_ = ac; } else { asrc = a.in2; ald = a.ld2; acc_ = ac - a.C1; }
  // B: column quad
  const int bcq = tid % BQ, bpx = tid / BQ;
  const bool bcol_ok = (tid < BKP * BQ) && (n0 + 4 * bcq < a.Cout);     // Cout padded to 4 in ldd

  const long long step0 = (long long)blockIdx.z * a.steps_per_split;
  long long nst = (a.Ptot + BKP - 1) / BKP - step0;
  if (nst > a.steps_per_split) nst = a.steps_per_split;
  const int nsteps = nst > 0 ? (int)nst : 0;

  f32x4 ra[A_LD], rb[B_LD];
  const int GHW = a.GH * a.GW;

  auto load_step = [&](int s) {
    const long long pbase = (step0 + s) * BKP;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const long long P = pbase + apx + j * A_PSTEP;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (arow_ok && P < a.Ptot) {
        const int n = (int)(P / GHW); const int rem = (int)(P - (long long)n * GHW);
        const int gy = rem / a.GW, gx = rem - gy * a.GW;
        const int iy = gy * a.in_sy + ady, ix = gx * a.in_sx + adx;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
          v = *(const f32x4*)(asrc + ((size_t)(n * a.H + iy) * a.W + ix) * ald + acc_);
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const long long P = pbase + bpx + j * B_PSTEP;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (bcol_ok && P < a.Ptot) v = *(const f32x4*)(a.dout + (size_t)P * a.ldd + n0 + 4 * bcq);
      rb[j] = v;
    }
  };
  auto store_step = [&](int buf) {
    float* Ab = As + buf * BKP * LDA;
    float* Bb = Bs + buf * BKP * LDB;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) *(f32x4*)(Ab + (apx + j * A_PSTEP) * LDA + 4 * arq) = ra[j];
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      if (tid < BKP * BQ) *(f32x4*)(Bb + (bpx + j * B_PSTEP) * LDB + 4 * bcq) = rb[j];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nsteps > 0) {
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    const bool more = (s + 1) < nsteps;
    if (more) load_step(s + 1);
    const float* Ab = As + buf * BKP * LDA + half * LDA + wm * WTM + l31;
    const float* Bb = Bs + buf * BKP * LDB + half * LDB + wn * WTN + l31;
#pragma unroll
    for (int kk = 0; kk < BKP / 2; ++kk) {
      float fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = Ab[2 * kk * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = Bb[2 * kk * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_step(buf ^ 1);
    __syncthreads();
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

struct RedArgs {
  const float* ws; float* dw; int splits, M, Cout, Cin, Cin_real, KH, KW, ntaps;
  int ky[SSG_MAX_TAPS], kx[SSG_MAX_TAPS];
};

// 32 gradient elements x ZL split-lanes per block: lane z adds slabs z, z+ZL, ... in order, then the
// ZL lane sums are added in lane order -> fixed summation order, short serial chains.
template <int ZL>
__global__ __launch_bounds__(32 * ZL) void wgrad_reduce_kernel(const RedArgs a) {
  __shared__ float red[ZL][32];
  const int el = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const long long idx = (long long)blockIdx.x * 32 + el;
  const long long tot = (long long)a.M * a.Cout;
  float s = 0.f;
  if (idx < tot) {
    // four independent loads in flight per lane; the additions stay in slab order
    int z = zl;
    for (; z + 3 * ZL < a.splits; z += 4 * ZL) {
      const float v0 = a.ws[(size_t)z * tot + idx], v1 = a.ws[(size_t)(z + ZL) * tot + idx];
      const float v2 = a.ws[(size_t)(z + 2 * ZL) * tot + idx], v3 = a.ws[(size_t)(z + 3 * ZL) * tot + idx];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; z < a.splits; z += ZL) s += a.ws[(size_t)z * tot + idx];
  }
  red[zl][el] = s;
  __syncthreads();
  if (zl != 0 || idx >= tot) return;
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < ZL; ++k) v += red[k][el];
  const int row = (int)(idx / a.Cout), co = (int)(idx - (long long)row * a.Cout);
  const int t = row / a.Cin, c = row - t * a.Cin;
  if (c >= a.Cin_real) return;
  a.dw[(((size_t)co * a.Cin_real + c) * a.KH + a.ky[t]) * a.KW + a.kx[t]] = v;
}

struct Plan { int variant, mt, nt, splits, steps_per_split, halo; };

// thin weight gradients on the 4x4x1 MFMA (conv_wgrad4.hip); SSG_WGRAD4=0 switches them off (A/B)
int wgrad4_kind(const ssg_wgrad_desc* d) {
  static const int on = [] { const char* e = getenv("SSG_WGRAD4"); return e ? atoi(e) : 1; }();
  return on ? ssg_wgrad4_kind(d) : 0;
}

bool wgrad_uses_dma(int variant) {
  static const int use_dma = [] { const char* e = getenv("SSG_WGRAD_DMA"); return e ? atoi(e) : 1; }();
  return use_dma && variant <= 1;
}

// LDS-resident pixel window (conv_wgrad_halo.hip) for the 3x3 window; SSG_WGRAD_HALO=0 switches it off (A/B)
bool wgrad_uses_halo(const ssg_wgrad_desc* d, int variant) {
  static const int on = [] { const char* e = getenv("SSG_WGRAD_HALO"); return e ? atoi(e) : 1; }();
  return on && variant <= 1 && wgrad_uses_dma(variant) && ssg_wgrad_halo_ok(d, variant);
}

Plan make_plan(const ssg_wgrad_desc* d) {
  Plan p;
  p.halo = 0;
  const int Cin = d->C1 + d->C2;
  const int M = d->ntaps * Cin;
  int bn;
  if (d->Cout > 64) { p.variant = 0; bn = 128; }
  else if (d->Cout > 32) { p.variant = 1; bn = 64; }
  else { p.variant = 2; bn = 32; }
  p.mt = (M + 127) / 128;
  p.nt = (d->Cout + bn - 1) / bn;
  long long steps = ((long long)d->N * d->GH * d->GW + BKP - 1) / BKP;
  if (wgrad_uses_halo(d, p.variant)) {
    // M tile = 9 taps x CB channels; a K-step = 16 pixels inside one image row
    p.halo = 1;
    p.mt = Cin / ssg_wgrad_halo_cb(p.variant);
    steps = (long long)d->N * d->GH * ((d->GW + BKP - 1) / BKP);
  }
  if ((d->flags & 1) && ssg_wgrad_k32_ok(d)) {
    // conv_wgrad_k32.hip: 512-thread workgroups (one per CU) on 9 x 64 x 64 tiles, a K-step = one image row of a 32-pixel column strip
    p.halo = 2;
    p.mt = Cin / 64; p.nt = d->Cout / 64;
    steps = ssg_wgrad_k32_steps(d);
    const long long tiles = (long long)p.mt * p.nt;
    // Slabs: a slab is one workgroup's accumulation chain, so its length bounds the fp32 rounding error -- at most 128 rows (4096
    // pixels, the slab of wgrad_halo_x3 at the bench sizes; in-register totals that cut the chain instead cost 72 registers and
    // 10 % of the kernel: DESIGN.md 3.10) and at least 8 rows (256 pixels, the shortest slab of the fp32 kernels); within that,
    // the count that leaves the fewest CUs idle in the last wave of 256 workgroups (one workgroup per CU)
    // ... and never longer than the chain the 16-pixel-step kernels would use on this shape (their planner below: ~1024 workgroups,
    // >= 16 steps per slab), so that the rounding error of the two families stays equal on every shape, not only at the bench sizes
    long long max_rows = 128;
    {
      const long long steps16 = (long long)d->N * d->GH * ((d->GW + BKP - 1) / BKP);
      const long long mtf = Cin / ssg_wgrad_halo_cb(p.variant), ntf = (d->Cout + (p.variant == 0 ? 127 : 63)) / (p.variant == 0 ? 128 : 64);
      long long wantf = 1024 / (mtf * ntf > 0 ? mtf * ntf : 1);
      if (wantf < 1) wantf = 1;
      if (wantf > steps16 / 16) wantf = steps16 / 16 > 0 ? steps16 / 16 : 1;
      if (wantf > 512) wantf = 512;
      const long long chain_px = (steps16 + wantf - 1) / wantf * BKP;
      if (chain_px / 32 < max_rows) max_rows = chain_px / 32;
      if (max_rows < 8) max_rows = 8;
    }
    const long long lo = ssg_wgrad_k32_flush() == 0 ? (steps + max_rows - 1) / max_rows : 1;
    long long hi = steps / 8;
    if (hi < lo) hi = lo;
    long long best = lo; double beff = 0;
    for (long long sp = lo; sp <= hi && sp <= lo + 1024; ++sp) {
      const long long wg = sp * tiles;
      const double eff = (double)wg / (256.0 * ((wg + 255) / 256));
      if (eff > beff + 1e-9) { beff = eff; best = sp; }
      if (eff >= 0.97 && wg >= 256) break;
    }
    p.steps_per_split = (int)((steps + best - 1) / best);
    p.splits = (int)((steps + p.steps_per_split - 1) / p.steps_per_split);
    return p;
  }
  static const int wgs = [] { const char* e = getenv("SSG_WGRAD_WGS"); return e ? atoi(e) : 1024; }();
  long long want = wgs / ((long long)p.mt * p.nt);       // ~4 workgroups per CU overall (2048: +0.4 % slab traffic time)
  if (want < 1) want = 1;
  long long maxs = steps / 16;                           // at least 16 K-steps (256 pixels) per split
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want > 512) want = 512;
  p.steps_per_split = (int)((steps + want - 1) / want);
  p.splits = (int)((steps + p.steps_per_split - 1) / p.steps_per_split);
  return p;
}

int validate(const ssg_wgrad_desc* d) {
  SSG_REQUIRE(d && d->in1 && d->dout && d->dw_oihw, SSG_EINVAL, "wgrad: null pointer");
  SSG_REQUIRE(d->C1 > 0 && d->C1 % 4 == 0 && d->C2 >= 0 && d->C2 % 4 == 0, SSG_EINVAL, "wgrad: channels must be multiples of 4");
  SSG_REQUIRE(d->C2 == 0 || d->in2, SSG_EINVAL, "wgrad: C2 > 0 needs in2");
  SSG_REQUIRE(d->ld1 % 4 == 0 && d->ld1 >= d->C1 && d->ldd % 4 == 0 && d->ldd >= d->Cout, SSG_EALIGN, "wgrad: strides");
  SSG_REQUIRE(d->C2 == 0 || (d->ld2 % 4 == 0 && d->ld2 >= d->C2), SSG_EALIGN, "wgrad: ld2");
  SSG_REQUIRE(ssg_aligned16(d->in1) && ssg_aligned16(d->in2) && ssg_aligned16(d->dout) && ssg_aligned16(d->ws), SSG_EALIGN, "wgrad: alignment");
  SSG_REQUIRE(d->ntaps >= 1 && d->ntaps <= SSG_MAX_TAPS, SSG_EINVAL, "wgrad: ntaps");
  SSG_REQUIRE(d->Cin_real > 0 && d->Cin_real <= d->C1 + d->C2, SSG_EINVAL, "wgrad: Cin_real");
  for (int t = 0; t < d->ntaps; ++t) {
    SSG_REQUIRE(d->dy[t] >= -2 && d->dy[t] <= 5 && d->dx[t] >= -2 && d->dx[t] <= 5, SSG_EINVAL, "wgrad: tap offset");
    SSG_REQUIRE(d->ky[t] >= 0 && d->ky[t] < d->KH && d->kx[t] >= 0 && d->kx[t] < d->KW, SSG_EINVAL, "wgrad: kernel position");
  }
  SSG_REQUIRE(d->ws_bytes >= ssg_conv2d_wgrad_workspace_bytes(d), SSG_EINVAL, "wgrad: workspace too small");
  return SSG_OK;
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int O, int I, int KH, int KW,
                                                           int transpose, int ntaps, unsigned long long kpos_bits,
                                                           int kmode, int Cred_pad, int Kp, int R, float* __restrict__ out,
                                                           const float* __restrict__ sigma) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)R * Kp) return;
  const int r = (int)(idx / Kp), k = (int)(idx - (long long)r * Kp);
  int t, c;
  if (kmode == 0) {
    const int span = ntaps * 16;
    const int chunk = k / span, rem = k - chunk * span;
    t = rem >> 4; c = chunk * 16 + (rem & 15);
  } else {
    t = k / Cred_pad; c = k - t * Cred_pad;
  }
  const int Cred = transpose ? O : I;
  float v = 0.f;
  if (t < ntaps && c < Cred) {
    const int kb = (int)((kpos_bits >> (6 * t)) & 63ull);
    const int ky = kb & 7, kx = kb >> 3;
    const int o = transpose ? c : r, i = transpose ? r : c;
    v = w[(((size_t)o * I + i) * KH + ky) * KW + kx];
    if (sigma) v *= 1.f / *sigma;            // spectral norm: W / sigma never exists in OIHW form, the scale rides the pack
  }
  out[idx] = v;
}

}  // namespace

extern "C" int64_t ssg_conv2d_wgrad_workspace_bytes(const ssg_wgrad_desc* d) {
  if (!d) return 0;
  if (wgrad4_kind(d)) {
    const int nz = ssg_wgrad4_slices(d, wgrad4_kind(d), nullptr, nullptr, nullptr);
    return (int64_t)nz * d->ntaps * (d->C1 + d->C2) * d->Cout * (int64_t)sizeof(float);
  }
  const Plan p = make_plan(d);
  return (int64_t)p.splits * d->ntaps * (d->C1 + d->C2) * d->Cout * (int64_t)sizeof(float);
}

// the x operand as a fused batch-norm apply (ssg_wgrad_desc.in_scale): the k32 kernel only, one input pointer
extern "C" int ssg_conv2d_wgrad_in_affine_ok(const ssg_wgrad_desc* d) {
  if (!d || !d->in1 || !d->dout || wgrad4_kind(d)) return 0;     // asked before the workspace exists: no validate() here
  if (d->C2 != 0 || (d->in_act != SSG_ACT_NONE && d->in_act != SSG_ACT_RELU && d->in_act != SSG_ACT_LRELU)) return 0;
  return make_plan(d).halo == 2 ? 1 : 0;
}

extern "C" int ssg_conv2d_wgrad_f32(const ssg_wgrad_desc* d, void* stream) {
  int rc = validate(d);
  if (rc != SSG_OK) return rc;
  SSG_REQUIRE(!d->in_scale || (d->in_shift && ssg_conv2d_wgrad_in_affine_ok(d)), SSG_EINVAL,
              "wgrad: in_scale on a descriptor whose kernel has no fused input transform (ssg_conv2d_wgrad_in_affine_ok == 0)");
  Plan p = make_plan(d);
  hipStream_t st = (hipStream_t)stream;
  const int w4 = wgrad4_kind(d);
  WgArgs a;
  a.in1 = d->in1; a.in2 = d->C2 ? d->in2 : d->in1; a.dout = d->dout; a.ws = d->ws;
  a.C1 = d->C1; a.C2 = d->C2; a.ld1 = d->ld1; a.ld2 = d->C2 ? d->ld2 : d->ld1;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.ldd = d->ldd; a.GH = d->GH; a.GW = d->GW;
  a.in_sy = d->in_sy; a.in_sx = d->in_sx; a.ntaps = d->ntaps;
  a.tap_bits = 0;
  for (int t = 0; t < d->ntaps; ++t)
    a.tap_bits |= (unsigned long long)(((d->dy[t] + 2) & 7) | (((d->dx[t] + 2) & 7) << 3)) << (6 * t);
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.in_act = d->in_act; a.in_slope = d->in_slope;
  a.M = d->ntaps * (d->C1 + d->C2);
  a.Ptot = (long long)d->N * d->GH * d->GW;
  a.steps_per_split = p.steps_per_split;
  if (w4) {
    p.splits = ssg_wgrad4_slices(d, w4, nullptr, nullptr, nullptr);
    rc = ssg_wgrad4_launch(d, w4, st);
    if (rc != SSG_OK) return rc;
  } else {
    dim3 grid((unsigned)p.mt, (unsigned)p.nt, (unsigned)p.splits);
    if (p.halo == 2) {
      rc = ssg_wgrad_k32_launch(a, grid, st);
      if (rc != SSG_OK) return rc;
    } else if (p.halo) {
      rc = ssg_wgrad_halo_launch(a, p.variant, grid, st, (d->flags & 1) != 0);
      if (rc != SSG_OK) return rc;
    } else if (wgrad_uses_dma(p.variant)) {
      rc = ssg_wgrad_dma_launch(a, p.variant, grid, st, (d->flags & 1) != 0);
      if (rc != SSG_OK) return rc;
    } else
    switch (p.variant) {
      case 0: hipLaunchKernelGGL((wgrad_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, a); break;
      case 1: hipLaunchKernelGGL((wgrad_kernel<128, 64, 2, 2>), grid, dim3(256), 0, st, a); break;
      default: hipLaunchKernelGGL((wgrad_kernel<128, 32, 4, 1>), grid, dim3(256), 0, st, a); break;
    }
    SSG_LAUNCH_CHECK();
  }
  RedArgs r;
  r.ws = d->ws; r.dw = d->dw_oihw; r.splits = p.splits; r.M = a.M; r.Cout = d->Cout; r.Cin = d->C1 + d->C2;
  r.Cin_real = d->Cin_real; r.KH = d->KH; r.KW = d->KW; r.ntaps = d->ntaps;
  for (int t = 0; t < SSG_MAX_TAPS; ++t) { r.ky[t] = t < d->ntaps ? d->ky[t] : 0; r.kx[t] = t < d->ntaps ? d->kx[t] : 0; }
  const long long tot = (long long)a.M * d->Cout;
  // many slabs over few elements (the thin layers): 32 split-lanes keep the serial chains short
  if (p.splits >= 256 && tot <= 32 * 1024)
    hipLaunchKernelGGL(wgrad_reduce_kernel<32>, dim3((unsigned)ssg_cdiv(tot, 32)), dim3(1024), 0, st, r);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((unsigned)ssg_cdiv(tot, 32)), dim3(256), 0, st, r);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_pack_weights_scaled_f32(const float* w_oihw, int O, int I, int KH, int KW, int transpose, int ntaps,
                                           const int* ky, const int* kx, int kmode, int Cred_pad, int Kp, const float* sigma, float* out,
                                           void* stream);
extern "C" int ssg_pack_weights_f32(const float* w_oihw, int O, int I, int KH, int KW, int transpose, int ntaps,
                                    const int* ky, const int* kx, int kmode, int Cred_pad, int Kp, float* out,
                                    void* stream) {
  return ssg_pack_weights_scaled_f32(w_oihw, O, I, KH, KW, transpose, ntaps, ky, kx, kmode, Cred_pad, Kp, nullptr, out, stream);
}
extern "C" int ssg_pack_weights_scaled_f32(const float* w_oihw, int O, int I, int KH, int KW, int transpose, int ntaps,
                                           const int* ky, const int* kx, int kmode, int Cred_pad, int Kp, const float* sigma, float* out,
                                           void* stream) {
  SSG_REQUIRE(w_oihw && out && ky && kx, SSG_EINVAL, "pack: null pointer");
  SSG_REQUIRE(ntaps >= 1 && ntaps <= SSG_MAX_TAPS && KH <= 8 && KW <= 8, SSG_EINVAL, "pack: taps");
  SSG_REQUIRE(Kp % 16 == 0 && Cred_pad % 4 == 0, SSG_EINVAL, "pack: Kp/Cred_pad");
  const int Cred = transpose ? O : I;
  SSG_REQUIRE(Cred_pad >= Cred, SSG_EINVAL, "pack: Cred_pad < reduced channels");
  if (kmode == 0) SSG_REQUIRE(Cred_pad % 16 == 0 && Kp == ntaps * Cred_pad, SSG_EINVAL, "pack: kmode 0 shape");
  else SSG_REQUIRE(Kp >= ntaps * Cred_pad, SSG_EINVAL, "pack: kmode 1 Kp");
  unsigned long long bits = 0;
  for (int t = 0; t < ntaps; ++t) {
    SSG_REQUIRE(ky[t] >= 0 && ky[t] < KH && kx[t] >= 0 && kx[t] < KW, SSG_EINVAL, "pack: kernel position");
    bits |= (unsigned long long)(ky[t] | (kx[t] << 3)) << (6 * t);
  }
  const int R = transpose ? I : O;
  const long long tot = (long long)R * Kp;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)ssg_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                     w_oihw, O, I, KH, KW, transpose, ntaps, bits, kmode, Cred_pad, Kp, R, out, sigma);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

// which kernel a wgrad descriptor maps to: 0..2 = wgrad<128,128>/<128,64>/<128,32>, 20/21 = wgrad_dma<128,128>/<128,64>, 30/31 = wgrad_halo<32,128>/<64,64>, 40/41 / 50/51 = the split-operand (x3) forms of 30/31 / 20/21, 60 = wgrad_k32 (conv_wgrad_k32.hip), 15/16 = wgrad4 (4x4x1 MFMA)
extern "C" int ssg_conv2d_wgrad_kernel_id(const ssg_wgrad_desc* d) {
  if (!d) return SSG_EINVAL;
  if (wgrad4_kind(d)) return 10 + wgrad4_kind(d);
  if (make_plan(d).halo == 2) return 60;
  if (make_plan(d).halo) return ((d->flags & 1) ? 40 : 30) + make_plan(d).variant;
  const int v = make_plan(d).variant;
  return v + (wgrad_uses_dma(v) ? ((d->flags & 1) ? 50 : 20) : 0);
}
