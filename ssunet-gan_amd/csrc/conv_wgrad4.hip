// Weight gradients of "thin" convolutions on v_mfma_f32_4x4x1_16B_f32 (gfx950).
//
// When one side of a conv has <= 4 channels (SPADE's C->3 / 3->h / h->C chain, the 3-channel image,
// logit and mask layers) a 32x32 MFMA tile pads the thin side 8-10x.  The 4x4x1 MFMA computes 16
// independent 4x4 outer products per instruction at the same FLOP rate, which is exactly this shape:
//     D_b[i][j] += A[4b+i] * B[4b+j]      (b = 0..15; measured on gfx950: D_b[i][j] sits in lane 4b+j, reg i)
//   thin-Cout (dout has <= 4 channels, in has C): A = in[p + tap][c0 + lane], B = dout[p][lane % 4]
//        -> lane 4b+j, reg i holds dw[co = j][tap][c = c0 + 4b + i]
//   thin-Cin  (in has 4 channels, dout has Cout): A = dout[p][co0 + lane],    B = in[p + tap][lane % 4]
//        -> lane 4b+j, reg i holds dw[co = co0 + 4b + i][tap][c = j]
// One wave owns a 64-channel group of the wide tensor and all taps (9 accumulators x 4 registers),
// streams pixels with whole 256-B rows per load, and ends with one slab write; slabs are summed by the
// ordered reducer of conv_wgrad.hip (bitwise reproducible, no atomics).  HBM-bound by construction.
#include "common.h"
#include "conv_thin.h"

namespace {

struct W4Args {
  const float* wide; const float* thin; float* ws;
  int Cw, ldw, ldt, N, H, W, Cout, Cin;
  int groups, nz;            // 64-channel groups of the wide tensor; slabs (= workgroups per group)
  long long units_per_z;     // row segments per slice
  int segs_per_row;
};

constexpr int SEG = 32;      // pixels per work unit (one row segment)
constexpr int U = 8;         // pixels per inner step: (U+KS-1)*KS + U loads, then U*KS*KS MFMAs

// KS = 3: taps t = (dy+1)*3 + (dx+1), pad 1;  KS = 1: the single tap (0,0).
template <bool THIN_COUT, int KS>
__global__ __launch_bounds__(256) void wgrad4_kernel(const W4Args a) {
  constexpr int R = KS / 2, NT = KS * KS, COLS = U + KS - 1;
  const int lane = threadIdx.x & 63;
  // the 4 waves of a workgroup share the channel group and own 4 consecutive pixel slices; their
  // accumulators are summed through LDS in wave order, so one workgroup writes one slab
  const int wave = threadIdx.x >> 6;
  const int zb = blockIdx.x / a.groups, cg = blockIdx.x - zb * a.groups;
  const int z = zb * 4 + wave;
  const int j = lane & 3;
  const int cw = cg * 64 + lane;                 // this lane's channel of the wide tensor
  const bool cw_ok = cw < a.Cw;
  // the tap-shifted tensor ("in") and the fixed one ("dout") are read through buffer descriptors: an
  // out-of-image tap / padded lane gets byte offset 0xffffffff, which the range check turns into 0.0f --
  // no branches, no selects (tensors are < 4 GiB, checked on the host)
  const float* tapT = THIN_COUT ? a.wide : a.thin;
  const float* fixT = THIN_COUT ? a.thin : a.wide;
  const unsigned ldtap = THIN_COUT ? a.ldw : a.ldt, ldfix = THIN_COUT ? a.ldt : a.ldw;
  const unsigned npix = (unsigned)(a.N * a.H * a.W);
  const auto tap_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tapT), 0, (int)(npix * ldtap * 4u), 0x00020000);
  const auto fix_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fixT), 0, (int)(npix * ldfix * 4u), 0x00020000);
  const unsigned chtap = THIN_COUT ? cw : j, chfix = THIN_COUT ? j : cw;
  const bool tap_lane_ok = THIN_COUT ? cw_ok : true, fix_lane_ok = THIN_COUT ? true : cw_ok;
  constexpr unsigned OOB = 0xffffffffu;

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int u0 = z * (int)a.units_per_z;
  const int total_units = a.N * a.H * a.segs_per_row;
  int u1 = u0 + (int)a.units_per_z; if (u1 > total_units) u1 = total_units;
  for (int u = u0; u < u1; ++u) {
    const int r = u / a.segs_per_row; const int seg = u - r * a.segs_per_row;
    const int y = r % a.H, n = r / a.H;
    const int x0 = seg * SEG;
    const int x1 = x0 + SEG < a.W ? x0 + SEG : a.W;
    unsigned rowoff[KS]; bool rowok[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int iy = y + q - R;
      rowok[q] = tap_lane_ok && (unsigned)iy < (unsigned)a.H;
      rowoff[q] = ((unsigned)((n * a.H + iy) * a.W) * ldtap + chtap) * 4u;
    }
    const unsigned fixoff = ((unsigned)((n * a.H + y) * a.W) * ldfix + chfix) * 4u;
    for (int xb = x0; xb < x1; xb += U) {
      float f[U], v[KS][COLS];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const bool ok = fix_lane_ok && (xb + k < x1);
        f[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            fix_rs, ok ? fixoff + (unsigned)(xb + k) * ldfix * 4u : OOB, 0, 0));
      }
#pragma unroll
      for (int c = 0; c < COLS; ++c) {
        const int ix = xb + c - R;
        const bool cok = (unsigned)ix < (unsigned)a.W;
        const unsigned xo = (unsigned)ix * ldtap * 4u;
#pragma unroll
        for (int q = 0; q < KS; ++q)
          v[q][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
              tap_rs, (cok && rowok[q]) ? rowoff[q] + xo : OOB, 0, 0));
      }
#pragma unroll
      for (int k = 0; k < U; ++k)
#pragma unroll
        for (int q = 0; q < KS; ++q)
#pragma unroll
          for (int e = 0; e < KS; ++e) {
            if (THIN_COUT) acc[q * KS + e] = __builtin_amdgcn_mfma_f32_4x4x1f32(v[q][k + e], f[k], acc[q * KS + e], 0, 0, 0);
            else acc[q * KS + e] = __builtin_amdgcn_mfma_f32_4x4x1f32(f[k], v[q][k + e], acc[q * KS + e], 0, 0, 0);
          }
    }
  }
  __shared__ float red[4][NT * 4][64];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][t * 4 + i][lane] = acc[t][i];
  __syncthreads();
  // slab layout of conv_wgrad.hip: [slab][row = t*Cin + c][Cout]
  float* slab = a.ws + (size_t)zb * NT * a.Cin * a.Cout;
  const int b4 = (lane >> 2) * 4;
  for (int e = wave; e < NT * 4; e += 4) {
    const int t = e >> 2, i = e & 3;
    const float sum = ((red[0][e][lane] + red[1][e][lane]) + red[2][e][lane]) + red[3][e][lane];
    int co, c;
    if (THIN_COUT) { co = j; c = cg * 64 + b4 + i; }
    else { co = cg * 64 + b4 + i; c = j; }
    if (co < a.Cout && c < a.Cin) slab[((size_t)t * a.Cin + c) * a.Cout + co] = sum;
  }
}

// ------------------------------------------------------------------ thin-Cin (in has 4 channels, 3x3) on the 32x32x2 MFMA
// dw[co][tap][c] = sum_p dout[p][co] * in[p + tap][c] as a GEMM with M = 64 output channels (two fragments), N = (tap, c) = 36
// columns (two fragments, the second holds tap 8 only) and K = pixels, two per MFMA: A = dout[p + h][co0 + m] (one dword per
// lane: lanes 0-31 / 32-63 read one whole 128-byte line each), B = in[p + h + tap(n)][c(n)].  8 instructions per 2 pixels
// instead of the 4x4x1 kernel's 110 per 8 (which runs at 2.2 TB/s of dout): the same slab / ordered-reduce protocol.
__global__ __launch_bounds__(256) void wgrad32_cin_kernel(const W4Args a) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int zb = blockIdx.x / a.groups, cg = blockIdx.x - zb * a.groups;
  const int z = zb * 4 + wave;
  const unsigned npix = (unsigned)(a.N * a.H * a.W);
  const auto do_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wide), 0, (int)(npix * (unsigned)a.ldw * 4u), 0x00020000);
  const auto in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.thin), 0, (int)(npix * (unsigned)a.ldt * 4u), 0x00020000);
  constexpr unsigned OOB = 0xffffffffu;
  const unsigned ldwb = (unsigned)a.ldw * 4u, ldtb = (unsigned)a.ldt * 4u;
  // A: this lane's two output channels; B: this lane's (tap, channel) columns n = l31 and 32 + l31 (only taps 0..8 exist)
  unsigned aoff[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) { const int co = cg * 64 + f * 32 + l31; aoff[f] = co < a.Cout ? (unsigned)co * 4u : OOB; }
  const int t0 = l31 >> 2, c = l31 & 3;                  // column n = l31: tap 0..7; column 32 + l31: tap 8 for l31 < 4
  const int dy0 = t0 / 3 - 1, dx0 = t0 % 3 - 1;
  const bool has1 = l31 < 4;

  f32x16 acc[2][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[f][g][q] = 0.f;

  const int u0 = z * (int)a.units_per_z;
  const int total_units = a.N * a.H * a.segs_per_row;
  int u1 = u0 + (int)a.units_per_z; if (u1 > total_units) u1 = total_units;
  constexpr int UP = 8;                                  // pixel pairs per inner step (16: 172 registers, 2 waves per SIMD, 3 % slower)
  for (int u = u0; u < u1; ++u) {
    const int r = u / a.segs_per_row; const int seg = u - r * a.segs_per_row;
    const int y = r % a.H, n = r / a.H;
    const int x0 = seg * SEG;
    const int x1 = x0 + SEG < a.W ? x0 + SEG : a.W;
    const unsigned rowpix = (unsigned)((n * a.H + y) * a.W);
    const bool r0ok = (unsigned)(y + dy0) < (unsigned)a.H, r1ok = has1 && y + 1 < a.H;
    const unsigned b0row = (rowpix + (unsigned)(dy0 * a.W)) * ldtb + (unsigned)c * 4u;
    const unsigned b1row = (rowpix + (unsigned)a.W) * ldtb + (unsigned)c * 4u;
    for (int xb = x0; xb < x1; xb += 2 * UP) {
      float av[2][UP], bv[2][UP];
#pragma unroll
      for (int k = 0; k < UP; ++k) {
        const int x = xb + 2 * k + h;
        const bool pok = x < x1;
        const unsigned po = (rowpix + (unsigned)x) * ldwb;
#pragma unroll
        for (int f = 0; f < 2; ++f)
          av[f][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(do_rs, (pok && aoff[f] != OOB) ? po + aoff[f] : OOB, 0, 0));
        const int ix0 = x + dx0, ix1 = x + 1;
        bv[0][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            in_rs, (pok && r0ok && (unsigned)ix0 < (unsigned)a.W) ? b0row + (unsigned)ix0 * ldtb : OOB, 0, 0));
        bv[1][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            in_rs, (pok && r1ok && ix1 < a.W) ? b1row + (unsigned)ix1 * ldtb : OOB, 0, 0));
      }
#pragma unroll
      for (int k = 0; k < UP; ++k)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int g = 0; g < 2; ++g)
            acc[f][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[f][k], bv[g][k], acc[f][g], 0, 0, 0);
    }
  }
  // fold the 4 waves through LDS in wave order; D[m = co][n]: lane holds column n = l31 (+32 g), rows (q&3) + 8*(q>>2) + 4*h (+32 f)
  __shared__ float red[4][36][64];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int nn = g * 32 + l31;
      if (nn < 36) {
#pragma unroll
        for (int q = 0; q < 16; ++q) red[wave][nn][f * 32 + (q & 3) + 8 * (q >> 2) + 4 * h] = acc[f][g][q];
      }
    }
  __syncthreads();
  float* slab = a.ws + (size_t)zb * 9 * a.Cin * a.Cout;          // [slab][row = t*Cin + c][Cout], Cin == 4: row = n
  for (int e = threadIdx.x; e < 36 * 64; e += 256) {
    const int nn = e >> 6, m = e & 63;
    const float sum = ((red[0][nn][m] + red[1][nn][m]) + red[2][nn][m]) + red[3][nn][m];
    const int co = cg * 64 + m;
    if (co < a.Cout) slab[(size_t)nn * a.Cout + co] = sum;
  }
}

// ------------------------------------------------------------------ tiny: in has 4 channels AND dout has <= 8 (SPADE's 3 -> nhidden
// conv, normalization.py:92).  VALU: thread (pixel slot k = tid / 8, output channel co = tid % 8) walks the workgroup's pixel
// range with stride 32, 36 accumulators dw[co][tap][c]; the 9 window loads are shared by the 8 channel threads of a pixel
// (same address).  The 32 pixel slots are folded through LDS in slot order, one slab per workgroup, ordered reduce after.
constexpr int TINY_PIX = 4096;        // pixels per workgroup

__global__ __launch_bounds__(256) void wgrad_tiny4_kernel(const W4Args a) {
  const int co = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const long long npix = (long long)a.N * a.H * a.W;
  const long long p0 = (long long)blockIdx.x * TINY_PIX;
  const long long p1 = p0 + TINY_PIX < npix ? p0 + TINY_PIX : npix;
  const auto in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.thin), 0, (int)((unsigned)npix * (unsigned)a.ldt * 4u), 0x00020000);
  const auto do_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wide), 0, (int)((unsigned)npix * (unsigned)a.ldw * 4u), 0x00020000);
  constexpr unsigned OOB = 0xffffffffu;
  const bool co_ok = co < a.Cout;
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (long long p = p0 + slot; p < p1; p += 32) {
    const int x = (int)(p % a.W); const int y = (int)((p / a.W) % a.H);
    const float g = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(do_rs, co_ok ? ((unsigned)p * (unsigned)a.ldw + (unsigned)co) * 4u : OOB, 0, 0));
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          in_rs, ok ? (unsigned)(p + (long long)(t / 3 - 1) * a.W + (t % 3 - 1)) * (unsigned)a.ldt * 4u : OOB, 0, 0));
      acc[t] += v * g;
    }
  }
  __shared__ float red[32][36][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) red[slot][t * 4 + c][co] = acc[t][c];
  __syncthreads();
  // slab layout of conv_wgrad.hip: [slab][row = t*Cin + c][Cout]
  float* slab = a.ws + (size_t)blockIdx.x * 9 * a.Cin * a.Cout;
  for (int e = threadIdx.x; e < 36 * 8; e += 256) {
    const int c8 = e & 7, row = e >> 3;
    float sum = 0.f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) sum += red[k][row][c8];
    const int t = row >> 2, c = row & 3;
    if (c8 < a.Cout && c < a.Cin) slab[((size_t)t * a.Cin + c) * a.Cout + c8] = sum;
  }
}

}  // namespace

// kind 5: dout thin (Cout <= 4), kind 6: in thin (Cin_pad == 4), kind 7: both (Cin_pad == 4, Cout <= 8; VALU),
// kind 8: in thin, 3x3, Cout >= 32 on >= 65536 pixels (32x32x2 MFMA)
int ssg_wgrad4_kind(const ssg_wgrad_desc* d) {
  if (d->C2 != 0 || d->in_sy != 1 || d->in_sx != 1 || d->GH != d->H || d->GW != d->W) return 0;
  if (d->ntaps == 9) {
    for (int t = 0; t < 9; ++t) if (d->dy[t] != t / 3 - 1 || d->dx[t] != t % 3 - 1) return 0;
  } else if (d->ntaps != 1 || d->dy[0] != 0 || d->dx[0] != 0) {
    return 0;
  }
  // 32-bit byte offsets (buffer descriptors) inside the kernel
  if ((long long)d->N * d->H * d->W * (d->ld1 > d->ldd ? d->ld1 : d->ldd) >= (1ll << 30)) return 0;
  if (d->Cout <= 4 && d->C1 >= 16 && d->C1 % 4 == 0) return 5;
  static const int tiny = [] { const char* e = getenv("SSG_TINY4"); return e ? atoi(e) : 1; }();
  if (tiny && d->C1 == 4 && d->Cout <= 8 && d->ntaps == 9) return 7;
  static const int w32 = [] { const char* e = getenv("SSG_WGRAD32"); return e ? atoi(e) : 1; }();
  if (w32 && d->C1 == 4 && d->ntaps == 9 && d->Cout >= 32 && (long long)d->N * d->H * d->W >= 65536) return 8;
  if (d->C1 == 4) return 6;
  return 0;
}

int ssg_wgrad4_slices(const ssg_wgrad_desc* d, int kind, int* groups, long long* units_per_z, int* segs_per_row) {
  if (kind == 7) {
    if (groups) *groups = 1;
    if (units_per_z) *units_per_z = TINY_PIX;
    if (segs_per_row) *segs_per_row = 1;
    return (int)(((long long)d->N * d->H * d->W + TINY_PIX - 1) / TINY_PIX);
  }
  const int cw = kind == 5 ? d->C1 : d->Cout;
  const int g = (cw + 63) / 64;
  const int spr = (d->W + SEG - 1) / SEG;
  const long long units = (long long)d->N * d->H * spr;
  long long nz = 8192 / g;                       // ~32 waves per CU in total
  if (nz > units) nz = units;
  if (nz < 1) nz = 1;
  const long long upz = (units + nz - 1) / nz;
  nz = (units + upz - 1) / upz;
  nz = (nz + 3) / 4;                             // slabs = workgroups per channel group (4 slices each)
  if (groups) *groups = g;
  if (units_per_z) *units_per_z = upz;
  if (segs_per_row) *segs_per_row = spr;
  return (int)nz;
}

int ssg_wgrad4_launch(const ssg_wgrad_desc* d, int kind, hipStream_t st) {
  W4Args a;
  a.ws = d->ws; a.N = d->N; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.Cin = d->C1;
  a.nz = ssg_wgrad4_slices(d, kind, &a.groups, &a.units_per_z, &a.segs_per_row);
  const dim3 grid((unsigned)(a.nz * a.groups)), block(256);
  if (kind == 7) {
    a.wide = d->dout; a.Cw = d->Cout; a.ldw = d->ldd; a.thin = d->in1; a.ldt = d->ld1;
    hipLaunchKernelGGL(wgrad_tiny4_kernel, grid, block, 0, st, a);
    SSG_LAUNCH_CHECK();
    return SSG_OK;
  }
  if (kind == 8) {
    a.wide = d->dout; a.Cw = d->Cout; a.ldw = d->ldd; a.thin = d->in1; a.ldt = d->ld1;
    hipLaunchKernelGGL(wgrad32_cin_kernel, grid, block, 0, st, a);
    SSG_LAUNCH_CHECK();
    return SSG_OK;
  }
  if (kind == 5) { a.wide = d->in1; a.Cw = d->C1; a.ldw = d->ld1; a.thin = d->dout; a.ldt = d->ldd; }
  else { a.wide = d->dout; a.Cw = d->Cout; a.ldw = d->ldd; a.thin = d->in1; a.ldt = d->ld1; }
  if (kind == 5 && d->ntaps == 9) hipLaunchKernelGGL((wgrad4_kernel<true, 3>), grid, block, 0, st, a);
  else if (kind == 5) hipLaunchKernelGGL((wgrad4_kernel<true, 1>), grid, block, 0, st, a);
  else if (d->ntaps == 9) hipLaunchKernelGGL((wgrad4_kernel<false, 3>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((wgrad4_kernel<false, 1>), grid, block, 0, st, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
