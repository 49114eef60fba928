// 1x1 convolutions with 64 input channels as a streaming GEMM on v_mfma_f32_32x32x2_f32 (gfx950).
//
// The BasicBlock shortcuts at the 512^2 / 256^2 levels (archs.py:218: 64 -> 64 / 128) and their input gradients have K = 64:
// on the LDS-DMA kernel a 128-pixel tile is 4 K-steps long, i.e. all prologue and epilogue (55 TFLOP/s, 2x its HBM time).
// Here the weights of a 64-channel output group live in registers (2 x 32 per lane), a wave streams 32-pixel blocks:
//   A = x[p0 + m][32h + s]   lane (m = lane&31, h = lane>>5) loads its half pixel row with 8 x 16-byte loads -- the wave reads
//                            32 pixels x 256 B = 8 KiB contiguous; K-step s pairs channel s (h = 0) with channel 32 + s (h = 1)
//   B = w[co][32h + s]       32 registers per N-fragment, loaded once per wave
// 64 MFMAs per block, then the 32x32 accumulators leave each lane 16 pixels of one channel: 32 dword stores whose lane
// halves cover one 128-byte line each (bias / residual / activation in registers).  Memory goes through buffer descriptors:
// rows past the end get offset 0xffffffff (dropped by the range check) instead of a branch.
#include "common.h"
#include "conv_thin.h"
#include <stdlib.h>

namespace {

struct C11Args {
  const float* in; const float* w; const float* bias; const float* res; float* out;
  int ld, ldr, ldo, Kp, Cout, act; float slope;
  long long P;
  int groups, blocks_per_wave, nt_store;
};

constexpr unsigned OOB = 0xffffffffu;

template <bool HAS_RES, bool NT>
__global__ __launch_bounds__(256) void conv1x1_k64_kernel(const C11Args a) {
  constexpr int AUX = NT ? 2 : 0;
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int cg = wid % a.groups; const long long wv = wid / a.groups;
  const unsigned npix = (unsigned)a.P;
  const auto in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)(npix * (unsigned)a.ld * 4u), 0x00020000);
  const auto out_rs = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(npix * (unsigned)a.ldo * 4u), 0x00020000);
  const auto res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_RES ? a.res : a.in), 0,
                                                        (int)(npix * (unsigned)(HAS_RES ? a.ldr : a.ld) * 4u), 0x00020000);
  const unsigned ldb = (unsigned)a.ld * 4u, ldob = (unsigned)a.ldo * 4u, ldrb = (unsigned)a.ldr * 4u;
  float wB[2][32], bv[2], keep[2];
  unsigned vo_out[2], vo_res[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int co = cg * 64 + f * 32 + l31;
    const bool ok = co < a.Cout;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 w4 = ok ? *(const f32x4*)(a.w + (size_t)co * a.Kp + 32 * h + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) wB[f][4 * q + e] = w4[e];
    }
    bv[f] = (a.bias && ok) ? a.bias[co] : 0.f;
    keep[f] = ok ? 1.f : 0.f;
    const bool cok = co < ((a.Cout + 3) & ~3);
    vo_out[f] = cok ? 4u * (unsigned)h * ldob + (unsigned)co * 4u : OOB;
    vo_res[f] = cok ? 4u * (unsigned)h * ldrb + (unsigned)co * 4u : OOB;
  }
  const bool is_relu = a.act == SSG_ACT_RELU;
  const float neg_slope = a.act == SSG_ACT_LRELU ? a.slope : 1.f;
  const unsigned aoff = (unsigned)l31 * ldb + 128u * (unsigned)h;            // this lane's half pixel row inside a block

  const long long b0 = wv * a.blocks_per_wave;
  const long long nblk = (a.P + 31) / 32;
  long long b1 = b0 + a.blocks_per_wave; if (b1 > nblk) b1 = nblk;
  for (long long blk = b0; blk < b1; ++blk) {
    const unsigned p0 = (unsigned)(blk * 32);
    const bool full = p0 + 32 <= npix;
    const bool mine = p0 + (unsigned)l31 < npix;
    f32x4 xa[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      xa[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rs, mine ? p0 * ldb + aoff + 16u * q : OOB, 0, 0));
    float rv[2][16];
    if (HAS_RES && full) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          rv[f][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
              res_rs, vo_res[f], (int)((p0 + (unsigned)((q & 3) + 8 * (q >> 2))) * ldrb), 0));
    }
    f32x16 acc[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[f][q] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s)
#pragma unroll
      for (int f = 0; f < 2; ++f)
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s >> 2][s & 3], wB[f][s], acc[f], 0, 0, 0);
    if (full) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float t = acc[f][q] + bv[f];
          if (HAS_RES) t += rv[f][q];
          t = (t < 0.f ? (is_relu ? 0.f : t * neg_slope) : t) * keep[f];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), out_rs, vo_out[f],
                                                (int)((p0 + (unsigned)((q & 3) + 8 * (q >> 2))) * ldob), AUX);
        }
    } else {                             // the last, partial block: a branch per pixel
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const unsigned pq = p0 + (unsigned)((q & 3) + 8 * (q >> 2));
          if (pq + 4u * (unsigned)h < npix) {
            float t = acc[f][q] + bv[f];
            if (HAS_RES) t += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(res_rs, vo_res[f], (int)(pq * ldrb), 0));
            t = (t < 0.f ? (is_relu ? 0.f : t * neg_slope) : t) * keep[f];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), out_rs, vo_out[f], (int)(pq * ldob), AUX);
          }
        }
    }
  }
}

}  // namespace

// 1 = this descriptor is a plain 1x1, 64-input-channel conv over a whole contiguous pixel range that the streaming kernel takes
int ssg_conv1x1_k64_ok(const ssg_conv_desc* d) {
  static const int on = [] { const char* e = getenv("SSG_CONV1X1_STREAM"); return e ? atoi(e) : 1; }();     // 0: LDS-DMA kernel (A/B)
  if (!on || d->ntaps != 1 || d->dy[0] || d->dx[0] || d->C1 != 64 || d->C2 != 0 || d->kmode != 0 || d->Kp != 64 || d->bnpart) return 0;
  if (d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1 || d->out_oy || d->out_ox) return 0;
  if (d->GH != d->H || d->GW != d->W || d->OH != d->H || d->OW != d->W) return 0;
  const long long P = (long long)d->N * d->H * d->W;
  if (P < 65536 || d->Cout < 32) return 0;
  if (P * d->ld1 >= (1ll << 30) || P * d->ldo >= (1ll << 30) || (d->res && P * d->ldr >= (1ll << 30))) return 0;      // 32-bit byte offsets
  if (((uintptr_t)d->in1 & 15) || d->ld1 % 4 || ((uintptr_t)d->w & 15)) return 0;
  return 1;
}

int ssg_conv1x1_k64_launch(const ssg_conv_desc* d, hipStream_t st) {
  C11Args a;
  a.in = d->in1; a.w = d->w; a.bias = d->bias; a.res = d->res; a.out = d->out;
  a.ld = d->ld1; a.ldr = d->ldr; a.ldo = d->ldo; a.Kp = d->Kp; a.Cout = d->Cout; a.act = d->act; a.slope = d->slope;
  a.P = (long long)d->N * d->H * d->W;
  a.groups = (d->Cout + 63) / 64;
  const long long nblk = (a.P + 31) / 32;
  long long waves = 6144 / a.groups;                     // ~6 waves per SIMD-quad in total
  if (waves > nblk) waves = nblk;
  a.blocks_per_wave = (int)((nblk + waves - 1) / waves);
  waves = (nblk + a.blocks_per_wave - 1) / a.blocks_per_wave;
  a.nt_store = a.P * d->ldo * 4 >= (256ll << 20);
  const dim3 grid((unsigned)((waves * a.groups + 3) / 4)), block(256);
  if (d->res) {
    if (a.nt_store) hipLaunchKernelGGL((conv1x1_k64_kernel<true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv1x1_k64_kernel<true, false>), grid, block, 0, st, a);
  } else {
    if (a.nt_store) hipLaunchKernelGGL((conv1x1_k64_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv1x1_k64_kernel<false, false>), grid, block, 0, st, a);
  }
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
