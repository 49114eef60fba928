// Implicit-GEMM convolution, LDS-DMA pipeline, with fp32 operands split into three bf16 terms on the bf16 matrix pipe
// (arithmetic: conv_igemm_halo_x3.hip).  Same contract and tiling as conv_igemm_dma.hip (kmode 0; any tap list, input stride,
// output stride / offset: 1x1 convs, stride-2 3x3, the parity-class launches of a stride-2 input gradient); selected when the
// descriptor carries split-packed weights.
//   * A tile [128 pixels of an 8 x 16 patch][16 channels] fp32, gathered per tap by LDS-DMA through a buffer descriptor (a lane
//     whose tap falls outside the image carries an out-of-range offset and receives zeros); split in registers after the
//     fragment read.  B tile = the pre-split weights of the step, [BN rows][96 B] (mfma_split.h), 12 / 6 contiguous KiB.
//   * wave layout 4 x 1 (one activation split per pixel row group), 3 stages, two K-steps in flight across the one barrier per step.
#include "common.h"
#include "lds_dma.h"
#include "conv_args.h"
#include "mfma_split.h"
#include "conv_slow.h"

namespace {

constexpr int NSTAGE = 3;

template <int BN>
__global__ __launch_bounds__(256, 2) void conv_igemm_dma_x3_kernel(const ConvArgs a) {
  constexpr int BM = 128, TH = 8;
  constexpr int NI = BN / 32;
  constexpr int A_PC = BM / 64;                          // A pieces (16 rows each) per wave per K-step
  constexpr int BPIECES = BN * XROW / 1024;              // 12 / 6
  constexpr int B_PC = (BPIECES + 3) / 4;
  constexpr int ASTG = BM * 64;                          // bytes
  constexpr int BSTG = BN * XROW;
  constexpr int STAGE = ASTG + BSTG;

  extern __shared__ __attribute__((aligned(1024))) unsigned char ldsb[];     // NSTAGE * STAGE bytes + 1 KiB dummy target
  unsigned char* const ldsDummy = ldsb + NSTAGE * STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int bid = blockIdx.x;
  if (a.xcd_swizzle) {
    const int per = (int)gridDim.x >> 3;
    if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
  }
  const int nyt = a.ntiles_n;
  const int nt = bid % nyt; bid /= nyt;
  const int n0 = nt * BN;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int n = bid / a.tiles_y;

  // ---- per-lane DMA source state: piece j of this wave covers tile rows [(wave*A_PC + j)*16, +16)
  const int lr = lane >> 2, lp = lane & 3;
  const unsigned OOB = 0xffffffffu;
  int a_iy0[A_PC], a_ix0[A_PC];
  unsigned a_q[A_PC];
#pragma unroll
  for (int j = 0; j < A_PC; ++j) {
    const int r = (wave * A_PC + j) * 16 + lr;
    const int gy = ty * TH + (r >> 4), gx = tx * 16 + (r & 15);
    const bool ok = (gy < a.GH) && (gx < a.GW);
    a_iy0[j] = ok ? gy * a.in_sy : -100000;
    a_ix0[j] = gx * a.in_sx;
    a_q[j] = 16u * (unsigned)(lp ^ ((r >> 2) & 3));
  }
  const int nsteps = a.nsteps;
  const unsigned npix = (unsigned)a.N * (unsigned)a.H * (unsigned)a.W;
  const auto in1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1), 0, (int)(npix * (unsigned)a.ld1 * 4u), 0x00020000);
  const auto in2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in2), 0, (int)(npix * (unsigned)a.ld2 * 4u), 0x00020000);
  const auto w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)((unsigned)nyt * (unsigned)nsteps * (unsigned)BSTG), 0x00020000);
  const unsigned w_lane = (unsigned)lane * 16u;
  const unsigned w_tile = (unsigned)nt * (unsigned)nsteps * (unsigned)BSTG;

  auto issue = [&](int s) {
    unsigned char* st = ldsb + (s % NSTAGE) * STAGE;
    const int chunk = s / a.ntaps;
    const int t = s - chunk * a.ntaps;
    const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
    const int dy = (tb & 7) - 2, dx = (tb >> 3) - 2;
    const int c0 = chunk * 16;
    const bool first = c0 < a.C1;
    const unsigned ldb = (unsigned)(first ? a.ld1 : a.ld2) * 4u;
    const int so = (first ? c0 : c0 - a.C1) * 4;
#pragma unroll
    for (int j = 0; j < A_PC; ++j) {
      const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const unsigned vo = ok ? (unsigned)((n * a.H + iy) * a.W + ix) * ldb + a_q[j] : OOB;
      ssg_lds_void* dst = (ssg_lds_void*)(st + (wave * A_PC + j) * 1024);
      if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(in1_rs, dst, 16, vo, so, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(in2_rs, dst, 16, vo, so, 0, 0);
    }
    const unsigned wso = w_tile + (unsigned)s * (unsigned)BSTG;
#pragma unroll
    for (int j = 0; j < B_PC; ++j) {
      const int g = wave + 4 * j;
      if (g < BPIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (ssg_lds_void*)(st + ASTG + g * 1024), 16, w_lane, wso + g * 1024, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (ssg_lds_void*)ldsDummy, 16, OOB, 0, 0, 0);
    }
  };

  f32x16 acc[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int half = lane >> 5, l31 = lane & 31;
  const int arow = wave * 32 + l31;                      // tile pixel of this lane's fragment row
  const int asw = (arow >> 2) & 3;
  const int aoff0 = arow * 64 + 16 * ((2 * half) ^ asw), aoff1 = arow * 64 + 16 * ((2 * half + 1) ^ asw);
  int boff[NI], bf[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = j * 32 + l31;
    boff[j] = ASTG + row * XROW; bf[j] = (row >> 3) & 1;
  }

  if (0 < nsteps) issue(0);
  if (1 < nsteps) issue(1);
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_PC + B_PC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 2 < nsteps) issue(s + 2);
    const unsigned char* st = ldsb + (s % NSTAGE) * STAGE;
    bf16x8 a1, a2, a3, b1[NI], b2[NI], b3[NI];
    const f32x4 u = *(const f32x4*)(st + aoff0), v = *(const f32x4*)(st + aoff1);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const unsigned char* row = st + boff[j];
      b1[j] = *(const bf16x8*)(row + 16 * ((0 + half) ^ bf[j]));
      b2[j] = *(const bf16x8*)(row + 16 * ((2 + half) ^ bf[j]));
      b3[j] = *(const bf16x8*)(row + 16 * ((4 + half) ^ bf[j]));
    }
    split3(u, v, a1, a2, a3);
#define SSG_X3_TERM(A, B)                                                                           \
  _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                   \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[j], acc[j], 0, 0, 0);
#ifndef SSG_X3_DYNPRIO
#define SSG_X3_DYNPRIO 1                                   // 1: wave priority falls through the step's six terms (as the k32 kernels).  Same-box A/B: +3-4 % here (stride-2 / 1x1 forward); the same in wgrad_dma_x3 lost 2-3 % and in the merged-parity halo kernel changed nothing: not there
#endif
    if (SSG_X3_DYNPRIO) __builtin_amdgcn_s_setprio(3);
    SSG_X3_TERM(a3, b1) SSG_X3_TERM(a2, b2)
    if (SSG_X3_DYNPRIO) __builtin_amdgcn_s_setprio(2);
    SSG_X3_TERM(a1, b3)
    if (SSG_X3_DYNPRIO) __builtin_amdgcn_s_setprio(1);
    SSG_X3_TERM(a2, b1) SSG_X3_TERM(a1, b2)
    if (SSG_X3_DYNPRIO) __builtin_amdgcn_s_setprio(0);
    SSG_X3_TERM(a1, b1)
#undef SSG_X3_TERM
  }
  wait_vmcnt<0>();
  wait_lds_reads();
  {                                                      // non-finite operands: conv_slow.h
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NI; ++j) bad |= ssg_nonfinite16(acc[j]);
    if (__builtin_amdgcn_readfirstlane(__syncthreads_or(bad))) {     // scalar condition: a uniform branch, the accumulators are dead inside it
      const ConvArgs& as = *ssg_reload_args<ConvArgs>();
#pragma unroll
      for (int j = 0; j < NI; ++j)
        ssg_slow_refill16(acc[j], (float*)ldsb + tid, 256, [&](int r) {
          const int p = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          return ssg_conv_slow_value(as, n, ty * TH + (p >> 4), tx * 16 + (p & 15), n0 + j * 32 + l31, 0, as.ntaps);
        });
    }
  }

  // ---- epilogue of conv_igemm_dma.hip for a 4 x 1 wave layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool want_bn = a.bnpart != nullptr;
  if (want_bn) ssg_bnpart_begin();
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = n0 + j * 32 + l31;
    const bool cok = co < a.Cout;
    const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
    float s1 = 0.f, s2 = 0.f; int nv = 0;
    const float piv = acc[j][0] + bv;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int gy = ty * TH + (p >> 4), gx = tx * 16 + (p & 15);
      if (gy < a.GH && gx < a.GW) {
        const size_t pix = ((size_t)(n * a.OH + gy * a.out_sy + a.out_oy) * a.OW + gx * a.out_sx + a.out_ox);
        float v = acc[j][r] + bv;
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
    if (want_bn) ssg_bnpart_put<BN, BN>((float*)ldsb, j, s1, s2, piv, nv, wave, 0, half, l31);
  }
  if (want_bn) ssg_bnpart_finish<NI, 4, BN, BN>(a, (float*)ldsb, (n * a.tiles_y + ty) * a.tiles_x + tx, n0, wave, 0, half, l31);
}

template <int BN>
int launch(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  a.tiles_x = (a.GW + 15) / 16;
  a.tiles_y = (a.GH + 7) / 8;
  static const int swz = [] { const char* e = getenv("SSG_XCD_SWIZZLE"); return e ? atoi(e) : 1; }();
  a.xcd_swizzle = swz;
  a.ntiles_n = (a.Cout + BN - 1) / BN;
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N * a.ntiles_n));
  constexpr int lds_bytes = NSTAGE * (128 * 64 + BN * XROW) + 1024;
  static_assert(lds_bytes <= 80 * 1024, "two workgroups per CU");
  static const hipError_t attr = hipFuncSetAttribute((const void*)conv_igemm_dma_x3_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (attr != hipSuccess) { ssg_set_error("conv dma x3: LDS attribute: %s", hipGetErrorString(attr)); return (int)attr; }
  hipLaunchKernelGGL((conv_igemm_dma_x3_kernel<BN>), grid, dim3(256), lds_bytes, st, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

}  // namespace

// Column tile of the split-operand LDS-DMA kernel for a launch (0 = none): whole 128- or 64-column tiles, tensors below 4 GB
// (32-bit byte offsets of the buffer descriptors), at least two K-steps
int ssg_conv_dma_x3_bn(const ConvArgs& a) {
  if (a.kmode != 0 || a.nsteps < 2 || (long long)a.N * a.GH * a.GW < 4096) return 0;      // tiny grids (the fc layers as 1x1) gain nothing from it
  const unsigned long long bytes = (unsigned long long)a.N * a.H * a.W * (unsigned long long)(a.ld1 > a.ld2 ? a.ld1 : a.ld2) * 4ull;
  if (bytes > 0xfffffff0ull) return 0;
  if (a.Cout % 128 == 0) return 128;
  if (a.Cout % 64 == 0) return 64;
  return 0;
}

int ssg_conv_igemm_dma_x3_launch(const ConvArgs& a, hipStream_t st) {
  return ssg_conv_dma_x3_bn(a) == 128 ? launch<128>(a, st) : launch<64>(a, st);
}
