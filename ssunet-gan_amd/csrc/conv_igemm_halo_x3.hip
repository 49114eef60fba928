// 3x3 stride-1 convolution (forward and input gradient) with fp32 operands SPLIT into three bf16 terms and multiplied on the
// bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate), fp32 accumulation.  Same contract, LDS halo image and
// epilogue as conv_igemm_halo.hip; selected when the descriptor carries split-packed weights (ssg_conv_desc.w_split).
//
// Arithmetic.  An fp32 value is x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): three 8-bit
// significands cover the 24 bits of x (the subtractions are exact).  A product of two bf16 values is exact in fp32, so
//   x*y = x1*y1 + (x1*y2 + x2*y1) + (x1*y3 + x2*y2 + x3*y1) + [x2*y3 + x3*y2 + x3*y3]
// and the bracket, the only part left out, is below 2^-23 |x*y| with the sign of rounding residuals: one fp32 ulp per
// product, the class of error an fp32 FMA chain has per step.  Six bf16 MFMAs replace one fp32 MFMA step of the same K at
// 16/6 = 2.7x its rate.  (This is what cuBLAS calls BF16x9 emulation of FP32, with the three terms of order 2^-24 dropped.)
//   * weights are split ONCE per pack (ssg_pack_weights_split_bf16x3): per Cout tile and K-step a dense [BN rows][96 B]
//     block = 3 planes x 16 channels x bf16, so a step's weights are 12 (BN = 128) contiguous 1-KiB DMA pieces; the two
//     k-halves of a plane swap places in every other group of 8 rows (bank-conflict-free fragment reads, see mfma_split.h).
//   * activations stay fp32 in HBM and arrive in the LDS halo by the same DMA as in conv_igemm_halo.hip.  Default form
//     (conv_igemm_halo_x3_kernel): each 16-channel halo image is split ONCE into a second LDS image of [pixel][96 B] rows and
//     the nine taps read ready fragments.  Older form (conv_igemm_halo_x3r_kernel, SSG_X3_PRESPLIT=0): fragments are split in
//     registers as they leave LDS, ~45 VALU per fragment and step.  Same results bit for bit (the split is exact either way).
//   * wave layout 4 x 1 (a wave = one 32-pixel tile row x all BN columns): 3 + 12 ds_read_b128 and 24 MFMAs per step and wave
//     (BN = 128).  LDS: 13 KB fp32 image + 20 KB split image + 3 x 12 KB weight stages = 70 KB -> 2 workgroups per CU; BN = 64:
//     52 KB -> 3.  SSG_X3_LAYOUT=22 selects 2 x 2 waves (6 + 6 reads): same speed.
//   * the loop is power-limited, not issue-limited: stamped in-kernel clock 1.82 GHz (the fp32-MFMA halo kernel: 2.37), and a
//     register-only v_mfma_f32_32x32x16_bf16 loop on random operands sustains 1.80 PFLOP/s on the same device (2.45 on
//     constants) -- tools/clock_probe.py x3, tools/micro_peak_bf16.py, DESIGN.md 3.9.
#include "common.h"
#include "lds_dma.h"
#include "conv_args.h"
#include "conv_halo_epilogue.h"
#include "mfma_split.h"
#include "conv_slow.h"

namespace {

#ifdef SSG_CLOCK_PROBE
// diagnostic build only (tools/clock_probe.py x3): per workgroup (shader cycles, 100-MHz ticks) spent in the main loop
__device__ unsigned long long* ssg_probe_buf_x3 = nullptr;
#endif

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, 2) void conv_igemm_halo_x3r_kernel(const ConvArgs a) {
  constexpr int TWL = 5, TW = 32, TH = BM / TW;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int HW = TW + 2, HR = (TH + 2) * HW;
  constexpr int AP = (HR + 15) / 16;                     // 1-KiB pieces per halo tile
  constexpr int APW = (AP + 3) / 4;                      // pieces per wave (dummy-padded), one per step
  constexpr int BPIECES = BN * XROW / 1024;              // 12 (BN = 128) / 6 (BN = 64) 1-KiB pieces per step
  constexpr int B_PC = (BPIECES + 3) / 4;                // per wave and step; indices >= BPIECES are dummies (BN = 64: waves 2, 3)
  constexpr int ABUF = AP * 256;                         // floats per halo buffer
  constexpr int BSTG = BN * XROW;                        // bytes per weight stage
  static_assert(APW <= 7, "A pieces must be issued before the last two steps of a chunk");

  extern __shared__ __attribute__((aligned(1024))) float lds[];     // 2 * ABUF floats + 3 * BSTG bytes + 1 KiB dummy target
  unsigned char* const ldsB = (unsigned char*)(lds + 2 * ABUF);
  unsigned char* const ldsDummy = ldsB + 3 * BSTG;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

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

  // ---- DMA sources through buffer descriptors (`buffer_load_dwordx4 ... offen lds`): the per-piece part of an address is a
  // SCALAR offset and a lane whose halo pixel lies outside the image carries a byte offset beyond the range, which the
  // hardware answers with zeros -- no pointer arithmetic, select or zero page on the vector unit (this kernel is bound by
  // vector-instruction issue: 24 MFMAs leave room for ~5 VALU instructions each, the operand split takes most of it)
  const int lr = lane >> 2, lp = lane & 3;
  const unsigned a_q = 16u * (unsigned)(lp ^ ((lr >> 2) & 3));       // byte offset of this lane's channel quad (rows of a piece are 16-aligned)
  const unsigned OOB = 0xffffffffu;                      // beyond any range (ssg_conv_halo_x3_ok admits tensors below 4 GB only)
  unsigned a_pix[APW];                                   // pixel index of this lane's halo row of piece k, or OOB
#pragma unroll
  for (int k = 0; k < APW; ++k) {
    const int g = wave + 4 * k;
    const int r = g * 16 + lr;
    const int hy = r / HW, hx = r - hy * HW;
    const int iy = ty * TH + hy - 1, ix = tx * TW + hx - 1;
    const bool ok = g < AP && r < HR && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    a_pix[k] = ok ? (unsigned)((n * a.H + iy) * a.W + ix) : OOB;
  }
  const int nchunks = (a.C1 + a.C2) >> 4;
  const int nsteps = nchunks * 9;
  const unsigned npix = (unsigned)a.N * (unsigned)a.H * (unsigned)a.W;
  const auto in1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1), 0, (int)(npix * (unsigned)a.ld1 * 4u), 0x00020000);
  const auto in2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in2), 0, (int)(npix * (unsigned)a.ld2 * 4u), 0x00020000);
  // weights: [Cout tile][step][BN rows][96 B]: a step is BPIECES consecutive lane-linear KiB, dealt to the waves round-robin
  const auto w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)((unsigned)nyt * (unsigned)nsteps * (unsigned)BSTG), 0x00020000);
  const unsigned w_lane = (unsigned)lane * 16u;
  const unsigned w_tile = (unsigned)nt * (unsigned)nsteps * (unsigned)BSTG;

  auto issue_a = [&](int chunk, int k) {
    const int g = wave + 4 * k;
    if (g >= AP) { return; }
    ssg_lds_void* dst = (ssg_lds_void*)(lds + (chunk & 1) * ABUF + g * 256);
    const int c0 = chunk * 16;
    const bool live = chunk < nchunks;                   // past the last chunk: a dummy piece (all lanes out of range) keeps vmcnt uniform
    if (c0 < a.C1) {
      const unsigned vo = (live && a_pix[k] != OOB) ? a_pix[k] * (unsigned)a.ld1 * 4u + a_q : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in1_rs, dst, 16, vo, c0 * 4, 0, 0);
    } else {
      const unsigned vo = (live && a_pix[k] != OOB) ? a_pix[k] * (unsigned)a.ld2 * 4u + a_q : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in2_rs, dst, 16, vo, (c0 - a.C1) * 4, 0, 0);
    }
  };
  auto issue_b = [&](int s) {
    unsigned char* st = ldsB + (s % 3) * BSTG;
    // past the last step the pieces are dummies that keep vmcnt uniform: they re-read step 0 (the scalar offset is not range checked)
    const unsigned so = w_tile + (s < nsteps ? (unsigned)s * (unsigned)BSTG : 0u);
#pragma unroll
    for (int j = 0; j < B_PC; ++j) {
      const int g = wave + 4 * j;                        // piece index: waves interleave
      if (g < BPIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (ssg_lds_void*)(st + g * 1024), 16, w_lane, so + g * 1024, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (ssg_lds_void*)ldsDummy, 16, OOB, 0, 0, 0);   // every lane out of range
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
  int rb[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int p = wm * WTM + i * 32 + l31;
    rb[i] = ((p >> TWL) + 1) * HW + (p & (TW - 1)) + 1;
  }
  // weight fragment j: row (wn*WTN + j*32 + l31) of the stage, slot (2*plane + half) at position slot ^ ((row >> 3) & 1)
  int boff[NI], bf[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = wn * WTN + j * 32 + l31;
    boff[j] = row * XROW; bf[j] = (row >> 3) & 1;
  }

  // ---- prologue
#pragma unroll
  for (int k = 0; k < APW; ++k) issue_a(0, k);
  issue_b(0);
  issue_b(1);
  const bool has_last = wave + 4 * (APW - 1) < AP;

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const float* Abuf = lds + (chunk & 1) * ABUF;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int s = chunk * 9 + t;
      const int tp = (t + 8) % 9;
      if (tp < APW - 1) wait_vmcnt<B_PC + 1>();
      else if (tp == APW - 1) { if (has_last) wait_vmcnt<B_PC + 1>(); else wait_vmcnt<B_PC>(); }
      else wait_vmcnt<B_PC>();
      wait_lds_reads();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (t < APW) issue_a(chunk + 1, t);
      issue_b(s + 2);

      const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
      const int toff = ((tb & 7) - 2) * HW + ((tb >> 3) - 2);
      const unsigned char* Bst = ldsB + (t % 3) * BSTG;                  // s % 3 == t % 3 (9 steps per chunk)
      bf16x8 a1[MI], a2[MI], a3[MI], b1[NI], b2[NI], b3[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = rb[i] + toff;
        const int sw = (r >> 2) & 3;
        const f32x4 u = *(const f32x4*)(Abuf + r * 16 + 4 * ((2 * half) ^ sw));
        const f32x4 v = *(const f32x4*)(Abuf + r * 16 + 4 * ((2 * half + 1) ^ sw));
        split3(u, v, a1[i], a2[i], a3[i]);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const unsigned char* row = Bst + boff[j];
        b1[j] = *(const bf16x8*)(row + 16 * ((0 + half) ^ bf[j]));
        b2[j] = *(const bf16x8*)(row + 16 * ((2 + half) ^ bf[j]));
        b3[j] = *(const bf16x8*)(row + 16 * ((4 + half) ^ bf[j]));
      }
      // small terms first; consecutive MFMAs go to different accumulators
#define SSG_X3_TERM(A, B)                                                                           \
  _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                   \
  _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                   \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B[j], acc[i][j], 0, 0, 0);
      SSG_X3_TERM(a3, b1) SSG_X3_TERM(a2, b2) SSG_X3_TERM(a1, b3)
      SSG_X3_TERM(a2, b1) SSG_X3_TERM(a1, b2)
      SSG_X3_TERM(a1, b1)
#undef SSG_X3_TERM
    }
  }
  wait_vmcnt<0>();
  wait_lds_reads();
  {                                                      // non-finite operands: conv_slow.h
    bool bad = false;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) bad |= ssg_nonfinite16(acc[i][j]);
    if (__builtin_amdgcn_readfirstlane(__syncthreads_or(bad))) {     // scalar condition: a uniform branch, the accumulators are dead inside it
      const ConvArgs& as = *ssg_reload_args<ConvArgs>();
      float* scr = (float*)lds + tid;
#pragma unroll 1
      for (int e = 0; e < MI * NI * 16; ++e) {
        const int i = e / (NI * 16), j = (e >> 4) % NI, r = e & 15;
        const int p = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        scr[e * 256] = ssg_conv_slow_value(as, n, ty * TH + (p >> TWL), tx * TW + (p & (TW - 1)), n0 + wn * WTN + j * 32 + l31, 0, 9);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = scr[((i * NI + j) * 16 + r) * 256];
    }
  }
  ssg_halo_epilogue<BM, BN, WAVES_M, WAVES_N, TWL, false>(a, acc, lds, n, ty, tx, n0, 0, wm, wn, half, l31);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Pre-split form (4 x 1 wave layout): the 16-channel halo image of a chunk is split ONCE, as it lands, into a second LDS image
// of [pixel][96 B] rows (the weight-row format), and the nine tap steps read ready bf16 fragments -- 3 + 3 * NI ds_read_b128 and
// no vector arithmetic per step, where the form above splits the same pixel again for each of the 9 taps that touch it (~45
// VALU per step; SQ counters of round 3: 2.7 VALU per MFMA, MFMA busy 61 %).  The fp32 image is single-buffered (the DMA of
// chunk c + 1 is issued after chunk c has been converted); a chunk boundary costs one conversion pass (2 x 13 KB read, 19 KB
// written by 256 threads) and one extra barrier per 9 steps.  LDS: 13 KB fp32 image + 20 KB split image + 3 weight stages + 1 KB.
// PARITY: the four parity classes of a 3x3 stride-2 input gradient in ONE launch.  Output pixel
// (2gy + py, 2gx + px) of class (py, px) sums 1 / 2 / 2 / 4 taps of the SAME 2x2 neighbourhood of dy around (gy, gx), so the four
// launches of the LDS-DMA kernel (which gathered dy once per tap and class, K = 1..4 taps deep: prologue-bound) become the nine tap
// steps of one halo tile accumulating into four accumulator sets -- tap t belongs to class 0 | 1 1 | 2 2 | 3 3 3 3, the order
// ops._conv_dgrad_impl packs them in -- and four strided epilogues.  4 x 32 registers of accumulators: BN = 64 only.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool PARITY = false>
__global__ __launch_bounds__(256, (BN == 64 && !PARITY) ? 3 : 2) void conv_igemm_halo_x3_kernel(const ConvArgs a) {
  static_assert(BM == 128, "4 rows of 32 pixels");
  static_assert(!PARITY || (BN == 64 && WAVES_M == 4), "the merged parity form is the 4 x 1 layout on 64 columns");
  constexpr int NCLS = PARITY ? 4 : 1;
  constexpr int TWL = 5, TW = 32, TH = 4;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int HW = TW + 2, HR = (TH + 2) * HW;
  constexpr int AP = (HR + 15) / 16;
  constexpr int APW = (AP + 3) / 4;
  constexpr int BPIECES = BN * XROW / 1024;
  constexpr int B_PC = (BPIECES + 3) / 4;
  constexpr int ABUF = AP * 256;                         // floats of the fp32 halo image
  constexpr int SBUF = ((AP * 16 * XROW + 1023) / 1024) * 1024;      // bytes of the split image (whole KiB: the DMA targets behind it stay aligned)
  constexpr int BSTG = BN * XROW;
  static_assert(APW <= 7, "A pieces must be issued before the last two steps of a chunk");

  extern __shared__ __attribute__((aligned(1024))) float lds[];
  unsigned char* const ldsS = (unsigned char*)(lds + ABUF);
  unsigned char* const ldsB = ldsS + SBUF;
  unsigned char* const ldsDummy = ldsB + 3 * BSTG;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

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

  const int lr = lane >> 2, lp = lane & 3;
  const unsigned a_q = 16u * (unsigned)(lp ^ ((lr >> 2) & 3));
  const unsigned OOB = 0xffffffffu;
  unsigned a_pix[APW];
#pragma unroll
  for (int k = 0; k < APW; ++k) {
    const int g = wave + 4 * k;
    const int r = g * 16 + lr;
    const int hy = r / HW, hx = r - hy * HW;
    const int iy = ty * TH + hy - 1, ix = tx * TW + hx - 1;
    const bool ok = g < AP && r < HR && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    a_pix[k] = ok ? (unsigned)((n * a.H + iy) * a.W + ix) : OOB;
  }
  const int nchunks = (a.C1 + a.C2) >> 4;
  const int nsteps = nchunks * 9;
  const unsigned npix = (unsigned)a.N * (unsigned)a.H * (unsigned)a.W;
  const auto in1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1), 0, (int)(npix * (unsigned)a.ld1 * 4u), 0x00020000);
  const auto in2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in2), 0, (int)(npix * (unsigned)a.ld2 * 4u), 0x00020000);
  const auto w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)((unsigned)nyt * (unsigned)nsteps * (unsigned)BSTG), 0x00020000);
  const unsigned w_lane = (unsigned)lane * 16u;
  const unsigned w_tile = (unsigned)nt * (unsigned)nsteps * (unsigned)BSTG;

  auto issue_a = [&](int chunk, int k) {
    const int g = wave + 4 * k;
    if (g >= AP) { return; }
    ssg_lds_void* dst = (ssg_lds_void*)(lds + g * 256);
    const int c0 = chunk * 16;
    const bool live = chunk < nchunks;
    if (c0 < a.C1) {
      const unsigned vo = (live && a_pix[k] != OOB) ? a_pix[k] * (unsigned)a.ld1 * 4u + a_q : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in1_rs, dst, 16, vo, c0 * 4, 0, 0);
    } else {
      const unsigned vo = (live && a_pix[k] != OOB) ? a_pix[k] * (unsigned)a.ld2 * 4u + a_q : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(in2_rs, dst, 16, vo, (c0 - a.C1) * 4, 0, 0);
    }
  };
  auto issue_b = [&](int s) {
    unsigned char* st = ldsB + (s % 3) * BSTG;
    const unsigned so = w_tile + (s < nsteps ? (unsigned)s * (unsigned)BSTG : 0u);
#pragma unroll
    for (int j = 0; j < B_PC; ++j) {
      const int g = wave + 4 * j;
      if (g < BPIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (ssg_lds_void*)(st + g * 1024), 16, w_lane, so + g * 1024, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (ssg_lds_void*)ldsDummy, 16, OOB, 0, 0, 0);
    }
  };

  f32x16 acc[NCLS][MI][NI];
#pragma unroll
  for (int q = 0; q < NCLS; ++q)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;

  const int half = lane >> 5, l31 = lane & 31;
  int rb[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int p = wm * WTM + i * 32 + l31;
    rb[i] = ((p >> TWL) + 1) * HW + (p & (TW - 1)) + 1;
  }
  int boff[NI], bf[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = wn * WTN + j * 32 + l31;
    boff[j] = row * XROW; bf[j] = (row >> 3) & 1;
  }
  // conversion items of this thread: (pixel, k-half) pairs tid and tid + 256 of the HR * 2
  constexpr int CV = (HR * 2 + 255) / 256;

#pragma unroll
  for (int k = 0; k < APW; ++k) issue_a(0, k);
  issue_b(0);
  issue_b(1);
  const bool has_last = wave + 4 * (APW - 1) < AP;
#ifdef SSG_CLOCK_PROBE
  const unsigned long long pt0 = __builtin_amdgcn_s_memtime(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif

  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int s = chunk * 9 + t;
      const int tp = (t + 8) % 9;
      if (tp < APW - 1) wait_vmcnt<B_PC + 1>();
      else if (tp == APW - 1) { if (has_last) wait_vmcnt<B_PC + 1>(); else wait_vmcnt<B_PC>(); }
      else wait_vmcnt<B_PC>();
      wait_lds_reads();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (t == 0) {
        // the fp32 image of this chunk has landed and nobody reads the split image of the last one any more: convert
#pragma unroll
        for (int c = 0; c < CV; ++c) {
          const int i = tid + c * 256;
          if (i < HR * 2) {
            const int px = i >> 1, h = i & 1;
            const int sw = (px >> 2) & 3;
            const f32x4 u = *(const f32x4*)(lds + px * 16 + 4 * ((2 * h) ^ sw));
            const f32x4 v = *(const f32x4*)(lds + px * 16 + 4 * ((2 * h + 1) ^ sw));
            bf16x8 p1, p2, p3;
            split3(u, v, p1, p2, p3);
            const int f = (px >> 3) & 1;
            unsigned char* dst = ldsS + px * XROW;
            *(bf16x8*)(dst + 16 * ((0 + h) ^ f)) = p1;
            *(bf16x8*)(dst + 16 * ((2 + h) ^ f)) = p2;
            *(bf16x8*)(dst + 16 * ((4 + h) ^ f)) = p3;
          }
        }
        wait_lds_reads();                                // lgkmcnt(0): the writes too
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (t < APW) issue_a(chunk + 1, t);
      issue_b(s + 2);

      const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
      const int toff = ((tb & 7) - 2) * HW + ((tb >> 3) - 2);
      const unsigned char* Bst = ldsB + (t % 3) * BSTG;
      bf16x8 a1[MI], a2[MI], a3[MI], b1[NI], b2[NI], b3[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = rb[i] + toff;
        const int f = (r >> 3) & 1;
        const unsigned char* row = ldsS + r * XROW;
        a1[i] = *(const bf16x8*)(row + 16 * ((0 + half) ^ f));
        a2[i] = *(const bf16x8*)(row + 16 * ((2 + half) ^ f));
        a3[i] = *(const bf16x8*)(row + 16 * ((4 + half) ^ f));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const unsigned char* row = Bst + boff[j];
        b1[j] = *(const bf16x8*)(row + 16 * ((0 + half) ^ bf[j]));
        b2[j] = *(const bf16x8*)(row + 16 * ((2 + half) ^ bf[j]));
        b3[j] = *(const bf16x8*)(row + 16 * ((4 + half) ^ bf[j]));
      }
      const int cls = PARITY ? (t < 1 ? 0 : (t < 3 ? 1 : (t < 5 ? 2 : 3))) : 0;        // constant after unrolling
#define SSG_X3_TERM(A, B)                                                                           \
  _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                   \
  _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                   \
      acc[cls][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B[j], acc[cls][i][j], 0, 0, 0);
      SSG_X3_TERM(a3, b1) SSG_X3_TERM(a2, b2) SSG_X3_TERM(a1, b3)
      SSG_X3_TERM(a2, b1) SSG_X3_TERM(a1, b2)
      SSG_X3_TERM(a1, b1)
#undef SSG_X3_TERM
    }
  }
  wait_vmcnt<0>();
  wait_lds_reads();
#ifdef SSG_CLOCK_PROBE
  if (ssg_probe_buf_x3 && tid == 0) {
    ssg_probe_buf_x3[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - pt0;
    ssg_probe_buf_x3[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - pr0;
  }
#endif
  {                                                      // non-finite operands: conv_slow.h
    bool bad = false;
#pragma unroll
    for (int q = 0; q < NCLS; ++q)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) bad |= ssg_nonfinite16(acc[q][i][j]);
    if (__builtin_amdgcn_readfirstlane(__syncthreads_or(bad))) {
      const ConvArgs& as = *ssg_reload_args<ConvArgs>();
      if constexpr (PARITY) {
        // 128 accumulators per lane: the slow path writes the four classes itself (the merged launch has no bias, residual or
        // statistics: ssg_conv_halo_x3_parity_ok) instead of refilling registers, which would spill
#pragma unroll 1
        for (int e = 0; e < 4 * MI * NI * 16; ++e) {
          const int q = e / (MI * NI * 16), i = (e / (NI * 16)) % MI, j = (e >> 4) % NI, r = e & 15;
          const int t_lo = q == 0 ? 0 : (q == 1 ? 1 : (q == 2 ? 3 : 5)), t_hi = q == 0 ? 1 : (q == 1 ? 3 : (q == 2 ? 5 : 9));   // taps of class q: 0 | 1 2 | 3 4 | 5..8
          const int p = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const int gy = ty * TH + (p >> TWL), gx = tx * TW + (p & (TW - 1)), co = n0 + wn * WTN + j * 32 + l31;
          const int GHq = (as.OH - (q >> 1) + 1) >> 1, GWq = (as.OW - (q & 1) + 1) >> 1;
          if (gy < GHq && gx < GWq && co < as.Cout) {
            const float v = ssg_conv_slow_value(as, n, gy, gx, co, t_lo, t_hi);
            as.out[((size_t)(n * as.OH + 2 * gy + (q >> 1)) * as.OW + 2 * gx + (q & 1)) * as.ldo + co] = ssg_act(v, as.act, as.slope);
          }
        }
        return;
      } else {
        float* scr = (float*)lds + tid;                   // element e at scr[e * 256] (<= 64 elements = 64 KB)
#pragma unroll 1
        for (int e = 0; e < MI * NI * 16; ++e) {
          const int i = e / (NI * 16), j = (e >> 4) % NI, r = e & 15;
          const int p = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          scr[e * 256] = ssg_conv_slow_value(as, n, ty * TH + (p >> TWL), tx * TW + (p & (TW - 1)), n0 + wn * WTN + j * 32 + l31, 0, 9);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][i][j][r] = scr[((i * NI + j) * 16 + r) * 256];
      }
    }
  }
  if constexpr (PARITY) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ConvArgs b = a;                                    // class (py, px) = (q >> 1, q & 1): its own grid extent and output phase
      b.out_oy = q >> 1; b.out_ox = q & 1;
      b.GH = (a.OH - (q >> 1) + 1) >> 1; b.GW = (a.OW - (q & 1) + 1) >> 1;
      ssg_halo_epilogue<BM, BN, WAVES_M, WAVES_N, TWL, false>(b, acc[q], lds, n, ty, tx, n0, 0, wm, wn, half, l31);
    }
  } else {
    ssg_halo_epilogue<BM, BN, WAVES_M, WAVES_N, TWL, false>(a, acc[0], lds, n, ty, tx, n0, 0, wm, wn, half, l31);
  }
}

// fp32 packed [R][Kp] (kmode 0: k = step*16 + c) -> split tiles [ceil(R/BN)][nsteps][BN][96 B]; one thread per (row, step, k-half)
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ w, int R, int Kp, int BN, unsigned char* __restrict__ out) {
  const int nsteps = Kp >> 4;
  const long long total = (long long)((R + BN - 1) / BN) * BN * nsteps * 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int h = (int)(i & 1);
    long long t = i >> 1;
    const int s = (int)(t % nsteps); t /= nsteps;
    const int row = (int)t;                              // global row (Cout index), padded to the tile
    const int tile = row / BN, rl = row - tile * BN;
    bf16x8 p1, p2, p3;
    if (row < R) {
      const float* src = w + (size_t)row * Kp + s * 16 + 8 * h;
      split3(*(const f32x4*)src, *(const f32x4*)(src + 4), p1, p2, p3);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { p1[e] = (__bf16)0.f; p2[e] = (__bf16)0.f; p3[e] = (__bf16)0.f; }
    }
    unsigned char* dst = out + ((size_t)tile * nsteps + s) * BN * XROW + (size_t)rl * XROW;
    const int f = (rl >> 3) & 1;
    *(bf16x8*)(dst + 16 * ((0 + h) ^ f)) = p1;
    *(bf16x8*)(dst + 16 * ((2 + h) ^ f)) = p2;
    *(bf16x8*)(dst + 16 * ((4 + h) ^ f)) = p3;
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  constexpr int TW = 32, TH = BM / TW;
  constexpr int AP = ((TH + 2) * (TW + 2) + 15) / 16;
  a.tiles_x = (a.GW + TW - 1) / TW;
  a.tiles_y = (a.GH + TH - 1) / TH;
  static const int swz = [] { const char* e = getenv("SSG_XCD_SWIZZLE"); return e ? atoi(e) : 1; }();
  a.xcd_swizzle = swz;
  a.ntiles_n = (a.Cout + BN - 1) / BN;
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N * a.ntiles_n));
  constexpr int lds_bytes = 2 * AP * 1024 + 3 * BN * XROW + 1024;
  static_assert(lds_bytes <= 80 * 1024, "two workgroups per CU");
  static const hipError_t attr = hipFuncSetAttribute((const void*)conv_igemm_halo_x3r_kernel<BM, BN, WAVES_M, WAVES_N>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (attr != hipSuccess) { ssg_set_error("conv halo x3: LDS attribute: %s", hipGetErrorString(attr)); return (int)attr; }
  hipLaunchKernelGGL((conv_igemm_halo_x3r_kernel<BM, BN, WAVES_M, WAVES_N>), grid, dim3(256), lds_bytes, st, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

template <int BN, int WAVES_M, int WAVES_N, bool PARITY = false>
int launch_p(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  constexpr int TW = 32, TH = 4;
  constexpr int AP = ((TH + 2) * (TW + 2) + 15) / 16;
  a.tiles_x = (a.GW + TW - 1) / TW;
  a.tiles_y = (a.GH + TH - 1) / TH;
  static const int swz = [] { const char* e = getenv("SSG_XCD_SWIZZLE"); return e ? atoi(e) : 1; }();
  a.xcd_swizzle = swz;
  a.ntiles_n = (a.Cout + BN - 1) / BN;
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N * a.ntiles_n));
  constexpr int lds_bytes = AP * 1024 + ((AP * 16 * XROW + 1023) / 1024) * 1024 + 3 * BN * XROW + 1024;
  static_assert(lds_bytes <= 80 * 1024, "two workgroups per CU");
  static const hipError_t attr = hipFuncSetAttribute((const void*)conv_igemm_halo_x3_kernel<128, BN, WAVES_M, WAVES_N, PARITY>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (attr != hipSuccess) { ssg_set_error("conv halo x3: LDS attribute: %s", hipGetErrorString(attr)); return (int)attr; }
  hipLaunchKernelGGL((conv_igemm_halo_x3_kernel<128, BN, WAVES_M, WAVES_N, PARITY>), grid, dim3(256), lds_bytes, st, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

}  // namespace

#ifdef SSG_CLOCK_PROBE
extern "C" int ssg_debug_set_probe_buffer_x3(void* p) {
  unsigned long long* v = (unsigned long long*)p;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(ssg_probe_buf_x3), &v, sizeof(v));
}
#endif

// Column-tile width the split path uses for a launch (= the BN the weights must be split-packed for): 128 where the fp32 path
// takes <128,128>, 64 otherwise.
int ssg_conv_halo_x3_bn(const ConvArgs& a, int variant) {
  const int hv = ssg_conv_halo_variant(a, variant);
  return hv == 0 ? 128 : 64;
}

// eligible: the 9 taps of a 3x3 window at unit stride on 32-wide tiles, no split-K, Cout a multiple of the column tile
bool ssg_conv_halo_x3_ok(const ConvArgs& a, int variant) {
  if (!ssg_conv_halo_ok(a)) return false;
  const int hv = ssg_conv_halo_variant(a, variant);
  if (hv >= 3) return false;                                   // 8x16-pixel tiles (images <= 16 wide)
  const int bn = hv == 0 ? 128 : 64;
  const unsigned long long bytes = (unsigned long long)a.N * a.H * a.W * (unsigned long long)(a.ld1 > a.ld2 ? a.ld1 : a.ld2) * 4ull;
  if (bytes > 0xfffffff0ull) return false;                    // 32-bit byte offsets of the buffer descriptors
  return a.Cout % bn == 0;
}

// merged parity classes of a 3x3 stride-2 input gradient: 9 taps with dy offsets in {0,1}^2 in class order, unit input stride,
// output stride 2 from phase (0,0), the grid of class (0,0) (the largest), whole 64-column tiles, no bias / residual / statistics
bool ssg_conv_halo_x3_parity_ok(const ConvArgs& a) {
  static const int on = [] { const char* e = getenv("SSG_X3_PARITY"); return e ? atoi(e) : 1; }();
  if (!on || !a.parity || a.kmode != 0 || a.ntaps != 9 || a.in_sy != 1 || a.in_sx != 1 || a.out_sy != 2 || a.out_sx != 2 || a.out_oy || a.out_ox) return false;
  if (a.GH != (a.OH + 1) / 2 || a.GW != (a.OW + 1) / 2 || a.Cout % 64 || a.bias || a.res || a.bnpart) return false;
  static const int want[9][2] = {{0, 0}, {0, 1}, {0, 0}, {1, 0}, {0, 0}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};   // (dy, dx) per tap
  for (int t = 0; t < 9; ++t) {
    const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
    if ((tb & 7) - 2 != want[t][0] || (tb >> 3) - 2 != want[t][1]) return false;
  }
  const unsigned long long bytes = (unsigned long long)a.N * a.H * a.W * (unsigned long long)(a.ld1 > a.ld2 ? a.ld1 : a.ld2) * 4ull;
  return bytes <= 0xfffffff0ull && a.W >= 17;               // 32-bit buffer offsets; 32-wide tiles
}

int ssg_conv_igemm_halo_x3_parity_launch(const ConvArgs& a, hipStream_t st) { return launch_p<64, 4, 1, true>(a, st); }

int ssg_conv_igemm_halo_x3_launch(const ConvArgs& a, int variant, hipStream_t st) {
  // wave layout: 4 x 1 (each wave one 32-pixel tile row x all BN columns) splits each activation fragment once per
  // workgroup instead of twice (2 x 2); SSG_X3_LAYOUT=22 selects the 2 x 2 layout (A/B)
  static const int layout = [] { const char* e = getenv("SSG_X3_LAYOUT"); return e ? atoi(e) : 41; }();
  static const int presplit = [] { const char* e = getenv("SSG_X3_PRESPLIT"); return e ? atoi(e) : 1; }();
  if (presplit) {
    const bool wide = ssg_conv_halo_x3_bn(a, variant) == 128;
    if (layout == 22) return wide ? launch_p<128, 2, 2>(a, st) : launch_p<64, 2, 2>(a, st);
    return wide ? launch_p<128, 4, 1>(a, st) : launch_p<64, 4, 1>(a, st);
  }
  if (ssg_conv_halo_x3_bn(a, variant) == 128) return layout == 22 ? launch<128, 128, 2, 2>(a, st) : launch<128, 128, 4, 1>(a, st);
  return layout == 22 ? launch<128, 64, 2, 2>(a, st) : launch<128, 64, 4, 1>(a, st);
}

extern "C" int64_t ssg_pack_weights_split_bytes(int R, int Kp, int BN) {
  if (BN == 1064 || BN == 1128 || BN == 1016 || BN == 1032) {   // k32 format (conv_igemm_halo_k32.hip): whole 32-channel steps of 9 taps; whole tiles (1016 / 1032: the one tile is padded with zero rows)
    const int bn = BN - 1000;
    if (R <= 0 || (bn >= 64 && R % bn) || (bn < 64 && R > bn) || Kp <= 0 || Kp % (32 * 9)) return 0;
    return (int64_t)((R + bn - 1) / bn) * (Kp / 32) * bn * 192;
  }
  if (R <= 0 || Kp <= 0 || Kp % 16 || (BN != 64 && BN != 128)) return 0;
  return (int64_t)((R + BN - 1) / BN) * (Kp / 16) * BN * XROW;
}

extern "C" int ssg_pack_weights_split_bf16x3(const float* w_packed, int R, int Kp, int BN, void* out, void* stream) {
  if (BN == 1064 || BN == 1128 || BN == 1016 || BN == 1032) {
    SSG_REQUIRE(w_packed && out && R > 0 && (BN >= 1064 ? R % (BN - 1000) == 0 : R <= BN - 1000) && Kp > 0 && Kp % (32 * 9) == 0, SSG_EINVAL, "pack_split k32: bad args");
    SSG_REQUIRE(ssg_aligned16(w_packed) && ssg_aligned16(out), SSG_EALIGN, "pack_split: 16-B alignment");
    return ssg_pack_split_k32_launch(w_packed, R, Kp, BN - 1000, out, (hipStream_t)stream);
  }
  SSG_REQUIRE(w_packed && out && R > 0 && Kp > 0 && Kp % 16 == 0 && (BN == 64 || BN == 128), SSG_EINVAL, "pack_split: bad args");
  SSG_REQUIRE(ssg_aligned16(w_packed) && ssg_aligned16(out), SSG_EALIGN, "pack_split: 16-B alignment");
  const long long total = (long long)((R + BN - 1) / BN) * BN * (Kp >> 4) * 2;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w_packed, R, Kp, BN, (unsigned char*)out);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
