// 3x3 stride-1 convolution (forward and input gradient), split-operand form on v_mfma_f32_16x16x32_bf16 with 32-channel chunks
// (round 4).  Same contract as conv_igemm_halo_x3.hip (fp32 tensors in HBM, every operand x = x1 + x2 + x3 in bf16 terms, six
// products, fp32 accumulation: mfma_split.h), different machine mapping -- chosen from measurements of loop skeletons on random
// operands (tools/mfma_lab.hip):
//   * the bf16 pipe is power-limited, and the 16x16x32 shape delivers 1.15x the FLOP/s of 32x32x16 at the clock the chip then
//     holds (bare loops: 2.00 vs 1.73 PFLOP/s); inside a ring skeleton the same change is worth +9 %;
//   * a 512-thread workgroup on a 256-pixel x 128-channel tile (8 waves = 4 x 2, each 64 pixels x 64 channels) streams HALF the
//     weight bytes per FLOP of the 128 x 128 tile through LDS and fetches 1.33x instead of 1.59x the pixels (halo overhead);
//     skeleton 1.67 PFLOP/s executed against 1.34 for the 256-thread 128 x 64 form;
//   * one K-step = one tap x 32 channels: 96 MFMAs per wave between two workgroup barriers (24 in the x3 kernel), and a 128-byte
//     line of the NHWC input is fetched once (the 16-channel chunks of the x3 kernel fetched it twice, 9 steps apart).
// Data flow per workgroup:
//   weights  [Cout tile][step = chunk32 * 9 + tap][fragment j][plane][lane][16 B]  (ssg_pack_weights_split_bf16x3, fmt 1128 / 1064)
//            -> ring of 3 stages x BN*192 B by LDS-DMA, lane-linear 1-KiB pieces: a fragment read is base + lane * 16, conflict-free;
//            the weights are the A operand (rows = output channels), so a lane ends up with 4 consecutive output channels of one
//            pixel and stores them as one 16-byte access;
//   pixels   fp32 NHWC -> registers (buffer loads, 32 B per lane: 8 channels of one halo pixel; out-of-image lanes read zeros
//            through the descriptor's range check) -> split3 -> LDS image [plane][k-group][NPIX][16 B] (single-buffered: the
//            loads for chunk c+1 are issued at tap 4 of chunk c, split at tap 8, written between two barriers at the chunk
//            boundary).  A pixel fragment of any tap is 16 consecutive pixels of one k-group row: with NPIX % 16 == 0 the
//            ds_read_b128 lane groups hit 16 distinct 16-byte bank slots at every alignment.
// LDS: 66 KB image + 72 KB ring = 138 KB (one workgroup per CU) for <8 rows, 128 ch>; 39 + 36 = 75 KB (two per CU) for <4 rows, 64 ch>.
#include "common.h"
#include "lds_dma.h"
#include "conv_args.h"
#include "mfma_split.h"
#include "conv_slow.h"
#include <stdlib.h>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef SSG_K32_PROBE
// diagnostic build only (tools/k32_probe.py): per workgroup s_memtime at kernel start / loop start / loop end / kernel end (+ s_memrealtime,
// 100 MHz, at the first and last: the in-kernel clock)
// 32 slots per workgroup: 0-3 the four stamps, 4-5 s_memrealtime, 6-7 / 24-25 the prologue / epilogue sub-stamps, 8 + 2 * wave + {0, 1}: cycles wave `wave` spent, summed over the K-steps,
// (0) on its own vmcnt / lgkmcnt waits at the top of a step and (1) inside the step's s_barrier (waiting for the slowest wave)
constexpr int SSG_PROBE_SLOTS = 32;
__device__ unsigned long long* ssg_probe_buf_k32 = nullptr;
#define SSG_STAMP(i) do { if (ssg_probe_buf_k32 && tid == 0) { ssg_probe_buf_k32[SSG_PROBE_SLOTS * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); \
                                                            if ((i) == 0 || (i) == 3) ssg_probe_buf_k32[SSG_PROBE_SLOTS * blockIdx.x + 4 + ((i) == 3)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define SSG_PROBE_NOW(v) unsigned long long v = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) :: "memory")
#else
#define SSG_STAMP(i) do { } while (0)
#endif

// XF: the input is act(in1 * in_scale[c] + in_shift[c]) (ssg_conv_desc.in_scale: the batch-norm apply between conv1 and conv2 of a
// residual block, never materialised): applied where a pixel chunk is split into its bf16 terms, constants from an LDS table
// filled once per workgroup; halo pixels outside the image stay zero (the padding of the ACTIVATED tensor).
constexpr int SSG_K32_XF_MAXC = 512;                     // <16, 64>: 154 KiB of image + ring leave 4 KiB for the table

// BWD: the launch is the input gradient that produces d(act(bn(x))) (ssg_conv_desc.bwd_x): the epilogue reads x, masks its result by
// act'(x * scale + shift), writes the masked gradient g and per-tile rows (sum g, sum g * (x - mean)) -- the two sums of a batch-norm backward,
// which a separate pass (ssg_bn_bwd_reduce_f32) would have read g and x again for.
template <int TH, int BN, int WAVES_M, int WAVES_N, bool XF = false, bool BWD = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (WAVES_M * WAVES_N == 8) ? 1 : 2) void conv_halo_k32_kernel(const ConvArgs a) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int TW = 32, BM = TH * TW;
  constexpr int HW = TW + 2, HR = (TH + 2) * HW;
  constexpr int NPIX = (HR + 15) / 16 * 16;
  constexpr int KGS = NPIX * 16 + 64;                    // bytes between the k-groups of a plane: + 64 B so that the 8-byte writes of the four k-groups of a pixel land on distinct banks (reads go 16 lanes = one k-group at a time and do not care)
  constexpr int PLANE = 4 * KGS;                         // bytes of one plane of the pixel image
  constexpr int IMG = (3 * PLANE + 1023) / 1024 * 1024;
  constexpr int BSTG = BN * 192;                         // bytes of one weight stage (one tap x 32 channels x 3 planes)
  constexpr int BPIECES = BSTG / 1024;
  constexpr int B_PC = (BPIECES + NW - 1) / NW;          // per wave and step; piece indices >= BPIECES are dummies (<16, 64>: 12 pieces, 8 waves)
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  // Pixel loads: one instruction = 8 consecutive halo pixels x the chunk's 128 bytes (lane = pixel p8, 16-byte quarter q): eight
  // whole lines.  (The first form -- lane = pixel, 32 bytes of one k-group each -- touched 64 lines per instruction, every line from
  // eight instructions of different waves: with the same bytes in lane order the K-step of the 16 x 32 x 64 tile ran 10-14 % faster
  // and that of the 8 x 32 x 128 tile 3-4 %, tools/k32_probe.py.)
  constexpr int NGRP = (HR + 7) / 8, GPW = (NGRP + NW - 1) / NW, NLD = GPW;
  constexpr int LD_T = 4;                                // tap step at which the next chunk's pixel loads are issued
  static_assert(IMG % 1024 == 0, "the ring behind the image stays 1-KiB aligned");
  static_assert(WTM == 64, "a wave owns two 32-pixel tile rows");

  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  unsigned char* const img = lds;
  unsigned char* const ring = lds + IMG;
  unsigned char* const ldsDummy = ring + 3 * BSTG;       // 1 KiB: target of the dummy pieces
  float* const xtab = (float*)(ldsDummy + 1024);         // XF: [2][C1] scale | shift

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l15 = lane & 15, kg = lane >> 4;

  SSG_STAMP(7);                                            // (probe build) kernel entry; stamp 0 follows the address set-up
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

  const int nchunks = (a.C1 + a.C2) >> 5;
  const int nsteps = nchunks * 9;
  const unsigned OOB = 0xffffffffu;
  const unsigned npix = (unsigned)a.N * (unsigned)a.H * (unsigned)a.W;
  const auto in1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1), 0, (int)(npix * (unsigned)a.ld1 * 4u), 0x00020000);
  const auto in2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in2), 0, (int)(npix * (unsigned)a.ld2 * 4u), 0x00020000);
  const auto w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)((unsigned)nyt * (unsigned)nsteps * (unsigned)BSTG), 0x00020000);
  const unsigned w_tile = (unsigned)nt * (unsigned)nsteps * (unsigned)BSTG;

  // ---- pixel load items of this thread: group (wave * GPW + k) of 8 halo pixels; lane = (pixel p8 of the group, quarter q of the chunk)
  const int p8 = lane >> 3, q8 = lane & 7;
  const int hp0 = wave * GPW * 8 + p8;                   // halo pixel of item 0; item k: + 8 k
  // pixel index of item k in the input tensor, or OOB -- recomputed where it is used (a dozen vector instructions per load, ten loads per
  // 9 K-steps) rather than kept: the hot loop has no registers for a table
  const int pix_base = (n * a.H + ty * TH - 1) * a.W + tx * TW - 1;
  auto px_pix_of = [&](int k) -> unsigned {
    const int hp = hp0 + 8 * k;
    const int hy = hp / HW, hx = hp - hy * HW;
    const bool ok = hp < HR && (unsigned)(ty * TH + hy - 1) < (unsigned)a.H && (unsigned)(tx * TW + hx - 1) < (unsigned)a.W;
    return ok ? (unsigned)(pix_base + hy * a.W + hx) : OOB;
  };
  const int px_dst0 = (q8 >> 1) * KGS + hp0 * 16 + (q8 & 1) * 8;      // byte offset of item 0 in a plane of the image; item k: + 128 k
  u32x4 raw[NLD];
  auto load_px = [&](int chunk) {
    const int c0 = chunk * 32;
    const bool live = chunk < nchunks;                   // past the last chunk: out-of-range lanes keep vmcnt uniform, nothing is fetched
    const bool first = c0 < a.C1;
    const unsigned ld4 = (unsigned)(first ? a.ld1 : a.ld2) * 4u;
    const unsigned so = (unsigned)(first ? c0 : c0 - a.C1) * 4u;
#pragma unroll
    for (int k = 0; k < GPW; ++k) {
      const unsigned pix = px_pix_of(k);
      const unsigned vo = (live && pix != OOB) ? pix * ld4 + (unsigned)q8 * 16u : OOB;
      if (first) raw[k] = __builtin_amdgcn_raw_buffer_load_b128(in1_rs, vo, so, 0);
      else raw[k] = __builtin_amdgcn_raw_buffer_load_b128(in2_rs, vo, so, 0);
    }
  };
  bf16x4 cv[GPW][3];
  auto convert_px = [&](int chunk) {                     // chunk: the one whose loads sit in raw[]
#pragma unroll
    for (int k = 0; k < GPW; ++k) {
      f32x4 v = __builtin_bit_cast(f32x4, raw[k]);
      if constexpr (XF) {
        const int cb = (chunk < nchunks ? chunk : nchunks - 1) * 32 + q8 * 4;       // this lane's 4 channels (C2 == 0)
        v = v * *(const f32x4*)(xtab + cb) + *(const f32x4*)(xtab + a.C1 + cb);     // bn_apply_kernel's expression
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ssg_act(v[e], a.in_act, a.in_slope);
        if (px_pix_of(k) == OOB) v = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      split3_4(v, cv[k][0], cv[k][1], cv[k][2]);
    }
  };
  auto write_px = [&]() {
#pragma unroll
    for (int k = 0; k < GPW; ++k) {
      if (hp0 + 8 * k < HR) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *(bf16x4*)(img + pl * PLANE + px_dst0 + 128 * k) = cv[k][pl];
      }
    }
  };
  auto issue_b = [&](int s) {
    unsigned char* st = ring + (s % 3) * BSTG;
    // past the last step the pieces are dummies that keep vmcnt uniform: they re-read step 0
    const unsigned so = w_tile + (s < nsteps ? (unsigned)s * (unsigned)BSTG : 0u);
#pragma unroll
    for (int j = 0; j < B_PC; ++j) {
      const int g = wave + NW * j;
      if (BPIECES % NW == 0 || g < BPIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (ssg_lds_void*)(st + g * 1024), 16, (unsigned)lane * 16u, so + g * 1024, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (ssg_lds_void*)ldsDummy, 16, OOB, 0, 0, 0);      // every lane out of range: keeps vmcnt uniform
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int pb[MI];                                            // byte offset of this lane's pixel in a plane of the image, tap (0, 0)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int p = wm * WTM + i * 16 + l15;
    pb[i] = kg * KGS + (((p >> 5) + 1) * HW + (p & 31) + 1) * 16;
  }
  const int wfrag = wn * NI * 3 * 1024 + lane * 16;      // this lane's 16 bytes of fragment (wn * NI + j), plane q: + (j * 3 + q) * 1024

  // ---- prologue: pixels of chunk 0 (loads first: they are the oldest vmcnt entries), two weight steps in flight
  SSG_STAMP(0);
  load_px(0);
  issue_b(0);
  issue_b(1);
  if constexpr (XF) {
    for (int i = tid; i < 2 * a.C1; i += NW * 64) xtab[i] = i < a.C1 ? a.in_scale[i] : a.in_shift[i - a.C1];
    __syncthreads();
  }
  wait_vmcnt<2 * B_PC>();
  SSG_STAMP(6);                                            // (probe build) chunk 0's pixels have landed
  convert_px(0);
  write_px();
  SSG_STAMP(1);

#ifdef SSG_K32_PROBE
  unsigned long long probe_own = 0, probe_bar = 0;
#endif
  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int s = chunk * 9 + t;
#ifndef SSG_K32_MIDBAR
#define SSG_K32_MIDBAR 0                                   // 1: the step's barrier sits in the MIDDLE of its MFMA stream (below); 0: at the top.  Measured equal (same-box A/B, +-0.5 % on five shapes): kept as a build switch
#endif
#if SSG_K32_MIDBAR
      // The barrier that makes a weight stage visible does not have to sit where the stage is first read.  The per-wave stamps of the
      // top-of-step form (tools/k32_probe.py) show the two waves of a SIMD running one AFTER the other (waves 0-3 finish their 96
      // MFMAs and sit ~1 500 cycles in the barrier while waves 4-7 run theirs), so behind a top-of-step barrier every wave reads its
      // fragments at once and the matrix pipe idles until the first of them land.  Here the barrier for stage s + 1 is arrived at after
      // the first half of step s's MFMAs: a wave that comes out of it still has half a step of MFMAs queued, and the fast wave of a
      // SIMD reads its next fragments while the slow one multiplies.  Stage s itself was made visible by the barrier in the middle of
      // step s - 1; the slot the pieces of step s + 2 overwrite (stage s - 1) was last read at the top of step s - 1, before that
      // barrier, by every wave.  Only the top of a chunk keeps a barrier of its own: the rewritten pixel image (and, for chunk 0, the
      // first stage) must be visible before tap 0 reads it -- every wave has passed the middle of tap 8, i.e. its last reads of the
      // old image, when the first wave writes.
      if (t == 0) {
        if (chunk > 0) write_px();
        wait_vmcnt<B_PC>();
        wait_lds_reads();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
#else
#ifdef SSG_K32_PROBE
      SSG_PROBE_NOW(pb0);
#endif
      if (t == LD_T + 1 || t == LD_T + 2) wait_vmcnt<B_PC + NLD>();      // the pixel loads issued at LD_T may still be in flight
      else wait_vmcnt<B_PC>();
      wait_lds_reads();
#ifdef SSG_K32_PROBE
      SSG_PROBE_NOW(pb1);
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef SSG_K32_PROBE
      SSG_PROBE_NOW(pb2);
      probe_own += pb1 - pb0; probe_bar += pb2 - pb1;
#endif
      if (t == 0 && chunk > 0) {
        // every wave has left the last tap of the previous chunk: replace the image
        write_px();
        wait_lds_reads();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
#endif
#ifndef SSG_K32_DMA_MID
#define SSG_K32_DMA_MID 1                                  // 1: the step's DMA / pixel loads are issued in the middle of its MFMA stream (A/B build switch)
#endif
      constexpr bool MID = SSG_K32_DMA_MID && TH != 16;   // <16, 64> keeps them at the top: pinned mid-stream, its ten pixel loads per lane spill 15-23 registers
      if constexpr (!MID) {
        issue_b(s + 2);
        if (t == LD_T) load_px(chunk + 1);
      }

      const int tb = (int)((a.tap_bits >> (6 * t)) & 63ull);
      const int toff = (((tb & 7) - 2) * HW + ((tb >> 3) - 2)) * 16;
      const unsigned char* st = ring + (t % 3) * BSTG + wfrag;            // s % 3 == t % 3 (9 steps per chunk)
      bf16x8 p[MI][3], w[NI][3];
#ifndef SSG_K32_READ_ORDER
#define SSG_K32_READ_ORDER 0                               // 1: fragments are read in the order the six product terms consume them (A/B build switch: +-0 on <8,128>, 31 spills on <16,64>)
#endif
#if SSG_K32_READ_ORDER
      // LDS returns reads in order and all eight waves read at once behind the barrier: the planes of the first term (w3, p1) first, so
      // that its MFMAs start after a third of the burst instead of all of it
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int qp = r, qw = 2 - r;                    // term order: (w3, p1), (w2, p2), (w1, p3), ...
#pragma unroll
        for (int i = 0; i < MI; ++i) p[i][qp] = *(const bf16x8*)(img + qp * PLANE + pb[i] + toff);
#pragma unroll
        for (int j = 0; j < NI; ++j) w[j][qw] = *(const bf16x8*)(st + (j * 3 + qw) * 1024);
      }
#else
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) p[i][q] = *(const bf16x8*)(img + q * PLANE + pb[i] + toff);
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) w[j][q] = *(const bf16x8*)(st + (j * 3 + q) * 1024);
#endif
#ifndef SSG_K32_DYNPRIO
#define SSG_K32_DYNPRIO 1                                  // 1: a wave's priority falls as it advances through a step (A/B build switch: +1 % on the <8,128> shapes, the older wave's barrier wait 1 500 -> 700 cycles per step)
#endif
      // The stamps show the older wave of each SIMD pair finishing its 96 MFMAs ~1 500 cycles before the younger one, which then
      // multiplies alone -- and a lone wave issues a 16-pass MFMA only every other slot.  Priority by progress instead of by age: a wave
      // early in its step outranks one that is late in it, so the pair stays together.
      if (SSG_K32_DYNPRIO) __builtin_amdgcn_s_setprio(3);
      // small terms first
#define SSG_K32_TERM(QW, QP)                                                                      \
  _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                  \
  _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][QW], p[i][QP], acc[i][j], 0, 0, 0);
      SSG_K32_TERM(2, 0) SSG_K32_TERM(1, 1)
      if constexpr (MID) {
        // Right after the barrier all eight waves issue their fragment reads at once; an LDS-DMA instruction issued into that burst
        // costs 100-185 cycles, among MFMAs ~60 (MI355X_MICROARCH.md): the pieces of step s + 2 go out after a third of the MFMAs
        // (same-box A/B, 16 images: 240.9 / 260.6 / 273.1 -> 250.8 / 267.4 / 281.8 TFLOP/s on 128 -> 128 at 256^2 / 256 -> 256 at 128^2 / 384 -> 384 at 64^2)
        __builtin_amdgcn_sched_barrier(0);
        issue_b(s + 2);
        if (t == LD_T) load_px(chunk + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (SSG_K32_DYNPRIO) __builtin_amdgcn_s_setprio(2);
      SSG_K32_TERM(0, 2)
#if SSG_K32_MIDBAR
      {
        // stage s + 1 (issued during step s - 1) has landed in this wave; in flight behind it: this step's pieces and, around LD_T, the
        // pixel loads (issued after the pieces of step LD_T, before those of LD_T + 1)
#ifdef SSG_K32_PROBE
        SSG_PROBE_NOW(pb0);
#endif
        if (t == LD_T || t == LD_T + 1) wait_vmcnt<B_PC + NLD>();
        else wait_vmcnt<B_PC>();
        wait_lds_reads();
#ifdef SSG_K32_PROBE
        SSG_PROBE_NOW(pb1);
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef SSG_K32_PROBE
        SSG_PROBE_NOW(pb2);
        probe_own += pb1 - pb0; probe_bar += pb2 - pb1;
#endif
      }
#endif
      if (SSG_K32_DYNPRIO) __builtin_amdgcn_s_setprio(1);
      SSG_K32_TERM(1, 0) SSG_K32_TERM(0, 1)
      if (SSG_K32_DYNPRIO) __builtin_amdgcn_s_setprio(0);
      SSG_K32_TERM(0, 0)
#undef SSG_K32_TERM
      if (t == 8) {                                      // the loads of LD_T landed before tap 7's barrier
#ifndef SSG_K32_CVT_FREE
#define SSG_K32_CVT_FREE 1
#endif
        // <8, 128> lets the scheduler weave the ~130 conversion instructions into tap 8's MFMAs (they fit its registers); the other
        // two tiles pin them behind the last MFMA, where the fragments are dead (woven in, <4, 64> spilled 3 registers)
        if (!(SSG_K32_CVT_FREE && TH == 8 && !XF)) __builtin_amdgcn_sched_barrier(0);
        convert_px(chunk + 1);
      }
    }
  }
  wait_vmcnt<0>();
  wait_lds_reads();
  SSG_STAMP(2);
#ifdef SSG_K32_PROBE
  if (ssg_probe_buf_k32 && lane == 0) {
    ssg_probe_buf_k32[SSG_PROBE_SLOTS * blockIdx.x + 8 + 2 * wave] = probe_own;
    ssg_probe_buf_k32[SSG_PROBE_SLOTS * blockIdx.x + 9 + 2 * wave] = probe_bar;
  }
#endif

  // ---- non-finite operands (conv_slow.h): a workgroup that holds a non-finite accumulator recomputes its tile with fp32 FMAs
  {
    bool bad = false;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= ssg_nonfinite(acc[i][j][r]);
    if (__builtin_amdgcn_readfirstlane(__syncthreads_or(bad))) {     // scalar condition: a uniform branch, the accumulators are dead inside it                         // also: every wave has left the main loop, LDS is scratch
      float* scr = (float*)lds + tid;                    // value e of this thread at scr[e * threads]
      // the argument block is re-read from the kernarg segment HERE (opaque pointer): kept live in SGPRs across the main loop for
      // this cold path, its fields spilled 6 registers of the hot loop
      const ConvArgs* ap = (const ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ap));
      const ConvArgs& as = *ap;
      for (int e = 0; e < MI * NI * 4; ++e) {
        const int i = e / (NI * 4), j = (e >> 2) % NI, r = e & 3;
        const int p = wm * WTM + i * 16 + l15;
        const int co = n0 + wn * WTN + j * 16 + kg * 4 + r;
        scr[e * (NW * 64)] = co < as.Cout ? ssg_conv_slow_value<XF>(as, n, ty * TH + (p >> 5), tx * TW + (p & 31), co, 0, 9) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = scr[((i * NI + j) * 4 + r) * (NW * 64)];
    }
  }

  SSG_STAMP(24);                                           // (probe build) the non-finite check is done
  // ---- epilogue.  acc[i][j][r]: pixel p = wm*64 + i*16 + l15, output channel n0 + wn*WTN + j*16 + kg*4 + r.
  const bool want_bn = a.bnpart != nullptr;
  if (want_bn) __syncthreads();                          // the image and the ring are dead for every wave: LDS becomes scratch
  double* const red = (double*)lds;                      // [WAVES_M][2][BN]
  size_t opix[MI]; bool pok[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int p = wm * WTM + i * 16 + l15;
    const int gy = ty * TH + (p >> 5), gx = tx * TW + (p & 31);
    pok[i] = gy < a.GH && gx < a.GW;
    opix[i] = (size_t)(n * a.OH + gy * a.out_sy + a.out_oy) * a.OW + gx * a.out_sx + a.out_ox;
  }
  float nvl = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i) nvl += pok[i] ? 1.f : 0.f;
  if (want_bn) {
    nvl = ssg_row16_sum(nvl);                              // valid pixels of this wave's 16-lane row group
  }
  // BWD: every x quad of this lane's outputs is requested before the first is used (the fragment registers of the main loop are dead: 64 free
  // registers) -- one load latency for the tile's epilogue instead of one per column fragment
  f32x4 bx[BWD ? MI : 1][BWD ? NI : 1];
  if constexpr (BWD) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        bx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pok[i]) bx[i][j] = *(const f32x4*)(a.bwd_x + opix[i] * a.bwd_ldx + n0 + wn * WTN + j * 16 + kg * 4);
      }
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = n0 + wn * WTN + j * 16 + kg * 4;
    const bool cok = BN >= 64 || co < a.Cout;              // narrow tiles: padding columns (Cout % 4 == 0: whole quads)
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && cok) bv = *(const f32x4*)(a.bias + co);
    if constexpr (BWD) {
      // backward statistics: g = v * act'(z), z = x * scale + shift recomputed with bn_apply_kernel's expression (same bits, same sign);
      // sums of g and g * (x - mean) over the wave's pixels, fp32 inside the 16-lane group, fp64 beyond; acc is overwritten by g
      const f32x4 ksc = *(const f32x4*)(a.bwd_scale + co), ksh = *(const f32x4*)(a.bwd_shift + co), kmu = *(const f32x4*)(a.bwd_mean + co);
      f32x4 s1v = {0.f, 0.f, 0.f, 0.f}, s2v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const f32x4 xv = bx[i][j];
        const f32x4 z = xv * ksc + ksh;
        f32x4 g = acc[i][j] + bv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (!(z[r] > 0.f)) g[r] *= (a.bwd_act == SSG_ACT_RELU ? 0.f : (a.bwd_act == SSG_ACT_LRELU ? a.bwd_slope : 1.f));
          if (!pok[i]) g[r] = 0.f;
        }
        acc[i][j] = g - bv;                              // the store below adds bv back
        s1v += g; s2v += g * (xv - kmu);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = ssg_row16_sum(s1v[r]), s2 = ssg_row16_sum(s2v[r]);
        if (l15 == 0) {
          const int col = wn * WTN + j * 16 + kg * 4 + r;
          red[(wm * 2 + 0) * BN + col] = (double)s1;
          red[(wm * 2 + 1) * BN + col] = (double)s2;
        }
      }
    } else if (want_bn) {
      // Column sums for the batch-norm statistics: fp32 sums of DEVIATIONS from a pivot shared by the 16 lanes of a row group (the
      // group's first value of the column), converted to sums of the values in fp64 once per wave and column:
      // S1 = s1 + n*c, S2 = s2 + 2*c*s1 + n*c^2 (plain fp32 sums of v and v^2 lose var = E[v^2] - mean^2 once |mean| >> std).
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float piv = ssg_row16_sum(acc[0][j][r] + bv[r]) * 0.0625f;   // any value the 16 lanes share will do: the mean of their first pixels (bitwise the same in every lane: each step adds a lane and its partner)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const float dv = pok[i] ? (acc[i][j][r] + bv[r]) - piv : 0.f;
          s1 += dv; s2 += dv * dv;
        }
        s1 = ssg_row16_sum(s1); s2 = ssg_row16_sum(s2);
        if (l15 == 0) {
          const double c = (double)piv, nn = (double)nvl;
          const int col = wn * WTN + j * 16 + kg * 4 + r;
          red[(wm * 2 + 0) * BN + col] = (double)s1 + nn * c;
          red[(wm * 2 + 1) * BN + col] = (double)s2 + 2.0 * c * (double)s1 + nn * c * c;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (!pok[i] || !cok) continue;
      f32x4 v = acc[i][j] + bv;
      if (a.res) v += *(const f32x4*)(a.res + opix[i] * a.ldr + co);
      if (a.act == SSG_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];
      } else if (a.act == SSG_ACT_LRELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * a.slope;
      }
      *(f32x4*)(a.out + opix[i] * a.ldo + co) = v;
    }
  }
  if (want_bn) {
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      double t1 = 0, t2 = 0;
#pragma unroll
      for (int k = 0; k < WAVES_M; ++k) { t1 += red[(k * 2 + 0) * BN + tid]; t2 += red[(k * 2 + 1) * BN + tid]; }
      double* dst = a.bnpart + (size_t)((n * a.tiles_y + ty) * a.tiles_x + tx) * 2 * a.Cout;
      dst[n0 + tid] = t1; dst[a.Cout + n0 + tid] = t2;
    }
  }
#ifdef SSG_K32_PROBE
  SSG_STAMP(25);                                           // every store has been issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the output stores have left
  SSG_STAMP(3);
#endif
}

// fp32 packed [R][Kp] (kmode 0 with 9 taps: k = (chunk16 * 9 + tap) * 16 + c) -> [R / BN][chunk32 * 9 + tap][BN / 16 fragments][3 planes][64 lanes][16 B]:
// lane l of fragment j holds output channel j*16 + (l & 15), channels chunk32*32 + (l >> 4)*8 .. +7.  One thread per (row, step, k-group).
__global__ __launch_bounds__(256) void pack_split_k32_kernel(const float* __restrict__ w, int R, int Kp, int BN, unsigned char* __restrict__ out) {
  const int nsteps = Kp >> 5;                            // 32-channel steps
  const long long total = (long long)((R + BN - 1) / BN) * BN * nsteps * 4;       // rows beyond R (narrow tiles): zero weights
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int g = (int)(i & 3);
    long long t = i >> 2;
    const int s = (int)(t % nsteps); t /= nsteps;
    const int row = (int)t;
    const int tile = row / BN, rl = row - tile * BN;
    const int chunk32 = s / 9, tap = s - chunk32 * 9;
    const int chunk16 = chunk32 * 2 + (g >> 1);
    const float* src = w + (size_t)row * Kp + (size_t)(chunk16 * 9 + tap) * 16 + (g & 1) * 8;
    bf16x8 p1, p2, p3;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    split3(row < R ? *(const f32x4*)src : z, row < R ? *(const f32x4*)(src + 4) : z, p1, p2, p3);
    const int j = rl >> 4, l = (rl & 15) + 16 * g;
    unsigned char* dst = out + ((size_t)tile * nsteps + s) * BN * 192 + (size_t)j * 3 * 1024 + (size_t)l * 16;
    *(bf16x8*)(dst) = p1; *(bf16x8*)(dst + 1024) = p2; *(bf16x8*)(dst + 2048) = p3;
  }
}

template <int TH, int BN, int WAVES_M, int WAVES_N, bool XF = false, bool BWD = false>
int launch(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int HR = (TH + 2) * 34, NPIX = (HR + 15) / 16 * 16;
  a.tiles_x = (a.GW + 31) / 32;
  a.tiles_y = (a.GH + TH - 1) / TH;
  static const int swz = [] { const char* e = getenv("SSG_XCD_SWIZZLE"); return e ? atoi(e) : 1; }();
  a.xcd_swizzle = swz;
  a.ntiles_n = (a.Cout + BN - 1) / BN;                    // narrow tiles (BN 16 / 32): the last columns may be padding
  dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N * a.ntiles_n));
  constexpr int lds_bytes = (3 * 4 * (NPIX * 16 + 64) + 1023) / 1024 * 1024 + 3 * BN * 192 + 1024;
  static_assert(lds_bytes <= 160 * 1024 && (NW != 4 || BN <= 32 || lds_bytes <= 80 * 1024), "LDS budget");
  constexpr int tab_bytes = XF ? 2 * SSG_K32_XF_MAXC * 4 : 0;       // scale | shift of up to SSG_K32_XF_MAXC input channels
  static_assert(lds_bytes + tab_bytes <= 160 * 1024, "LDS budget with the input-transform table");
  static const hipError_t attr = hipFuncSetAttribute((const void*)conv_halo_k32_kernel<TH, BN, WAVES_M, WAVES_N, XF, BWD>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes + tab_bytes);
  if (attr != hipSuccess) { ssg_set_error("conv halo k32: LDS attribute: %s", hipGetErrorString(attr)); return (int)attr; }
  hipLaunchKernelGGL((conv_halo_k32_kernel<TH, BN, WAVES_M, WAVES_N, XF, BWD>), grid, dim3(NW * 64), lds_bytes + (XF ? 2 * a.C1 * 4 : 0), st, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

}  // namespace

#ifdef SSG_K32_PROBE
extern "C" int ssg_debug_set_probe_buffer_k32(void* p) {
  unsigned long long* v = (unsigned long long*)p;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(ssg_probe_buf_k32), &v, sizeof(v));
}
#endif

// Split-pack format code of the k32 kernel for this launch, or 0 when it does not take it: 1128 = 8 x 32-pixel tiles x 128
// channels (512 threads, one workgroup per CU), 1064 = 4 x 32 x 64 (256 threads, two per CU), 2064 = 16 x 32 x 64 (512 threads; the
// weights are packed as for 1064).  Needs whole 32-channel chunks on
// both inputs, whole column tiles, 16-byte-aligned rows everywhere, and -- a tile per CU being a lot of work -- a grid that
// fills the chip evenly.  SSG_K32=0 switches the family off (A/B), SSG_K32=2 forces it where the shape is legal.
static int g_k32_mode = -1;                               // -1: not read yet; SSG_K32 or ssg_conv_set_k32_mode
extern "C" int ssg_conv_set_k32_mode(int mode) {          // 0 = off, 1 = where the grid fills the chip (default), 2 = wherever legal (tests)
  SSG_REQUIRE(mode >= 0 && mode <= 2, SSG_EINVAL, "k32 mode %d", mode);
  g_k32_mode = mode;
  return SSG_OK;
}

int ssg_conv_halo_k32_fmt(const ConvArgs& a) {
  if (g_k32_mode < 0) { const char* e = getenv("SSG_K32"); g_k32_mode = e ? atoi(e) : 1; }
  const int on = g_k32_mode;
  if (!on || !ssg_conv_halo_ok(a) || a.parity) return 0;
  if ((a.C1 & 31) || (a.C2 & 31) || a.GW < 17) return 0;
  if ((a.ldo & 3) || ((uintptr_t)a.out & 15) || (a.res && ((a.ldr & 3) || ((uintptr_t)a.res & 15))) || (a.bias && ((uintptr_t)a.bias & 15))) return 0;
  const unsigned long long bytes = (unsigned long long)a.N * a.H * a.W * (unsigned long long)(a.ld1 > a.ld2 ? a.ld1 : a.ld2) * 4ull;
  if (bytes > 0xfffffff0ull) return 0;                  // 32-bit byte offsets of the buffer descriptors
  if (a.Cout <= 32) {
    // narrow layers (SPADE's x -> segmentation-map convs: 512 -> 16 at 128^2, 768 -> 24 at 64^2): 8 x 32-pixel tiles, all output
    // channels in one 16- or 32-column tile, 256 threads.  Two LDS fragment reads per four MFMAs instead of one: the LDS pipe is the
    // bound, at several times the rate of the fp32 256 x 32 register kernel these shapes ran on (41-44 TFLOP/s).
    static const int narrow = [] { const char* e = getenv("SSG_K32_NARROW"); return e ? atoi(e) : 1; }();
    const long long wgs = (long long)a.N * ((a.GH + 7) / 8) * ((a.GW + 31) / 32);
    if (!narrow || a.Cout < 12 || (a.Cout & 3) || a.C1 + a.C2 < 128) return 0;
    return (on == 2 || wgs >= 192) ? (a.Cout <= 16 ? 1016 : 1032) : 0;
  }
  if (a.Cout % 128 == 0) {
    const long long wgs = (long long)a.N * ((a.GH + 7) / 8) * ((a.GW + 31) / 32) * (a.Cout / 128);
    const long long waves = (wgs + 255) / 256;
    if (on == 2 || wgs >= 2048 || (wgs >= 256 && wgs * 10 >= waves * 256 * 8)) return 1128;   // >= 80 % of the last wave of tiles filled
  }
  if (a.Cout % 64 == 0) {
    static const int t16 = [] { const char* e = getenv("SSG_K32_T16"); return e ? atoi(e) : 1; }();
    const long long wgs16 = (long long)a.N * ((a.GH + 15) / 16) * ((a.GW + 31) / 32) * (a.Cout / 64);
    if (t16 && !a.in_scale && a.GH >= 16 && (wgs16 >= 2048 || (on == 2 && a.GH % 16 == 0))) return 2064;   // (with the fused input transform the 16-row tile spills: those launches take <4,64>)   // 16 x 32-pixel x 64-channel tiles, 512 threads (same pack as 1064)
    const long long wgs = (long long)a.N * ((a.GH + 3) / 4) * ((a.GW + 31) / 32) * (a.Cout / 64);
    if (on == 2 || wgs >= 1536) return 1064;
  }
  return 0;
}

void ssg_conv_halo_k32_tile(int fmt, int* th, int* tw) { *tw = 32; *th = (fmt == 1128 || fmt == 1016 || fmt == 1032) ? 8 : (fmt == 2064 ? 16 : 4); }

// fused input transform (ssg_conv_desc.in_scale): the <8,128> and <4,64> tiles, one input pointer, a table that fits beside the image
bool ssg_conv_halo_k32_in_affine_ok(const ConvArgs& a, int fmt) {
  if (fmt != 1128 && fmt != 1064) return false;
  if (a.C2 != 0 || a.C1 > SSG_K32_XF_MAXC) return false;
  return a.in_act == SSG_ACT_NONE || a.in_act == SSG_ACT_RELU || a.in_act == SSG_ACT_LRELU;
}

// backward-statistics epilogue (ssg_conv_desc.bwd_x): the three wide tiles, whole 4-channel quads everywhere
bool ssg_conv_halo_k32_bwd_stats_ok(const ConvArgs& a, int fmt) {
  if (fmt != 1128 && fmt != 1064 && fmt != 2064) return false;
  if ((a.bwd_ldx & 3) || ((uintptr_t)a.bwd_x & 15) || ((uintptr_t)a.bwd_scale & 15) || ((uintptr_t)a.bwd_shift & 15) || ((uintptr_t)a.bwd_mean & 15)) return false;
  return a.bwd_act == SSG_ACT_NONE || a.bwd_act == SSG_ACT_RELU || a.bwd_act == SSG_ACT_LRELU;
}

int ssg_conv_igemm_halo_k32_launch(const ConvArgs& a, int fmt, hipStream_t st) {
  if (a.bwd_x) {
    if (!a.bnpart || a.in_scale || !ssg_conv_halo_k32_bwd_stats_ok(a, fmt)) { ssg_set_error("conv halo k32: bwd_x on a launch without the backward-statistics epilogue"); return SSG_EINVAL; }
    if (fmt == 1128) return launch<8, 128, 4, 2, false, true>(a, st);
    if (fmt == 1064) return launch<4, 64, 2, 2, false, true>(a, st);
    return launch<16, 64, 8, 1, false, true>(a, st);
  }
  if (a.in_scale) {
    if (!a.in_shift || !ssg_conv_halo_k32_in_affine_ok(a, fmt)) { ssg_set_error("conv halo k32: in_scale on a launch without the fused input transform"); return SSG_EINVAL; }
    if (fmt == 1128) return launch<8, 128, 4, 2, true>(a, st);
    return launch<4, 64, 2, 2, true>(a, st);
  }
  if (fmt == 1128) return launch<8, 128, 4, 2>(a, st);
  if (fmt == 1064) return launch<4, 64, 2, 2>(a, st);
  if (fmt == 2064) return launch<16, 64, 8, 1>(a, st);
  if (fmt == 1016) return launch<8, 16, 4, 1>(a, st);
  if (fmt == 1032) return launch<8, 32, 4, 1>(a, st);
  ssg_set_error("conv halo k32: unknown format %d", fmt);
  return SSG_EINVAL;
}

int ssg_pack_split_k32_launch(const float* w_packed, int R, int Kp, int BN, void* out, hipStream_t st) {
  const long long total = (long long)((R + BN - 1) / BN) * BN * (Kp >> 5) * 4;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_split_k32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w_packed, R, Kp, BN, (unsigned char*)out);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
