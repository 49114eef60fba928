// bf16 pointwise (1x1) convolution = GEMM on v_mfma_f32_32x32x16_bf16 (gfx950), fp32 accumulate.
// ABI + reference citations: include/ssunet_hip.h (ssg_gemm_bf16, ssg_gemm_wgrad_bf16, ssg_pack_weights_bf16).
//
// These are the dense layers of the EfficientNet MBConv blocks (efficientnet_pytorch/model.py:40-58: expand, project,
// head), 88-92 % of that encoder's FLOPs.  At B4 / 1024^2 their arithmetic intensity is 10-400 FLOP/B against a bf16
// ridge of ~400: almost all of them are HBM-bound, so the kernels are built to stream the activation tensor once with
// 16-B-per-lane LDS-DMA and to keep the weight panel in L2/LDS; the MFMA work rides along.
//
//   forward / input gradient   out[p][n] = sum_k x[p][k] * w[n][k]   (+ res[p][n])
//       operands: x NHWC-with-stride bf16 (k contiguous), w packed bf16 [n][Kp] (k contiguous): both are "row holds k"
//       images, so both MFMA fragments are plain ds_read_b128 of one 64-B LDS row per lane.  The WEIGHTS are the A
//       operand and the PIXELS the B operand: D[m = channel][n = pixel] then leaves each lane with 4 consecutive output
//       channels of one pixel per register quad -> one packed 8-byte bf16 store per quad.
//   weight gradient            dw[m][n] = sum_p dy[p][m] * x[p][n]
//       both operands reduce over PIXELS, which are the strided axis of NHWC: the tiles are staged [pixel][channel] as
//       they lie in memory and the fragments are read with ds_read_b64_tr_b16 (hardware transpose: 4 pixels x 16
//       channels per 16-lane group, delivered channel-major).  Split-K over pixel ranges, fp32 slabs, ordered reduce.
#include "common.h"
#include "lds_dma.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(128))) unsigned short ssg_zero_page_bf16[64];

__device__ __forceinline__ void dma16h(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((ssg_gbl_void*)src, (ssg_lds_void*)lds_dst, 16, 0, 0);
}

struct GemmArgs {
  const unsigned short* x; const unsigned short* w; const unsigned short* res; unsigned short* out;
  long long P; int K, N;            // pixels, reduced channels (valid, % 8 == 0), output channels (valid, % 8 == 0)
  int ldx, ldo, ldr, Kp;            // pixel strides (elements), packed-weight row length (% 32 == 0)
  int nsteps, ntiles_n;
};

constexpr int GB_M = 128;           // pixels per tile
constexpr int GB_N = 128;           // output channels per tile
constexpr int G_STAGES = 3;
constexpr int G_STAGE_BYTES = (GB_M + GB_N) * 64;     // 32 bf16 = 64 B per row

// ---------------------------------------------------------------- forward / dgrad
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;             // wm: channel half (A operand), wn: pixel half (B operand)

  int bid = blockIdx.x;
  {
    const int per = (int)gridDim.x >> 3;                // XCD-contiguous tile runs (bijective on the full part)
    if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
  }
  const int n0 = (bid % a.ntiles_n) * GB_N;
  const long long p0 = (long long)(bid / a.ntiles_n) * GB_M;

  // DMA source state: piece j of this wave = tile rows [(wave*2 + j)*16, +16); lane l -> row +(l>>2), LDS 16-B slot l&3,
  // which holds k-chunk (l&3) ^ ((row>>2)&3)
  const int lr = lane >> 2, lp = lane & 3;
  const unsigned short* xs[2]; const unsigned short* ws[2];
  int kq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 16 + lr;
    const int q = lp ^ ((r >> 2) & 3);
    kq[j] = 8 * q;
    xs[j] = (p0 + r < a.P) ? a.x + (size_t)(p0 + r) * a.ldx + 8 * q : nullptr;
    ws[j] = (n0 + r < a.N) ? a.w + (size_t)(n0 + r) * a.Kp + 8 * q : nullptr;   // rows past N read the zero page: no assumption about the pack's row padding
  }
  const unsigned short* zero = ssg_zero_page_bf16;

  auto issue = [&](int s) {
    unsigned char* st = lds + (s % G_STAGES) * G_STAGE_BYTES;
    const int k0 = s * 32;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned short* p = (xs[j] && k0 + kq[j] < a.K) ? xs[j] + k0 : zero;
      dma16h(p, st + GB_N * 64 + (wave * 2 + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16h(ws[j] ? ws[j] + k0 : zero, st + (wave * 2 + j) * 1024);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int half = lane >> 5, l31 = lane & 31;
  const int sw = (l31 >> 2) & 3;
  const int nsteps = a.nsteps;
  if (0 < nsteps) issue(0);
  if (1 < nsteps) issue(1);
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) wait_vmcnt<4>(); else wait_vmcnt<0>();
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 2 < nsteps) issue(s + 2);
    const unsigned char* st = lds + (s % G_STAGES) * G_STAGE_BYTES;
    const unsigned char* Wb = st + (wm * 64 + l31) * 64;
    const unsigned char* Xb = st + GB_N * 64 + (wn * 64 + l31) * 64;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int off = 16 * ((2 * kk + half) ^ sw);
      bf16x8 fw[2], fx[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fw[i] = *(const bf16x8*)(Wb + i * 32 * 64 + off);
#pragma unroll
      for (int j = 0; j < 2; ++j) fx[j] = *(const bf16x8*)(Xb + j * 32 * 64 + off);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
    }
  }

  // epilogue: D[m = channel][n = pixel]; lane = pixel column l31, register r = channel (r&3) + 8*(r>>2) + 4*half
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long long pix = p0 + wn * 64 + j * 32 + l31;
    if (pix >= a.P) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = n0 + wm * 64 + i * 32 + 8 * g + 4 * half;
        if (co >= a.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
        if (a.res) {
          const bf16x4 rv = *(const bf16x4*)(a.res + (size_t)pix * a.ldr + co);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        *(bf16x4*)(a.out + (size_t)pix * a.ldo + co) = o;
      }
    }
  }
}

// ---------------------------------------------------------------- weight gradient
struct WgArgsH {
  const unsigned short* dy; const unsigned short* x; float* slabs;
  long long P; int M, N;            // pixels, dy channels (rows of dw), x channels (columns of dw); both % 8 == 0
  int ldd, ldx;
  int tiles_m, tiles_n, splits; long long steps_per_split;   // K-steps of 32 pixels per split
};

constexpr int W_STAGE_BYTES = 4 * 32 * 128;       // 2 tensors x 2 sub-images of [32 pixels][64 channels]
constexpr int W_STAGES = 3;

__global__ __launch_bounds__(256) void gemm_wgrad_bf16_kernel(const WgArgsH a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int bid = blockIdx.x;
  const int tn = bid % a.tiles_n; bid /= a.tiles_n;
  const int tm = bid % a.tiles_m; bid /= a.tiles_m;
  const int split = bid;
  const int m0 = tm * 128, n0 = tn * 128;
  const long long step0 = (long long)split * a.steps_per_split;
  long long nst = (a.P + 31) / 32 - step0;
  if (nst > a.steps_per_split) nst = a.steps_per_split;
  const int nsteps = nst > 0 ? (int)nst : 0;

  // DMA: a stage holds 16 pieces of 8 pixels x 128 B: piece id = tensor*8 + sub*4 + pq (pq: pixel octet 0..3).
  // Wave w issues pieces 4w .. 4w+3, i.e. tensor = w>>1, sub = w&1, pq = j.  Lane l: pixel pl = l>>3, LDS 16-B slot
  // l&7, which holds channel chunk (l&7) ^ (((pl>>1)&1)<<2).
  const int pl = lane >> 3;
  const int chunk = (lane & 7) ^ (((pl >> 1) & 1) << 2);
  const int tensor = wave >> 1, sub = wave & 1;
  const unsigned short* base = tensor == 0 ? a.dy : a.x;
  const int ld = tensor == 0 ? a.ldd : a.ldx;
  const int ch = (tensor == 0 ? m0 : n0) + sub * 64 + 8 * chunk;
  const bool ch_ok = ch < (tensor == 0 ? a.M : a.N);
  const unsigned short* zero = ssg_zero_page_bf16;

  auto issue = [&](int s) {
    unsigned char* st = lds + (s % W_STAGES) * W_STAGE_BYTES + (tensor * 2 + sub) * (32 * 128);
    const long long pb = (step0 + s) * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long p = pb + j * 8 + pl;
      const unsigned short* src = (ch_ok && p < a.P) ? base + (size_t)p * ld + ch : zero;
      dma16h(src, st + j * 1024);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read addressing (ds_read_b64_tr_b16): 16-lane group g = lane>>4 takes channels 16*(g&1) .. +15 of the
  // 32-channel MFMA block and pixels 8*(g>>1) + 4t .. +3 (t = 0, 1) of the 16-pixel k-step; lane 4q+p of the group
  // supplies the address of pixel row q, channels 4p .. 4p+3.
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int cblk = 16 * (g & 1) + 4 * pp;                 // channel offset inside the 32-channel block
  const int prow = 8 * (g >> 1) + q;                      // pixel row inside the k-step (t adds 4)
  // byte offset inside a [32 px][64 ch] sub-image for (pixel, channel c in 0..63):
  //   pixel*128 + ((c>>3) ^ (((pixel>>1)&1)<<2))*16 + (c&7)*2;   (pixel>>1)&1 == (q>>1)&1 for every t, kk
  const int swz = ((q >> 1) & 1) << 2;
  int offA[2], offB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = (wm * 64 + i * 32 + cblk) & 63;        // channel inside its 64-channel sub-image
    offA[i] = (0 * 2 + ((wm * 64 + i * 32) >> 6)) * (32 * 128) + prow * 128 + (((ca >> 3) ^ swz) << 4) + (ca & 7) * 2;
    const int cb = (wn * 64 + i * 32 + cblk) & 63;
    offB[i] = (1 * 2 + ((wn * 64 + i * 32) >> 6)) * (32 * 128) + prow * 128 + (((cb >> 3) ^ swz) << 4) + (cb & 7) * 2;
  }

  if (0 < nsteps) issue(0);
  if (1 < nsteps) issue(1);
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) wait_vmcnt<4>(); else wait_vmcnt<0>();
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 2 < nsteps) issue(s + 2);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    unsigned char* stb = lds + (s % W_STAGES) * W_STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      s16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(stb + offA[i] + (kk * 16 + t * 4) * 128));
          const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(stb + offB[i] + (kk * 16 + t * 4) * 128));
#pragma unroll
          for (int e = 0; e < 4; ++e) { fa[i][4 * t + e] = va[e]; fb[i][4 * t + e] = vb[e]; }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
    }
  }

  // slab [split][M][N] (valid part only); D col = lane&31 = n, row = (r&3) + 8*(r>>2) + 4*half = m
  const int half = lane >> 5, l31 = lane & 31;
  float* slab = a.slabs + (size_t)split * a.M * a.N;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
    if (n >= a.N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < a.M) slab[(size_t)m * a.N + n] = acc[i][j][r];
      }
    }
  }
}

// Ordered slab reduce.  One thread per output when there are few slabs; with many (thin layers: up to 2048 slabs of a few KB)
// 32 lanes share an output: lane k adds slabs k, k+32, ... in order, then the 32 lane sums are folded by a fixed xor tree --
// a fixed summation order either way (bitwise reproducible), 1/32 of the serial chain.
__global__ __launch_bounds__(256) void gemm_wgrad_reduce_kernel(const float* __restrict__ slabs, int splits, long long MN, float* __restrict__ dw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= MN) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += slabs[(size_t)k * MN + i];
  dw[i] = s;
}
__global__ __launch_bounds__(256) void gemm_wgrad_reduce32_kernel(const float* __restrict__ slabs, int splits, long long MN, float* __restrict__ dw) {
  const int lane = threadIdx.x & 31;
  const long long i = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
  float s = 0.f;
  if (i < MN)
    for (int k = lane; k < splits; k += 32) s += slabs[(size_t)k * MN + i];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
  if (lane == 0 && i < MN) dw[i] = s;
}

// w fp32 [O][I] -> bf16 [rows_pad][Kp], rows = O (transpose 0) or I (transpose 1), zero padded
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ w, int O, int I, int transpose, int rows_pad, int Kp,
                                                        unsigned short* __restrict__ out) {
  const long long total = (long long)rows_pad * Kp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / Kp), k = (int)(i - (long long)r * Kp);
    const int rows = transpose ? I : O, cols = transpose ? O : I;
    float v = 0.f;
    if (r < rows && k < cols) v = transpose ? w[(size_t)k * I + r] : w[(size_t)r * I + k];
    const __bf16 b = (__bf16)v;
    out[i] = __builtin_bit_cast(unsigned short, b);
  }
}

int wg_splits(long long P, int M, int N) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  const long long nk = (P + 31) / 32;
  // ~512 workgroups when the slabs are large; up to ~2048 when a slab is a few KB (thin layers over 10^5..10^6 pixels:
  // with 512 the 24 x 24 gradient over 1 M pixels ran on 256 workgroups at 0.6 TB/s)
  const long long slab_bytes = (long long)M * N * 4;
  long long target = slab_bytes <= (64 << 10) ? 2048 : (slab_bytes <= (512 << 10) ? 1024 : 512);
  long long s = target / (tiles > 0 ? tiles : 1);
  if (s > nk / 8) s = nk / 8;                 // at least 8 K-steps per split
  if (s < 1) s = 1;
  if (s > 2048) s = 2048;
  return (int)s;
}

}  // namespace

extern "C" int ssg_gemm_bf16(const void* x, int64_t P, int K, int ldx, const void* w_packed, int Kp, int N, const void* res, int ldr,
                             void* out, int ldo, void* stream) {
  SSG_REQUIRE(x && w_packed && out && P > 0 && K > 0 && N > 0, SSG_EINVAL, "gemm_bf16: bad args");
  SSG_REQUIRE(K % 8 == 0 && N % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && Kp % 32 == 0 && Kp >= K && (!res || ldr % 8 == 0) &&
                  ssg_aligned16(x) && ssg_aligned16(w_packed) && ssg_aligned16(out) && (!res || ssg_aligned16(res)),
              SSG_EALIGN, "gemm_bf16: channel counts / strides must be multiples of 8 and pointers 16-B aligned");
  GemmArgs a;
  a.x = (const unsigned short*)x; a.w = (const unsigned short*)w_packed; a.res = (const unsigned short*)res; a.out = (unsigned short*)out;
  a.P = P; a.K = K; a.N = N; a.ldx = ldx; a.ldo = ldo; a.ldr = ldr; a.Kp = Kp;
  a.nsteps = (K + 31) / 32;
  a.ntiles_n = (N + GB_N - 1) / GB_N;
  const long long tiles = (long long)a.ntiles_n * ((P + GB_M - 1) / GB_M);
  SSG_REQUIRE(tiles < (1ll << 31), SSG_EINVAL, "gemm_bf16: grid too large");
  hipLaunchKernelGGL(gemm_bf16_kernel, dim3((unsigned)tiles), dim3(256), G_STAGES * G_STAGE_BYTES, (hipStream_t)stream, a);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int64_t ssg_gemm_wgrad_bf16_workspace_bytes(int64_t P, int M, int N) {
  return (int64_t)wg_splits(P, M, N) * M * N * (int64_t)sizeof(float);
}

extern "C" int ssg_gemm_wgrad_bf16(const void* dy, int ldd, const void* x, int ldx, int64_t P, int M, int N, float* dw, void* ws,
                                   int64_t ws_bytes, void* stream) {
  SSG_REQUIRE(dy && x && dw && ws && P > 0 && M > 0 && N > 0, SSG_EINVAL, "gemm_wgrad_bf16: bad args");
  SSG_REQUIRE(M % 8 == 0 && N % 8 == 0 && ldd % 8 == 0 && ldx % 8 == 0 && ssg_aligned16(dy) && ssg_aligned16(x), SSG_EALIGN,
              "gemm_wgrad_bf16: channel counts / strides must be multiples of 8 and pointers 16-B aligned");
  WgArgsH a;
  a.dy = (const unsigned short*)dy; a.x = (const unsigned short*)x; a.slabs = (float*)ws;
  a.P = P; a.M = M; a.N = N; a.ldd = ldd; a.ldx = ldx;
  a.tiles_m = (M + 127) / 128; a.tiles_n = (N + 127) / 128;
  a.splits = wg_splits(P, M, N);
  SSG_REQUIRE(ws_bytes >= (int64_t)a.splits * M * N * (int64_t)sizeof(float), SSG_EINVAL, "gemm_wgrad_bf16: workspace too small");
  const long long nk = (P + 31) / 32;
  a.steps_per_split = (nk + a.splits - 1) / a.splits;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gemm_wgrad_bf16_kernel, dim3((unsigned)(a.tiles_m * a.tiles_n * a.splits)), dim3(256), W_STAGES * W_STAGE_BYTES, st, a);
  SSG_LAUNCH_CHECK();
  const long long MN = (long long)M * N;
  if (a.splits >= 64 && MN <= (1 << 18))
    hipLaunchKernelGGL(gemm_wgrad_reduce32_kernel, dim3((unsigned)((MN + 7) / 8)), dim3(256), 0, st, (const float*)ws, a.splits, MN, dw);
  else
    hipLaunchKernelGGL(gemm_wgrad_reduce_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, st, (const float*)ws, a.splits, MN, dw);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_pack_weights_bf16(const float* w, int O, int I, int transpose, int rows_pad, int Kp, void* out, void* stream) {
  const int rows = transpose ? I : O, cols = transpose ? O : I;
  SSG_REQUIRE(w && out && O > 0 && I > 0 && rows_pad >= rows && rows_pad % 128 == 0 && Kp >= cols && Kp % 32 == 0, SSG_EINVAL,
              "pack_weights_bf16: bad args");
  long long g = ((long long)rows_pad * Kp + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, O, I, transpose, rows_pad, Kp,
                     (unsigned short*)out);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
