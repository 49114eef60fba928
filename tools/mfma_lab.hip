// Stand-alone main-loop laboratory for the split-operand (bf16x3) convolution kernels on gfx950.  Not part of the library:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_lab.hip -o tools/mfma_lab     (cross-compiles without a GPU)
//   ./tools/mfma_lab [seconds-per-case]
// It times, on RANDOM operands (the bf16 pipe is power-limited: constants flatter it), the executed bf16 TFLOP/s of
//   bare32 / bare16  register-only v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 loops (1 or 2 waves per SIMD)
//   ring<...>        the skeleton of a halo conv main loop: a pixel image resident in LDS, a weight ring filled by LDS-DMA from
//                    an L2-resident buffer, one workgroup barrier per K-step, fragment reads by ds_read_b128, six MFMAs per
//                    fragment pair -- for the tile / wave-count / MFMA-shape combinations under consideration
//   copy_*           HBM copy ceilings (grid-stride, one float4 per thread, non-temporal)
// Results feed DESIGN.md; nothing here computes a convolution.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <chrono>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------------------------------------ bare loops
template <int SHAPE>   // 32: 32x32x16 (4 accumulators of 16), 16: 16x16x32 (16 accumulators of 4): same output tile per wave
__global__ __launch_bounds__(512) void bare_kernel(float* out, int iters, const bf16x8* __restrict__ data) {
  bf16x8 a[4], b[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { a[u] = data[(u * 256 + threadIdx.x) & 4095]; b[u] = data[(1024 + u * 256 + threadIdx.x + blockIdx.x) & 4095]; }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[i], acc[i], 0, 0, 0);
      asm volatile("" : "+v"(a[0]));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)           // 2 x 16 MFMAs of 16384 FLOP = the 16 x 32768 FLOP of the other loop
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u * 2 + (i >> 3)) & 3], b[i & 3], acc[i], 0, 0, 0);
      asm volatile("" : "+v"(a[0]));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ------------------------------------------------------------------------------------------------ ring skeleton
// A workgroup of NW waves owns BM pixels x BN output channels.  Per K-step of 32 channels (one tap): the weights of the step,
// BN rows x 3 planes x 32 k x bf16 = BN * 192 B in fragment-major order, arrive in a ring stage by LDS-DMA (1-KiB pieces, dealt
// round-robin to the waves), two steps in flight behind a counted vmcnt; the pixel image [3 planes][4 k-groups][NPIX][16 B]
// stays put (REFRESH = 1: every 9 steps it is rewritten from global fp32 data through registers and the operand split, between
// two barriers, as a chunk change of the real kernel would).  SHAPE 16: weights are the A operand (rows = output channels),
// pixels the B operand; SHAPE 32: 32x32x16 fragments, two K halves per step.
struct RingArgs {
  const unsigned char* w;    // NSRC steps x BN*192 B (L2-resident, shared by all workgroups)
  const float* px;           // fp32 pixels for the refresh (large: HBM)
  long long px_floats;
  const unsigned char* pinit; // random bf16 bytes for the initial image
  float* out;
  int nsteps;                // K-steps per workgroup
  int nsrc;                  // steps in w
};

template <int SHAPE, int NW, int BM, int BN, int WAVES_M, int WAVES_N, bool REFRESH, bool BARRIER, bool DMA>
__global__ __launch_bounds__(NW * 64, (NW == 8) ? 1 : 2) void ring_kernel(const RingArgs a) {
  constexpr int TW = 32, TH = BM / TW;
  constexpr int HW = TW + 2, HR = (TH + 2) * HW;
  constexpr int NPIX = (HR + 15) / 16 * 16;
  constexpr int PLANE = 4 * NPIX * 16;                   // bytes of one plane of the image
  constexpr int IMG = 3 * PLANE;
  constexpr int BSTG = BN * 192;
  constexpr int BPIECES = BSTG / 1024;
  constexpr int B_PC = (BPIECES + NW - 1) / NW;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  static_assert(WAVES_M * WAVES_N == NW, "wave layout");
  static_assert(BPIECES % NW == 0, "uniform pieces per wave");

  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  unsigned char* const img = lds;
  unsigned char* const ring = lds + (IMG + 1023) / 1024 * 1024;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // initial image: random bf16
  for (int i = tid; i < IMG / 16; i += NW * 64) *(uint4*)(img + i * 16) = ((const uint4*)a.pinit)[(i + blockIdx.x * 7) & 16383];
  __syncthreads();

  const auto w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.w), 0, a.nsrc * BSTG, 0x00020000);
  auto issue_b = [&](int s) {
    unsigned char* st = ring + (s % 3) * BSTG;
    const unsigned so = (unsigned)(s % a.nsrc) * (unsigned)BSTG;
#pragma unroll
    for (int j = 0; j < B_PC; ++j) {
      const int g = wave + NW * j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_void*)(st + g * 1024), 16, (unsigned)lane * 16u, so + g * 1024, 0, 0);
    }
  };

  const int l15 = lane & 15, kg = lane >> 4, l31 = lane & 31, half = lane >> 5;
  float total = 0.f;

  if constexpr (SHAPE == 16) {
    constexpr int MI = WTM / 16, NI = WTN / 16;           // pixel fragments, channel fragments per wave
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int pb[MI];                                          // byte offset of this lane's pixel row in a plane, tap (0,0)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int p = wm * WTM + i * 16 + l15;             // pixel of the tile
      const int hp = ((p >> 5) + 1) * HW + (p & 31) + 1;
      pb[i] = (kg * NPIX + hp) * 16;
    }
    if (DMA) { issue_b(0); issue_b(1); }
    for (int s = 0; s < a.nsteps; ++s) {
      const int t = s % 9;
      if (DMA) wait_vmcnt<B_PC>();
      wait_lgkm0();
      if (BARRIER) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (REFRESH && t == 0 && s) {
        // chunk change: 32 channels of every halo pixel from global fp32 -> split -> image (the loads would be issued steps earlier
        // in the real kernel; here they are issued and consumed in place, which over-states their cost)
        constexpr int ITEMS = HR * 4;                    // (pixel, k-group) pairs, 32 B of fp32 each
        for (int it = tid; it < ITEMS; it += NW * 64) {
          const int px = it % HR, g = it / HR;
          const long long src = ((long long)(blockIdx.x * 977 + s * 131 + px) * 128 + g * 8) % (a.px_floats - 8);
          const f32x4 u = *(const f32x4*)(a.px + (src & ~3ll)), v = *(const f32x4*)(a.px + (src & ~3ll) + 4);
          bf16x8 p1, p2, p3;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = e < 4 ? u[e] : v[e - 4];
            const __bf16 h = (__bf16)x; const float r = x - (float)h;
            const __bf16 m = (__bf16)r; const float r2 = r - (float)m;
            p1[e] = h; p2[e] = m; p3[e] = (__bf16)r2;
          }
          unsigned char* d = img + (g * NPIX + px) * 16;
          *(bf16x8*)(d) = p1; *(bf16x8*)(d + PLANE) = p2; *(bf16x8*)(d + 2 * PLANE) = p3;
        }
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (DMA) issue_b(s + 2);
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const int toff = (dy * HW + dx) * 16;
      const unsigned char* st = ring + (s % 3) * BSTG;
      bf16x8 p[MI][3], w[NI][3];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) p[i][q] = *(const bf16x8*)(img + q * PLANE + pb[i] + toff);
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) w[j][q] = *(const bf16x8*)(st + ((wn * NI + j) * 3 + q) * 1024 + lane * 16);
#define TERM(QA, QB)                                                                              \
  _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                  \
  _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][QA], p[i][QB], acc[i][j], 0, 0, 0);
      TERM(2, 0) TERM(1, 1) TERM(0, 2) TERM(1, 0) TERM(0, 1) TERM(0, 0)
#undef TERM
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    constexpr int MI = WTM / 32, NI = WTN / 32;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int pb[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int p = wm * WTM + i * 32 + l31;
      const int hp = ((p >> 5) + 1) * HW + (p & 31) + 1;
      pb[i] = hp * 16;                                   // + (2 * kh + half) * NPIX * 16 per K half kh
    }
    if (DMA) { issue_b(0); issue_b(1); }
    for (int s = 0; s < a.nsteps; ++s) {
      const int t = s % 9;
      if (DMA) wait_vmcnt<B_PC>();
      wait_lgkm0();
      if (BARRIER) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (DMA) issue_b(s + 2);
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const int toff = (dy * HW + dx) * 16;
      const unsigned char* st = ring + (s % 3) * BSTG;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        bf16x8 p[MI][3], w[NI][3];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int q = 0; q < 3; ++q) p[i][q] = *(const bf16x8*)(img + q * PLANE + (2 * kh + half) * NPIX * 16 + pb[i] + toff);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q) w[j][q] = *(const bf16x8*)(st + (((wn * NI + j) * 2 + kh) * 3 + q) * 1024 + lane * 16);
#define TERM(QA, QB)                                                                              \
  _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                  \
  _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p[i][QA], w[j][QB], acc[i][j], 0, 0, 0);
        TERM(2, 0) TERM(1, 1) TERM(0, 2) TERM(1, 0) TERM(0, 1) TERM(0, 0)
#undef TERM
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) total += acc[i][j][r];
  }
  wait_vmcnt<0>();
  a.out[blockIdx.x * NW * 64 + tid] = total;
}

// ------------------------------------------------------------------------------------------------ copies
__global__ __launch_bounds__(256) void copy_gs(const f32x4* __restrict__ s, f32x4* __restrict__ d, long long nq) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long long)gridDim.x * 256) d[i] = s[i];
}
__global__ __launch_bounds__(256) void copy_one(const f32x4* __restrict__ s, f32x4* __restrict__ d, long long nq) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < nq) d[i] = s[i];
}
__global__ __launch_bounds__(256) void copy_one_nt(const f32x4* __restrict__ s, f32x4* __restrict__ d, long long nq) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < nq) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
__global__ __launch_bounds__(256) void copy_four(const f32x4* __restrict__ s, f32x4* __restrict__ d, long long nq) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) ;
  const long long st = (long long)gridDim.x * 256;
  f32x4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) if (i + k * st < nq) v[k] = s[i + k * st];
#pragma unroll
  for (int k = 0; k < 4; ++k) if (i + k * st < nq) d[i + k * st] = v[k];
}

// each block owns a CONTIGUOUS chunk (blocked partition), threads stride 256 inside it; U loads in flight per thread
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_chunk(const f32x4* __restrict__ s, f32x4* __restrict__ d, long long nq) {
  const long long per = (nq + gridDim.x - 1) / gridDim.x;
  const long long b0 = (long long)blockIdx.x * per;
  long long b1 = b0 + per; if (b1 > nq) b1 = nq;
  for (long long i = b0 + threadIdx.x; i < b1; i += 256 * U) {
    f32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) if (i + 256 * k < b1) v[k] = NT ? __builtin_nontemporal_load(s + i + 256 * k) : s[i + 256 * k];
#pragma unroll
    for (int k = 0; k < U; ++k) if (i + 256 * k < b1) { if (NT) __builtin_nontemporal_store(v[k], d + i + 256 * k); else d[i + 256 * k] = v[k]; }
  }
}

// ------------------------------------------------------------------------------------------------ host
template <class F>
static double time_ms(F&& launch, double seconds) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize());
  // keep the chip busy for `seconds` so the clock settles, and time the last third
  auto t0 = std::chrono::steady_clock::now();
  int n = 0; double ms1 = 0;
  while (true) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 4; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms1 = ms / 4; ++n;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) break;
  }
  // final measurement: 3 batches, take the median
  double v[3];
  for (int k = 0; k < 3; ++k) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 4; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v[k] = ms / 4;
  }
  (void)ms1;
  double a = v[0], b = v[1], c = v[2];
  double med = a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return med;
}

template <int SHAPE, int NW, int BM, int BN, int WM, int WN, bool REFRESH, bool BARRIER, bool DMA>
static void run_ring(const char* name, RingArgs a, int wgs, double secs) {
  constexpr int HR = (BM / 32 + 2) * 34, NPIX = (HR + 15) / 16 * 16;
  constexpr int lds_bytes = (3 * 4 * NPIX * 16 + 1023) / 1024 * 1024 + 3 * BN * 192;
  auto k = ring_kernel<SHAPE, NW, BM, BN, WM, WN, REFRESH, BARRIER, DMA>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  const double ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(wgs), dim3(NW * 64), lds_bytes, 0, a); }, secs);
  CK(hipGetLastError());
  const double flop = (double)wgs * a.nsteps * 6.0 * 2.0 * BM * BN * 32;
  printf("%-44s LDS %6d B  %4d WGs x %d thr  %8.3f ms  %7.1f executed bf16 TFLOP/s  (= %6.1f algorithmic)\n", name, lds_bytes, wgs, NW * 64, ms,
         flop / ms / 1e9, flop / 6 / ms / 1e9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 1.5;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  // random bf16 data (values ~ N(0,1) rounded) and random fp32
  std::vector<uint16_t> hb(4096 * 8 * 8);
  std::vector<float> hf(1 << 24);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (auto& v : hf) { const double u = (double)(rnd() >> 11) / 9007199254740992.0, w = (double)(rnd() >> 11) / 9007199254740992.0; v = (float)((u + w - 1.0) * 2.45); }
  for (size_t i = 0; i < hb.size(); ++i) { uint32_t u; float f = hf[i]; memcpy(&u, &f, 4); hb[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
  unsigned char *dbf, *dw; float *dpx, *dout;
  const int NSRC = 36;
  CK(hipMalloc(&dbf, hb.size() * 2)); CK(hipMemcpy(dbf, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&dw, NSRC * 256 * 192));
  for (int i = 0; i < NSRC * 256 * 192; i += (int)hb.size() * 2) CK(hipMemcpy(dw + i, hb.data(), std::min<size_t>(hb.size() * 2, NSRC * 256 * 192 - i), hipMemcpyHostToDevice));
  const long long PXF = 1ll << 28;                       // 1 GiB of fp32 pixels
  CK(hipMalloc(&dpx, PXF * 4));
  for (long long i = 0; i < PXF; i += (long long)hf.size()) CK(hipMemcpy(dpx + i, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dout, 4096 * 512 * 4));

  // ---- bare loops
  for (int thr : {256, 512}) {
    for (int shape : {32, 16}) {
      const int iters = 20000, wgs = 256;
      const double ms = time_ms([&] {
        if (shape == 32) hipLaunchKernelGGL(bare_kernel<32>, dim3(wgs), dim3(thr), 0, 0, dout, iters, (const bf16x8*)dbf);
        else hipLaunchKernelGGL(bare_kernel<16>, dim3(wgs), dim3(thr), 0, 0, dout, iters, (const bf16x8*)dbf);
      }, secs);
      printf("bare %s, random operands, %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s\n", shape == 32 ? "32x32x16" : "16x16x32", thr / 256, ms,
             (double)wgs * (thr / 64) * iters * 16 * 32768.0 / ms / 1e9);
      fflush(stdout);
    }
  }

  // bench.py's probe geometry: 768 workgroups of 256 threads (3 waves per SIMD), long launches
  for (int shape : {32, 16}) {
    const int iters = 200000, wgs = 768;
    const double ms = time_ms([&] {
      if (shape == 32) hipLaunchKernelGGL(bare_kernel<32>, dim3(wgs), dim3(256), 0, 0, dout, iters, (const bf16x8*)dbf);
      else hipLaunchKernelGGL(bare_kernel<16>, dim3(wgs), dim3(256), 0, 0, dout, iters, (const bf16x8*)dbf);
    }, 0.1);
    printf("bare %s, random operands, 768 x 256 threads x %d iterations: %8.3f ms  %7.1f TFLOP/s\n", shape == 32 ? "32x32x16" : "16x16x32", iters, ms,
           (double)wgs * 4 * iters * 16 * 32768.0 / ms / 1e9);
    fflush(stdout);
  }
  if (argc > 2 && !strcmp(argv[2], "bare")) return 0;

  // ---- ring skeletons
  const bool only_copy = argc > 2 && !strcmp(argv[2], "copy");
  RingArgs a{dw, dpx, PXF, dbf, dout, 0, NSRC};
  if (!only_copy) {
  a.nsteps = 36 * 8;                                      // 8 tiles' worth of a Cin = 128 layer per workgroup
  //                 SHAPE NW  BM   BN  WM WN refresh barrier dma
  run_ring<32, 4, 128,  64, 4, 1, false, true, true>("A  256thr 128x64 32x32x16 4x1", a, 512, secs);
  run_ring<16, 4, 128,  64, 4, 1, false, true, true>("B  256thr 128x64 16x16x32 4x1", a, 512, secs);
  run_ring<16, 4, 128,  64, 2, 2, false, true, true>("B2 256thr 128x64 16x16x32 2x2", a, 512, secs);
  run_ring<16, 8, 256, 128, 4, 2, false, true, true>("C  512thr 256x128 16x16x32 4x2", a, 256, secs);
  run_ring<16, 8, 256, 128, 8, 1, false, true, true>("C1 512thr 256x128 16x16x32 8x1", a, 256, secs);
  run_ring<32, 8, 256, 128, 4, 2, false, true, true>("C3 512thr 256x128 32x32x16 4x2", a, 256, secs);
  run_ring<16, 8, 256, 128, 4, 2, true,  true, true>("Cr 512thr 256x128 16x16x32 4x2 +refresh", a, 256, secs);
  run_ring<16, 8, 256, 128, 4, 2, false, false, true>("Cnb  ... no barrier (timing only)", a, 256, secs);
  run_ring<16, 8, 256, 128, 4, 2, false, true, false>("Cnd  ... no weight DMA (timing only)", a, 256, secs);
  run_ring<16, 8, 256, 128, 4, 2, false, false, false>("Cnn  ... neither (LDS reads + MFMA only)", a, 256, secs);
  run_ring<16, 4, 128,  64, 4, 1, true,  true, true>("Br 256thr 128x64 16x16x32 4x1 +refresh", a, 512, secs);

  }
  // ---- copies (1 GiB -> reuse dpx as source, second buffer as destination)
  float* dd; CK(hipMalloc(&dd, PXF * 4));
  const long long nq = PXF / 4;
  struct { const char* n; int kind; } cps[] = {{"copy grid-stride 4096 WGs", 0}, {"copy one float4 per thread", 1}, {"copy one float4 per thread, nt", 2}, {"copy 4 float4 per thread", 3},
                                               {"copy contiguous chunks, 4096 WGs, 1 in flight", 4}, {"copy contiguous chunks, 4096 WGs, 2 in flight", 5},
                                               {"copy contiguous chunks, 4096 WGs, 4 in flight", 6}, {"copy contiguous chunks, 4096 WGs, 4 in flight, nt", 7},
                                               {"copy contiguous chunks, 16384 WGs, 2 in flight", 8}, {"copy contiguous chunks, 65536 WGs, 1 in flight", 9}};
  for (auto& c : cps) {
    const double ms = time_ms([&] {
      if (c.kind == 0) hipLaunchKernelGGL(copy_gs, dim3(4096), dim3(256), 0, 0, (const f32x4*)dpx, (f32x4*)dd, nq);
      else if (c.kind == 1) hipLaunchKernelGGL(copy_one, dim3((unsigned)(nq / 256)), dim3(256), 0, 0, (const f32x4*)dpx, (f32x4*)dd, nq);
      else if (c.kind == 2) hipLaunchKernelGGL(copy_one_nt, dim3((unsigned)(nq / 256)), dim3(256), 0, 0, (const f32x4*)dpx, (f32x4*)dd, nq);
      else if (c.kind == 3) hipLaunchKernelGGL(copy_four, dim3((unsigned)(nq / 1024)), dim3(256), 0, 0, (const f32x4*)dpx, (f32x4*)dd, nq);
      else if (c.kind == 4) hipLaunchKernelGGL((copy_chunk<1, false>), dim3(4096), dim3(256), 0, 0, (const f32x4*)dpx, (f32x4*)dd, nq);
      else if (c.kind == 5) hipLaunchKernelGGL((copy_chunk<2, false>), dim3(4096), dim3(256), 0, 0, (const f32x4*)dpx, (f32x4*)dd, nq);
      else if (c.kind == 6) hipLaunchKernelGGL((copy_chunk<4, false>), dim3(4096), dim3(256), 0, 0, (const f32x4*)dpx, (f32x4*)dd, nq);
      else if (c.kind == 7) hipLaunchKernelGGL((copy_chunk<4, true>), dim3(4096), dim3(256), 0, 0, (const f32x4*)dpx, (f32x4*)dd, nq);
      else if (c.kind == 8) hipLaunchKernelGGL((copy_chunk<2, false>), dim3(16384), dim3(256), 0, 0, (const f32x4*)dpx, (f32x4*)dd, nq);
      else hipLaunchKernelGGL((copy_chunk<1, false>), dim3(65536), dim3(256), 0, 0, (const f32x4*)dpx, (f32x4*)dd, nq);
    }, 0.5);
    printf("%-36s %8.3f ms  %6.2f TB/s (read + write)\n", c.n, ms, 2.0 * PXF * 4 / ms / 1e9);
    fflush(stdout);
  }
  return 0;
}
