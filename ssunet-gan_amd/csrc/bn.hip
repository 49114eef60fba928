// Training-mode batch norm over NHWC rows [P pixels][C channels] for gfx950 -- HBM-bound kernels.
// ABI + reference citations: include/ssunet_hip.h (ssg_bn_*, ssg_channel_sum_f32).
//
// Layout choice: channels are the contiguous axis, so a wave's 64 lanes read TQ channel
// quads x (64/TQ) consecutive pixels = whole 256-B..1-KiB contiguous segments with 16-B loads.
// Per-channel reductions are "column" reductions: each thread keeps a float4 partial for its
// channel quad over a strided pixel set, the pixel rows of a block are combined through LDS
// in fp64, and every block writes one fp64 partial row; a second kernel adds the rows in
// block order.  No atomics: results are bitwise reproducible and ready to be all-reduced
// across ranks (sync-BN) between the two stages.
#include "common.h"

namespace {

constexpr int RED_BLOCK = 256;
constexpr int MAX_PARTS = 1024;

struct RedGeom { int TQ, PR, groups, parts; long long rows_per_part; };

// Q = channel quads per thread (1: fp32 and every 4-channel-granular caller; 2: bf16 with C % 8 == 0)
__host__ RedGeom red_geom(long long P, int C, int Q = 1) {
  RedGeom g;
  const int CQ = (C + 4 * Q - 1) / (4 * Q);
  g.TQ = CQ >= 64 ? 64 : 1;
  if (CQ < 64) { while (g.TQ < CQ) g.TQ <<= 1; }
  g.PR = RED_BLOCK / g.TQ;
  g.groups = (CQ + g.TQ - 1) / g.TQ;
  long long parts = (P + (long long)g.PR * 8 - 1) / ((long long)g.PR * 8);   // >= 8 rows per thread
  if (parts > MAX_PARTS) parts = MAX_PARTS;
  if (parts < 1) parts = 1;
  g.rows_per_part = (P + parts - 1) / parts;
  g.parts = (int)((P + g.rows_per_part - 1) / g.rows_per_part);
  return g;
}

// MODE 0: (sum x, sum x^2)      MODE 1: (sum g, sum g*xhat), g = dy * act'(y)      MODE 2: (sum x, -)
template <int MODE, typename T>
__global__ __launch_bounds__(RED_BLOCK) void col_reduce_kernel(
    const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy, long long P, int C,
    int ldx, int ldy, int lddy, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, const float* __restrict__ shift, int act, float slope,
    int TQ, int PR, long long rows_per_part, double* __restrict__ part /* [parts][2][Cpad] */) {
  constexpr int Q = SsgQ<T>::value;              // channel quads per 16-byte access
  constexpr int V = 4 * Q;
  __shared__ double red[2][RED_BLOCK][V];
  const int tid = threadIdx.x;
  const int tq = tid % TQ, pr = tid / TQ;
  const int cg = blockIdx.y * TQ + tq;           // channel group of V channels
  const int CG = (C + V - 1) / V;
  const bool cok = cg < CG;
  const long long p0 = (long long)blockIdx.x * rows_per_part;
  long long p1 = p0 + rows_per_part; if (p1 > P) p1 = P;
  // fp64 accumulators: var = E[x^2] - mean^2 cancels catastrophically in fp32 for channels whose
  // variance is far below mean^2 (deep layers with few pixels); x*x is exact in fp64.
  typedef typename SsgAcc<T>::type acc_t;
  acc_t s1[V], s2[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { s1[e] = 0; s2[e] = 0; }
  float mu[V], is[V], sc[V], sh[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { mu[e] = 0.f; is[e] = 0.f; sc[e] = 0.f; sh[e] = 0.f; }
  if (MODE == 1 && cok) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int c = V * cg + e;
      if (c < C) { mu[e] = mean[c]; is[e] = invstd[c]; if ((!y || act == SSG_ACT_SWISH) && scale) { sc[e] = scale[c]; sh[e] = shift[c]; } }
    }
  }
  if (cok) {
    for (long long p = p0 + pr; p < p1; p += PR) {
      f32x4 xq[Q];
      ldq(x + p * ldx + V * cg, xq);
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const acc_t v = (acc_t)xq[q][e]; s1[4 * q + e] += v; s2[4 * q + e] += v * v; }
      } else if (MODE == 2) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) s1[4 * q + e] += (acc_t)xq[q][e];
      } else {
        f32x4 gq[Q], yq[Q];
        ldq(dy + p * lddy + V * cg, gq);
        const bool use_y = act != SSG_ACT_NONE && act != SSG_ACT_SWISH && y;
        if (use_y) ldq(y + p * ldy + V * cg, yq);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = 4 * q + e;
            float g = gq[q][e];
            const float xv = xq[q][e];
            if (act == SSG_ACT_SWISH) {
              // smooth activation: its derivative is a function of the pre-activation z = x*scale + shift (recomputed)
              g *= ssg_swish_grad(xv * sc[k] + sh[k]);
            } else if (act != SSG_ACT_NONE) {
              // activation mask: from the saved output, or -- when the forward had no residual -- recomputed with
              // the forward's own expression x*scale + shift (same fp32 fma, same inputs: bitwise the same sign)
              const float yv = use_y ? yq[q][e] : xv * sc[k] + sh[k];
              if (!(yv > 0.f)) g *= (act == SSG_ACT_RELU ? 0.f : slope);
            }
            const acc_t xh = ((acc_t)xv - (acc_t)mu[k]) * (acc_t)is[k];
            s1[k] += (acc_t)g; s2[k] += (acc_t)g * xh;
          }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) { red[0][tid][e] = (double)s1[e]; red[1][tid][e] = (double)s2[e]; }
  __syncthreads();
  if (pr == 0 && cok) {
    double a1[V], a2[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { a1[e] = 0; a2[e] = 0; }
    for (int r = 0; r < PR; ++r)
#pragma unroll
      for (int e = 0; e < V; ++e) { a1[e] += red[0][r * TQ + tq][e]; a2[e] += red[1][r * TQ + tq][e]; }
    const int Cpad = V * CG;
    double* dst = part + (size_t)blockIdx.x * 2 * Cpad;
#pragma unroll
    for (int e = 0; e < V; ++e) { dst[V * cg + e] = a1[e]; dst[Cpad + V * cg + e] = a2[e]; }
  }
}

// SPADE modulate backward fused with the bias gradients of the gamma / beta convs (normalization.py:116-121):
//   dx = dy*(1+gamma), dgamma = dy*x, dbeta = dy, and per channel sum(dgamma) | sum(dbeta) -- the column sums
// used to re-read the freshly written 2C-channel tensor; here they ride the producing pass (same block geometry and
// fp64 partials as col_reduce_kernel, finished by col_reduce_final_kernel).
__global__ __launch_bounds__(RED_BLOCK) void modulate_bwd_sums_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ gb, int ldgb, const float* __restrict__ dy, int lddy,
    long long P, int C, float* __restrict__ dx, int lddx, float* __restrict__ dgb, int lddgb,
    int TQ, int PR, long long rows_per_part, double* __restrict__ part /* [parts][2][C] */) {
  __shared__ double red[2][RED_BLOCK][4];
  const int tid = threadIdx.x;
  const int tq = tid % TQ, pr = tid / TQ;
  const int cq = blockIdx.y * TQ + tq;
  const int CQ = C / 4;
  const bool cok = cq < CQ;
  const long long p0 = (long long)blockIdx.x * rows_per_part;
  long long p1 = p0 + rows_per_part; if (p1 > P) p1 = P;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (cok) {
    for (long long p = p0 + pr; p < p1; p += PR) {
      const f32x4 xv = *(const f32x4*)(x + p * ldx + 4 * cq);
      const f32x4 g = *(const f32x4*)(gb + p * ldgb + 4 * cq);
      const f32x4 d = *(const f32x4*)(dy + p * lddy + 4 * cq);
      const f32x4 dg = d * xv;
      st4(dx + p * lddx + 4 * cq, d * (1.f + g));
      st4(dgb + p * lddgb + 4 * cq, dg);
      st4(dgb + p * lddgb + C + 4 * cq, d);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += (double)dg[e]; s2[e] += (double)d[e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][tid][e] = s1[e]; red[1][tid][e] = s2[e]; }
  __syncthreads();
  if (pr == 0 && cok) {
    double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    for (int r = 0; r < PR; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) { a1[e] += red[0][r * TQ + tq][e]; a2[e] += red[1][r * TQ + tq][e]; }
    double* dst = part + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < 4; ++e) { dst[4 * cq + e] = a1[e]; dst[C + 4 * cq + e] = a2[e]; }
  }
}

// Second stage: 32 channels x 32 part-lanes per block; each lane adds every 32nd partial row in
// row order, then the 32 lane sums are added in lane order -> fixed summation order (bitwise
// reproducible), 1/32 of the serial chain of a one-thread-per-channel loop.
constexpr int FIN_LANES = 32;

// per-channel batch-norm constants from the fp64 sums (shared by bn_finalize_kernel and the fused second reduce stage, so that
// the two routes give the same bits)
struct BnFin {
  const float* weight; const float* bias; float eps, momentum; int var_mode;
  float* running_mean; float* running_var; float* mean; float* invstd; float* scale; float* shift;
};

__device__ __forceinline__ void bn_finalize_channel(int c, double s1, double s2, double count, const BnFin& f) {
  // Every multiply-add below is an EXPLICIT fma: hipcc contracts a*b + c on its own (-ffp-contract=fast, and the backend
  // does it per inlining context whatever the source pragmas say), so the two kernels that share this function gave
  // running_var values one ulp apart until the fusion was spelled out (tests/test_round2_gpu.py compares a synchronised
  // world-1 run, which finalizes in its own launch, with the local fused route bit for bit).
  const double m = s1 / count;
  double var = fma(-m, m, s2 / count);
  if (var < 0) var = 0;
  double is;
  if (f.var_mode == 0) is = 1.0 / sqrt(var + (double)f.eps);
  else is = 1.0 / sqrt(var < (double)f.eps ? (double)f.eps : var);
  const float mf = (float)m, isf = (float)is;
  f.mean[c] = mf; f.invstd[c] = isf;
  const float w = f.weight ? f.weight[c] : 1.f, b = f.bias ? f.bias[c] : 0.f;
  const float sc = w * isf;
  f.scale[c] = sc; f.shift[c] = fmaf(-mf, sc, b);
  if (f.running_mean) f.running_mean[c] = fmaf(f.momentum, mf, (1.f - f.momentum) * f.running_mean[c]);
  if (f.running_var) {
    const double unb = count > 1 ? var * count / (count - 1) : var;
    f.running_var[c] = fmaf(f.momentum, (float)unb, (1.f - f.momentum) * f.running_var[c]);
  }
}

// `fin.mean != nullptr`: a local (un-synchronised) batch norm finishes here -- mean / invstd / scale / shift and the running
// estimates from the channel's two totals, one launch instead of final + finalize (`fin_count` = pixels).
__global__ __launch_bounds__(32 * FIN_LANES) void col_reduce_final_kernel(const double* __restrict__ part, int parts, int C, int C4,
                                                                          double* __restrict__ sums, float* __restrict__ fsum,
                                                                          double count_out, BnFin fin, double fin_count) {
  __shared__ double red[2][FIN_LANES][32];
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a1 = 0, a2 = 0;
  if (c < C) {
#pragma unroll 4
    for (int b = lane; b < parts; b += FIN_LANES) { a1 += part[(size_t)b * 2 * C4 + c]; a2 += part[(size_t)b * 2 * C4 + C4 + c]; }
  }
  red[0][lane][cl] = a1; red[1][lane][cl] = a2;
  __syncthreads();
  if (lane == 0 && c < C) {
    double t1 = 0, t2 = 0;
#pragma unroll
    for (int k = 0; k < FIN_LANES; ++k) { t1 += red[0][k][cl]; t2 += red[1][k][cl]; }
    if (sums) { sums[c] = t1; sums[C + c] = t2; }
    if (fsum) fsum[c] = (float)t1;
    if (fin.mean) bn_finalize_channel(c, t1, t2, fin_count, fin);
  }
  // sync-BN: this rank's pixel count rides the vector that is all-reduced (ranks may hold different local batches)
  if (sums && count_out > 0 && blockIdx.x == 0 && threadIdx.x == 0) sums[2 * C] = count_out;
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count_arg, int C, BnFin fin) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double count = count_arg > 0 ? count_arg : sums[2 * C];
  bn_finalize_channel(c, sums[c], sums[C + c], count, fin);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, long long P, int C, int ld,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const T* __restrict__ res, int ldr, int act, float slope,
                                                       T* __restrict__ y, int ldy) {
  constexpr int Q = SsgQ<T>::value;          // channel quads per 16-byte access
  const int CG = C / (4 * Q);
  const long long total = P * CG;
  SSG_CHUNK_LOOP(i, total) {
    const long long p = i / CG; const int c0 = 4 * Q * (int)(i - p * CG);
    f32x4 xv[Q], rv[Q], o[Q];
    ldq(x + p * ld + c0, xv);
    if (res) ldq(res + p * ldr + c0, rv);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const f32x4 sc = *(const f32x4*)(scale + c0 + 4 * q), sh = *(const f32x4*)(shift + c0 + 4 * q);
      f32x4 v = xv[q] * sc + sh;
      if (res) v += rv[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ssg_act(v[e], act, slope);
      o[q] = v;
    }
    stq(y + p * ldy + c0, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy, long long P, int C, int ldx,
    int ldy, int lddy, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ weight,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const double* __restrict__ sums, double count_arg, int act, float slope, T* __restrict__ dx, int lddx,
    T* __restrict__ dres, int lddres, float* __restrict__ dweight, float* __restrict__ dbias) {
  const double count = count_arg > 0 ? count_arg : sums[2 * C];
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      if (dweight) dweight[c] = (float)sums[C + c];
      if (dbias) dbias[c] = (float)sums[c];
    }
  }
  // per-channel constants once per block, evaluated in fp64 (the divisions are not repeated per element) and kept in the
  // accumulation type of the tensor family: fp64 for fp32 tensors, fp32 for bf16 (common.h)
  typedef typename SsgAcc<T>::type acc_t;
  extern __shared__ double kc_raw[];
  acc_t* kc = (acc_t*)kc_raw;                    // [5][C]: mean, invstd, w*invstd, m1, m2
  acc_t* k_mean = kc; acc_t* k_is = kc + C; acc_t* k_ws = kc + 2 * C; acc_t* k_m1 = kc + 3 * C; acc_t* k_m2 = kc + 4 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const double is = (double)invstd[c];
    k_mean[c] = (acc_t)mean[c]; k_is[c] = (acc_t)is; k_ws[c] = (acc_t)((weight ? (double)weight[c] : 1.0) * is);
    k_m1[c] = (acc_t)(sums[c] / count); k_m2[c] = (acc_t)(sums[C + c] / count);
  }
  __syncthreads();
  constexpr int Q = SsgQ<T>::value;
  const int CG = C / (4 * Q);
  const long long totalg = P * CG;
  SSG_CHUNK_LOOP(i, totalg) {
    const long long p = i / CG; const int c0 = 4 * Q * (int)(i - p * CG);
    f32x4 gq[Q], xq[Q], yq[Q], oq[Q];
    ldq(dy + p * lddy + c0, gq);
    ldq(x + p * ldx + c0, xq);
    const bool use_y = act != SSG_ACT_NONE && act != SSG_ACT_SWISH && y;
    if (use_y) ldq(y + p * ldy + c0, yq);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int cb = c0 + 4 * q;
      f32x4 g = gq[q];
      const f32x4 xv = xq[q];
      if (act == SSG_ACT_SWISH) {
        const f32x4 z = xv * *(const f32x4*)(scale + cb) + *(const f32x4*)(shift + cb);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] *= ssg_swish_grad(z[e]);
      } else if (act != SSG_ACT_NONE) {
        const f32x4 yv = use_y ? yq[q] : xv * *(const f32x4*)(scale + cb) + *(const f32x4*)(shift + cb);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (!(yv[e] > 0.f)) g[e] *= (act == SSG_ACT_RELU ? 0.f : slope);
      }
      gq[q] = g;
      if (dx) {
        f32x4 o;
        // fp64 arithmetic, as ATen's CPU batch-norm backward (accscalar = double for float tensors):
        // g - mean(g) - xhat*mean(g*xhat) cancels heavily, and an fp32-rounded per-channel constant
        // would add the SAME error to every pixel (a coherent bias that the next dgrad/bias-grad
        // sums amplify).  The kernel is HBM-bound; fp64 VALU work is free here.
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = cb + e;
          const acc_t xh = ((acc_t)xv[e] - k_mean[c]) * k_is[c];
          o[e] = (float)(k_ws[c] * ((acc_t)g[e] - k_m1[c] - xh * k_m2[c]));
        }
        oq[q] = o;
      }
    }
    if (dres) stq(dres + p * lddres + c0, gq);
    if (dx) stq(dx + p * lddx + c0, oq);
  }
}

int elem_grid(long long total) { return ssg_elem_grid(total, 4); }     // 4 items per thread: bn_bwd_apply has a per-block prologue (5 fp64 constants per channel)

template <int MODE, typename T>
int run_reduce(const T* x, const T* y, const T* dy, long long P, int C, int ldx, int ldy, int lddy,
               const float* scale, const float* shift,
               const float* mean, const float* invstd, int act, float slope, double* sums, float* fsum, void* ws,
               hipStream_t st, double count_out = 0.0, BnFin fin = BnFin{}, double fin_count = 0.0) {
  constexpr int Q = SsgQ<T>::value;
  const RedGeom g = red_geom(P, C, Q);
  const int C4 = 4 * Q * ((C + 4 * Q - 1) / (4 * Q));
  double* part = (double*)ws;
  hipLaunchKernelGGL((col_reduce_kernel<MODE, T>), dim3((unsigned)g.parts, (unsigned)g.groups), dim3(RED_BLOCK), 0, st, x, y, dy,
                     P, C, ldx, ldy, lddy, mean, invstd, scale, shift, act, slope, g.TQ, g.PR, g.rows_per_part, part);
  SSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(col_reduce_final_kernel, dim3((unsigned)((C + 31) / 32)), dim3(32 * FIN_LANES), 0, st, part, g.parts, C, C4, sums, fsum, count_out, fin, fin_count);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

}  // namespace

// One query serves the fp32 reducers (4 channels per thread) and the bf16 ones (8): the larger of the two geometries, each
// with its own channel padding -- not "Q = 1 happens to be the larger" (ADVICE r2).
extern "C" int64_t ssg_bn_workspace_bytes(int64_t P, int C) {
  const RedGeom g1 = red_geom(P, C, 1), g2 = red_geom(P, C, 2);
  const int64_t b1 = (int64_t)g1.parts * 2 * 4 * ((C + 3) / 4) * (int64_t)sizeof(double);
  const int64_t b2 = (int64_t)g2.parts * 2 * 8 * ((C + 7) / 8) * (int64_t)sizeof(double);
  return b1 > b2 ? b1 : b2;
}

namespace {

template <typename T>
int bn_stats_impl(const T* x, int64_t P, int C, int ld, double* sums, int with_count, void* ws, void* stream) {
  SSG_REQUIRE(x && sums && ws && P > 0 && C > 0, SSG_EINVAL, "bn_stats: bad args");
  SSG_REQUIRE(C % (4 * SsgQ<T>::value) == 0 && ld % (4 * SsgQ<T>::value) == 0 && ld >= C && ssg_aligned16(x), SSG_EALIGN,
              "bn_stats: C / ld must be multiples of 4 (fp32) or 8 (bf16), rows 16-byte aligned");
  return run_reduce<0, T>(x, nullptr, nullptr, P, C, ld, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.f, sums, nullptr, ws, (hipStream_t)stream,
                          with_count ? (double)P : 0.0);
}

template <typename T>
int bn_apply_impl(const T* x, int64_t P, int C, int ld, const float* scale, const float* shift, const T* res, int ldr, int act, float slope,
                  T* y, int ldy, void* stream) {
  SSG_REQUIRE(x && y && scale && shift && P > 0 && C > 0, SSG_EINVAL, "bn_apply: bad args");
  constexpr int G = 4 * SsgQ<T>::value;
  SSG_REQUIRE(C % G == 0 && ld % G == 0 && ldy % G == 0 && (!res || ldr % G == 0) && ssg_aligned16(x) && ssg_aligned16(y) && (!res || ssg_aligned16(res)),
              SSG_EALIGN, "bn_apply: C/ld must be multiples of 4 (fp32) or 8 (bf16), rows 16-byte aligned");
  SSG_REQUIRE(!(act == SSG_ACT_SWISH && res), SSG_EINVAL, "bn_apply: swish with a residual is not supported (its backward needs the pre-activation)");
  hipLaunchKernelGGL(bn_apply_kernel<T>, dim3((unsigned)elem_grid(P * (C / (4 * SsgQ<T>::value)))), dim3(256), 0, (hipStream_t)stream, x, (long long)P, C, ld,
                     scale, shift, res, ldr, act, slope, y, ldy);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

template <typename T>
int bn_bwd_reduce_impl(const T* x, const T* y, const T* dy, int64_t P, int C, int ldx, int ldy, int lddy, const float* mean,
                       const float* invstd, const float* scale, const float* shift, int act, float slope, double* sums, int with_count,
                       void* ws, void* stream) {
  SSG_REQUIRE(x && dy && mean && invstd && sums && ws && P > 0 && C > 0, SSG_EINVAL, "bn_bwd_reduce: bad args");
  SSG_REQUIRE(act == SSG_ACT_NONE || (y && act != SSG_ACT_SWISH) || (scale && shift), SSG_EINVAL,
              "bn_bwd_reduce: activation gradient needs y (ReLU family only) or (scale, shift)");
  SSG_REQUIRE(C % (4 * SsgQ<T>::value) == 0 && ldx % (4 * SsgQ<T>::value) == 0 && lddy % (4 * SsgQ<T>::value) == 0 && ssg_aligned16(x) && ssg_aligned16(dy),
              SSG_EALIGN, "bn_bwd_reduce: C / ld must be multiples of 4 (fp32) or 8 (bf16), rows 16-byte aligned");
  return run_reduce<1, T>(x, y, dy, P, C, ldx, ldy, lddy, scale, shift, mean, invstd, act, slope, sums, nullptr, ws, (hipStream_t)stream,
                          with_count ? (double)P : 0.0);
}

template <typename T>
int bn_bwd_apply_impl(const T* x, const T* y, const T* dy, int64_t P, int C, int ldx, int ldy, int lddy, const float* mean,
                      const float* invstd, const float* weight, const float* scale, const float* shift, const double* sums, double count,
                      int act, float slope, T* dx, int lddx, T* dres, int lddres, float* dweight, float* dbias, void* stream) {
  SSG_REQUIRE(x && dy && mean && invstd && sums && P > 0 && C > 0, SSG_EINVAL, "bn_bwd_apply: bad args");
  SSG_REQUIRE(act == SSG_ACT_NONE || (y && act != SSG_ACT_SWISH) || (scale && shift), SSG_EINVAL,
              "bn_bwd_apply: activation gradient needs y (ReLU family only) or (scale, shift)");
  constexpr int G = 4 * SsgQ<T>::value;
  SSG_REQUIRE(C % G == 0 && ldx % G == 0 && lddy % G == 0 && (!y || ldy % G == 0) && (!dx || lddx % G == 0) && (!dres || lddres % G == 0) &&
                  ssg_aligned16(x) && ssg_aligned16(dy) && (!dx || ssg_aligned16(dx)), SSG_EALIGN,
              "bn_bwd_apply: C/ld must be multiples of 4 (fp32) or 8 (bf16), rows 16-byte aligned");
  SSG_REQUIRE(C <= 4096, SSG_EINVAL, "bn_bwd_apply: C > 4096");
  if ((size_t)5 * C * sizeof(double) > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)bn_bwd_apply_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * C * (int)sizeof(double));
    if (e != hipSuccess) { ssg_set_error("bn_bwd_apply: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3((unsigned)elem_grid(P * (C / (4 * SsgQ<T>::value)))), dim3(256), (size_t)5 * C * sizeof(double), (hipStream_t)stream, x, y, dy,
                     (long long)P, C, ldx, ldy, lddy, mean, invstd, weight, scale, shift, sums, count, act, slope, dx, lddx, dres, lddres,
                     dweight, dbias);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

}  // namespace

extern "C" int ssg_bn_stats_f32(const float* x, int64_t P, int C, int ld, double* sums, int with_count, void* ws, void* stream) {
  return bn_stats_impl<float>(x, P, C, ld, sums, with_count, ws, stream);
}
extern "C" int ssg_bn_stats_bf16(const void* x, int64_t P, int C, int ld, double* sums, int with_count, void* ws, void* stream) {
  return bn_stats_impl<ssg_bf16>((const ssg_bf16*)x, P, C, ld, sums, with_count, ws, stream);
}

namespace {
// The conv epilogue leaves one row per M-tile (up to 32 768 rows for a 64-channel layer at 16 x 512^2): folding them with
// col_reduce_final_kernel alone would put C/32 workgroups on tens of MB.  First fold slices of rows in parallel (grid.y),
// 32 channels x 8 row-lanes per block, rows of a lane in order, lanes in order: a fixed summation order.
constexpr int FOLD_Z = 64;
__global__ __launch_bounds__(256) void bnpart_fold_kernel(const double* __restrict__ part, int rows, int C, int rows_per_z,
                                                          double* __restrict__ out /* [gridDim.y][2][C] */) {
  __shared__ double red[2][8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int r0 = blockIdx.y * rows_per_z;
  int r1 = r0 + rows_per_z; if (r1 > rows) r1 = rows;
  double a1 = 0, a2 = 0;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 8) { a1 += part[(size_t)r * 2 * C + c]; a2 += part[(size_t)r * 2 * C + C + c]; }
  red[0][rl][cl] = a1; red[1][rl][cl] = a2;
  __syncthreads();
  if (rl == 0 && c < C) {
    double t1 = 0, t2 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { t1 += red[0][k][cl]; t2 += red[1][k][cl]; }
    out[(size_t)blockIdx.y * 2 * C + c] = t1; out[(size_t)blockIdx.y * 2 * C + C + c] = t2;
  }
}
}  // namespace

extern "C" int64_t ssg_bn_stats_from_partials_workspace_bytes(int rows, int C) {
  return rows > 4 * FOLD_Z ? (int64_t)FOLD_Z * 2 * C * (int64_t)sizeof(double) : 16;
}

extern "C" int ssg_bn_stats_from_partials_f32(const double* part, int rows, int C, double* sums, double count, void* ws, void* stream) {
  SSG_REQUIRE(part && sums && ws && rows > 0 && C > 0, SSG_EINVAL, "bn_stats_from_partials: bad args");
  hipStream_t st = (hipStream_t)stream;
  const double* src = part; int nrows = rows;
  if (rows > 4 * FOLD_Z) {
    const int rpz = (rows + FOLD_Z - 1) / FOLD_Z;
    const int nz = (rows + rpz - 1) / rpz;
    hipLaunchKernelGGL(bnpart_fold_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)nz), dim3(256), 0, st, part, rows, C, rpz, (double*)ws);
    SSG_LAUNCH_CHECK();
    src = (const double*)ws; nrows = nz;
  }
  hipLaunchKernelGGL(col_reduce_final_kernel, dim3((unsigned)((C + 31) / 32)), dim3(32 * FIN_LANES), 0, st, src, nrows, C, C, sums, (float*)nullptr, count, BnFin{}, 0.0);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_channel_sum_f32(const float* x, int64_t P, int C, int ld, float* out, void* ws, void* stream) {
  SSG_REQUIRE(x && out && ws && P > 0 && C > 0, SSG_EINVAL, "channel_sum: bad args");
  SSG_REQUIRE(ld % 4 == 0 && ld >= C && ssg_aligned16(x), SSG_EALIGN, "channel_sum: alignment");
  return run_reduce<2, float>(x, nullptr, nullptr, P, C, ld, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.f, nullptr, out, ws, (hipStream_t)stream);
}

extern "C" int ssg_bn_finalize_f32(const double* sums, double count, int C, const float* weight, const float* bias,
                                   float eps, float momentum, int var_mode, float* running_mean, float* running_var,
                                   float* mean, float* invstd, float* scale, float* shift, void* stream) {
  SSG_REQUIRE(sums && mean && invstd && scale && shift && C > 0, SSG_EINVAL, "bn_finalize: bad args");
  const BnFin fin{weight, bias, eps, momentum, var_mode, running_mean, running_var, mean, invstd, scale, shift};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 127) / 128)), dim3(128), 0, (hipStream_t)stream, sums, count, C, fin);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

// Local batch norm: statistics and finalize in one call (the second reduce stage finishes the channel) -- one launch less per
// batch-norm forward than ssg_bn_stats_* + ssg_bn_finalize_f32, same bits.
namespace {
template <typename T>
int bn_stats_finalize_impl(const T* x, int64_t P, int C, int ld, const ssg_bn_fin* f, void* ws, void* stream) {
  SSG_REQUIRE(x && f && ws && P > 0 && C > 0 && f->mean && f->invstd && f->scale && f->shift, SSG_EINVAL, "bn_stats_finalize: bad args");
  SSG_REQUIRE(C % (4 * SsgQ<T>::value) == 0 && ld % (4 * SsgQ<T>::value) == 0 && ld >= C && ssg_aligned16(x), SSG_EALIGN,
              "bn_stats_finalize: C / ld must be multiples of 4 (fp32) or 8 (bf16), rows 16-byte aligned");
  const BnFin fin{f->weight, f->bias, f->eps, f->momentum, f->var_mode, f->running_mean, f->running_var, f->mean, f->invstd, f->scale, f->shift};
  return run_reduce<0, T>(x, nullptr, nullptr, P, C, ld, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.f, nullptr, nullptr, ws, (hipStream_t)stream,
                          0.0, fin, (double)P);
}
}  // namespace

extern "C" int ssg_bn_stats_finalize_f32(const float* x, int64_t P, int C, int ld, const ssg_bn_fin* fin, void* ws, void* stream) {
  return bn_stats_finalize_impl<float>(x, P, C, ld, fin, ws, stream);
}
extern "C" int ssg_bn_stats_finalize_bf16(const void* x, int64_t P, int C, int ld, const ssg_bn_fin* fin, void* ws, void* stream) {
  return bn_stats_finalize_impl<ssg_bf16>((const ssg_bf16*)x, P, C, ld, fin, ws, stream);
}

extern "C" int ssg_bn_stats_from_partials_finalize_f32(const double* part, int rows, int C, double count, const ssg_bn_fin* f, void* ws, void* stream) {
  SSG_REQUIRE(part && f && ws && rows > 0 && C > 0 && count > 0 && f->mean && f->invstd && f->scale && f->shift, SSG_EINVAL,
              "bn_stats_from_partials_finalize: bad args");
  hipStream_t st = (hipStream_t)stream;
  const BnFin fin{f->weight, f->bias, f->eps, f->momentum, f->var_mode, f->running_mean, f->running_var, f->mean, f->invstd, f->scale, f->shift};
  const double* src = part; int nrows = rows;
  if (rows > 4 * FOLD_Z) {
    const int rpz = (rows + FOLD_Z - 1) / FOLD_Z;
    const int nz = (rows + rpz - 1) / rpz;
    hipLaunchKernelGGL(bnpart_fold_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)nz), dim3(256), 0, st, part, rows, C, rpz, (double*)ws);
    SSG_LAUNCH_CHECK();
    src = (const double*)ws; nrows = nz;
  }
  hipLaunchKernelGGL(col_reduce_final_kernel, dim3((unsigned)((C + 31) / 32)), dim3(32 * FIN_LANES), 0, st, src, nrows, C, C, (double*)nullptr,
                     (float*)nullptr, 0.0, fin, count);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}

extern "C" int ssg_bn_apply_f32(const float* x, int64_t P, int C, int ld, const float* scale, const float* shift,
                                const float* res, int ldr, int act, float slope, float* y, int ldy, void* stream) {
  return bn_apply_impl<float>(x, P, C, ld, scale, shift, res, ldr, act, slope, y, ldy, stream);
}
extern "C" int ssg_bn_apply_bf16(const void* x, int64_t P, int C, int ld, const float* scale, const float* shift,
                                 const void* res, int ldr, int act, float slope, void* y, int ldy, void* stream) {
  return bn_apply_impl<ssg_bf16>((const ssg_bf16*)x, P, C, ld, scale, shift, (const ssg_bf16*)res, ldr, act, slope, (ssg_bf16*)y, ldy, stream);
}

extern "C" int ssg_bn_bwd_reduce_f32(const float* x, const float* y, const float* dy, int64_t P, int C, int ldx, int ldy,
                                     int lddy, const float* mean, const float* invstd, const float* scale, const float* shift,
                                     int act, float slope, double* sums, int with_count, void* ws, void* stream) {
  return bn_bwd_reduce_impl<float>(x, y, dy, P, C, ldx, ldy, lddy, mean, invstd, scale, shift, act, slope, sums, with_count, ws, stream);
}
extern "C" int ssg_bn_bwd_reduce_bf16(const void* x, const void* y, const void* dy, int64_t P, int C, int ldx, int ldy,
                                      int lddy, const float* mean, const float* invstd, const float* scale, const float* shift,
                                      int act, float slope, double* sums, int with_count, void* ws, void* stream) {
  return bn_bwd_reduce_impl<ssg_bf16>((const ssg_bf16*)x, (const ssg_bf16*)y, (const ssg_bf16*)dy, P, C, ldx, ldy, lddy, mean, invstd, scale, shift,
                                      act, slope, sums, with_count, ws, stream);
}

extern "C" int ssg_bn_bwd_apply_f32(const float* x, const float* y, const float* dy, int64_t P, int C, int ldx, int ldy,
                                    int lddy, const float* mean, const float* invstd, const float* weight,
                                    const float* scale, const float* shift, const double* sums,
                                    double count, int act, float slope, float* dx, int lddx, float* dres, int lddres,
                                    float* dweight, float* dbias, void* stream) {
  return bn_bwd_apply_impl<float>(x, y, dy, P, C, ldx, ldy, lddy, mean, invstd, weight, scale, shift, sums, count, act, slope, dx, lddx, dres,
                                  lddres, dweight, dbias, stream);
}
extern "C" int ssg_bn_bwd_apply_bf16(const void* x, const void* y, const void* dy, int64_t P, int C, int ldx, int ldy,
                                     int lddy, const float* mean, const float* invstd, const float* weight,
                                     const float* scale, const float* shift, const double* sums,
                                     double count, int act, float slope, void* dx, int lddx, void* dres, int lddres,
                                     float* dweight, float* dbias, void* stream) {
  return bn_bwd_apply_impl<ssg_bf16>((const ssg_bf16*)x, (const ssg_bf16*)y, (const ssg_bf16*)dy, P, C, ldx, ldy, lddy, mean, invstd, weight, scale,
                                     shift, sums, count, act, slope, (ssg_bf16*)dx, lddx, (ssg_bf16*)dres, lddres, dweight, dbias, stream);
}

extern "C" int ssg_spade_modulate_bwd_sums_f32(const float* x, int ldx, const float* gb, int ldgb, const float* dy, int lddy, int64_t P,
                                               int C, float* dx, int lddx, float* dgb, int lddgb, double* sums, void* ws, void* stream) {
  SSG_REQUIRE(x && gb && dy && dx && dgb && sums && ws && P > 0 && C > 0 && C % 4 == 0 && ldgb >= C && lddgb >= 2 * C, SSG_EINVAL,
              "modulate_bwd_sums: bad args");
  const RedGeom g = red_geom(P, C);
  double* part = (double*)ws;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(modulate_bwd_sums_kernel, dim3((unsigned)g.parts, (unsigned)g.groups), dim3(RED_BLOCK), 0, st, x, ldx, gb, ldgb, dy, lddy,
                     (long long)P, C, dx, lddx, dgb, lddgb, g.TQ, g.PR, g.rows_per_part, part);
  SSG_LAUNCH_CHECK();
  // sums[0:C] = sum dgamma, sums[C:2C] = sum dbeta (ssg_bn_workspace_bytes(P, C) bytes of workspace)
  hipLaunchKernelGGL(col_reduce_final_kernel, dim3((unsigned)((C + 31) / 32)), dim3(32 * FIN_LANES), 0, st, part, g.parts, C, C, sums, nullptr, 0.0, BnFin{}, 0.0);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
