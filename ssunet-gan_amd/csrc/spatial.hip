// Pool / unpool / upsample / adaptive-avgpool / layout kernels for gfx950 (HBM-bound).
// ABI + reference citations: include/ssunet_hip.h.  One thread handles one 16-byte channel
// quad of one pixel, so consecutive lanes read/write consecutive 16-B pieces of an NHWC row.
#include "common.h"

namespace {

int elem_grid(long long total) { return ssg_elem_grid(total, 2); }

#define GRID_STRIDE(i, total) SSG_CHUNK_LOOP(i, total)      // contiguous chunk per block (common.h)

// ---------------------------------------------------------------- max pool 2x2 (+argmax byte)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ldx,
                                                          float* __restrict__ y, int ldy, uint8_t* __restrict__ idx) {
  const int OH = H / 2, OW = W / 2, CQ = C / 4;
  const long long total = (long long)N * OH * OW * CQ;
  GRID_STRIDE(i, total) {
    const int cq = (int)(i % CQ); long long r = i / CQ;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH); const int n = (int)(r / OH);
    const float* b = x + ((size_t)(n * H + 2 * oy) * W + 2 * ox) * ldx + 4 * cq;
    f32x4 best = *(const f32x4*)b;
    int bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const f32x4 v = *(const f32x4*)(b + ((size_t)(k >> 1) * W + (k & 1)) * ldx);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = k; }   // first max wins, NaN wins (ATen CPU)
    }
    const size_t o = ((size_t)(n * OH + oy) * OW + ox);
    *(f32x4*)(y + o * ldy + 4 * cq) = best;
    *(uint32_t*)(idx + o * C + 4 * cq) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
  }
}

// scatter src[n,oy,ox,c] to the argmax position of its 2x2 window in dst (zeros elsewhere):
// = max-pool backward and = max-unpool forward.
// `res` (optional, laid out like dst) is added: the gradient of a tensor that feeds BOTH a max-pool and a skip connection
// (archs.py:628-667: every encoder output) is formed in this pass instead of by a separate 3-tensor add.
__global__ __launch_bounds__(256) void scatter2x2_kernel(const float* __restrict__ src, int lds_, const uint8_t* __restrict__ idx,
                                                         int N, int OH, int OW, int C, float* __restrict__ dst, int ldd,
                                                         const float* __restrict__ res, int ldr) {
  const int CQ = C / 4, H = OH * 2, W = OW * 2;
  const long long total = (long long)N * OH * OW * CQ;
  GRID_STRIDE(i, total) {
    const int cq = (int)(i % CQ); long long r = i / CQ;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH); const int n = (int)(r / OH);
    const size_t o = ((size_t)(n * OH + oy) * OW + ox);
    const f32x4 v = *(const f32x4*)(src + o * lds_ + 4 * cq);
    const uint32_t pk = *(const uint32_t*)(idx + o * C + 4 * cq);
    float* b = dst + ((size_t)(n * H + 2 * oy) * W + 2 * ox) * ldd + 4 * cq;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (((pk >> (8 * e)) & 255u) == (uint32_t)k) ? v[e] : 0.f;
      if (res) w += *(const f32x4*)(res + (((size_t)(n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)) * ldr + 4 * cq));
      *(f32x4*)(b + ((size_t)(k >> 1) * W + (k & 1)) * ldd) = w;
    }
  }
}

// gather dst[n,oy,ox,c] = src[window position idx]: = max-unpool backward.
__global__ __launch_bounds__(256) void gather2x2_kernel(const float* __restrict__ src, int lds_, const uint8_t* __restrict__ idx,
                                                        int N, int OH, int OW, int C, float* __restrict__ dst, int ldd) {
  const int CQ = C / 4, H = OH * 2, W = OW * 2;
  const long long total = (long long)N * OH * OW * CQ;
  GRID_STRIDE(i, total) {
    const int cq = (int)(i % CQ); long long r = i / CQ;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH); const int n = (int)(r / OH);
    const size_t o = ((size_t)(n * OH + oy) * OW + ox);
    const uint32_t pk = *(const uint32_t*)(idx + o * C + 4 * cq);
    const float* b = src + ((size_t)(n * H + 2 * oy) * W + 2 * ox) * lds_ + 4 * cq;
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = (pk >> (8 * e)) & 3;
      out[e] = b[((size_t)(k >> 1) * W + (k & 1)) * lds_ + e];
    }
    *(f32x4*)(dst + o * ldd + 4 * cq) = out;
  }
}

// ---------------------------------------------------------------- bilinear x2, align_corners=True
// ATen: scale = (in-1)/(out-1) (0 if out==1); src = scale*dst; i0 = (int)src; i1 = i0 + (i0 < in-1);
//       l1 = src - i0; l0 = 1 - l1.
__device__ __forceinline__ void lerp_coord(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  // the product is ROUNDED before the subtraction below (ATen's kernels do): left to the compiler it is contracted into
  // fma(scale, o, -i0), which moves the weight by up to half an ulp of the coordinate (1.4e-5 of the result at 2H = 140)
#pragma clang fp contract(off)
  const float s = scale * (float)o;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

// Linear workgroup ids go round-robin over the 8 XCDs, each with its own L2; an output row and its neighbours read the same two
// source rows, so dealing consecutive rows to different XCDs made every L2 fetch the source on its own (r3/r4_a PMC: 6.6x the
// source bytes from HBM in the forward, 2.3x the gradient bytes in the backward).  ssg_xcd_band gives each XCD one contiguous band
// of rows instead, walked in order.
// One workgroup = one part of one output image row (n * OH + oy): the row's vertical coordinates are computed
// once and the column index stays 32-bit (the flat 64-bit index of round 1 spent three 64-bit divisions per 16 bytes written:
// 2.9 TB/s on a write-bound pass).
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ldx,
                                                           float* __restrict__ y, int ldy, float sy, float sx, unsigned parts) {
  const int OH = 2 * H, OW = 2 * W, CQ = C / 4;
  const unsigned v = ssg_xcd_band(blockIdx.x, gridDim.x);
  const int row = (int)(v / parts);
  const unsigned part = v - (unsigned)row * parts;
  const int n = row / OH, oy = row - n * OH;
  int y0, y1; float ly0, ly1;
  lerp_coord(oy, sy, H, y0, y1, ly0, ly1);
  const float* b0 = x + ((size_t)n * H + y0) * W * ldx;
  const float* b1 = x + ((size_t)n * H + y1) * W * ldx;
  float* out = y + (size_t)row * OW * ldy;
  const unsigned cols = (unsigned)OW * (unsigned)CQ;
  for (unsigned j = part * 256u + threadIdx.x; j < cols; j += parts * 256u) {
    const unsigned ox = j / (unsigned)CQ, cq = j - ox * (unsigned)CQ;
    int x0, x1; float lx0, lx1;
    lerp_coord((int)ox, sx, W, x0, x1, lx0, lx1);
    const f32x4 v00 = *(const f32x4*)(b0 + (size_t)x0 * ldx + 4 * cq), v01 = *(const f32x4*)(b0 + (size_t)x1 * ldx + 4 * cq);
    const f32x4 v10 = *(const f32x4*)(b1 + (size_t)x0 * ldx + 4 * cq), v11 = *(const f32x4*)(b1 + (size_t)x1 * ldx + 4 * cq);
    const f32x4 o = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    *(f32x4*)(out + (size_t)ox * ldy + 4 * cq) = o;
  }
}

// backward as a gather over input pixels (deterministic): each input row iy collects from the
// few output rows whose (y0|y1) equals iy, recomputed with the forward's exact arithmetic.
__device__ __forceinline__ void cand_range(int i, float scale, int out, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  lo = (int)floorf((float)(i - 1) / scale) - 1;
  hi = (int)ceilf((float)(i + 1) / scale) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, int lddy, int N, int H, int W, int C,
                                                           float* __restrict__ dx, int lddx, float sy, float sx, unsigned parts) {
  const int OH = 2 * H, OW = 2 * W, CQ = C / 4;
  const unsigned v = ssg_xcd_band(blockIdx.x, gridDim.x);
  const int row = (int)(v / parts);                       // n * H + iy
  const unsigned part = v - (unsigned)row * parts;
  const int n = row / H, iy = row - n * H;
  int ylo, yhi;
  cand_range(iy, sy, OH, ylo, yhi);
  const unsigned cols = (unsigned)W * (unsigned)CQ;
  for (unsigned j = part * 256u + threadIdx.x; j < cols; j += parts * 256u) {
    const int ix = (int)(j / (unsigned)CQ), cq = (int)(j - (unsigned)ix * (unsigned)CQ);
    int xlo, xhi;
    cand_range(ix, sx, OW, xlo, xhi);
    f32x4 acc = {0, 0, 0, 0};
    const float* b = dy + (size_t)n * OH * OW * lddy + 4 * cq;
    for (int oy = ylo; oy <= yhi; ++oy) {
      int y0, y1; float ly0, ly1;
      lerp_coord(oy, sy, H, y0, y1, ly0, ly1);
      float wy = 0.f;
      if (y0 == iy) wy += ly0;
      if (y1 == iy) wy += ly1;
      if (wy == 0.f && y0 != iy && y1 != iy) continue;
      f32x4 rowacc = {0, 0, 0, 0};
      for (int ox = xlo; ox <= xhi; ++ox) {
        int x0, x1; float lx0, lx1;
        lerp_coord(ox, sx, W, x0, x1, lx0, lx1);
        if (x0 != ix && x1 != ix) continue;
        const f32x4 g = *(const f32x4*)(b + ((size_t)oy * OW + ox) * lddy);
        // keep the forward's grouping: contribution = ly * (lx * g)
        if (x0 == ix) rowacc += lx0 * g;
        if (x1 == ix) rowacc += lx1 * g;
      }
      if (y0 == iy) acc += ly0 * rowacc;
      if (y1 == iy) acc += ly1 * rowacc;
    }
    *(f32x4*)(dx + ((size_t)(n * H + iy) * W + ix) * lddx + 4 * cq) = acc;
  }
}

// ---------------------------------------------------------------- bilinear x2, streaming forms (C % 16 == 0)
// The gather kernels above issue four 16-byte loads per 16 bytes written (forward) and scan a 7 x 7 candidate window per input
// pixel (backward): 3.1-3.5 TB/s of algorithmic bytes at the decoder's shapes even with the XCD bands (tools/micro_spatial.py).
// Here a thread owns one (column, channel quad) and walks DOWN a band of rows with the horizontal work kept in registers:
//   forward : h(y) = lx0 * x[y][x0] + lx1 * x[y][x1] of the two live source rows; out[oy] = ly0 * h(y0) + ly1 * h(y1); a new
//             source row costs two loads, an output row one store (one load per output element instead of four);
//   backward: h(oy) = sum over the <= 5 output columns that read input column ix of their weight * dy[oy][ox]; the two live input
//             rows accumulate ly0 * h and ly1 * h and leave in row order (the vertical index never moves by more than one per
//             output row: scale < 1) -- every gradient element is loaded by the <= 2 threads whose columns it feeds.
// Same arithmetic grouping as the gather kernels (ly * (lx * v)), rows in ascending order: the results agree to the last bit
// except where x0 == x1 (last column), whose two weights are added before the multiply.
constexpr int BIL_FWD_BAND = 32;                         // output rows per workgroup
constexpr int BIL_BWD_BAND = 16;                         // input rows per workgroup

template <int CQT>                                       // channel quads per workgroup (16 / 8 / 4); 256 / CQT columns
__global__ __launch_bounds__(256) void bilinear_fwd_stream_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ldx,
                                                                  float* __restrict__ y, int ldy, float sy, float sx,
                                                                  int nct, int ncb, int nbands) {
  constexpr int OXT = 256 / CQT;
  const int OH = 2 * H, OW = 2 * W;
  unsigned u = blockIdx.x;
  const int ct = (int)(u % (unsigned)nct); u /= (unsigned)nct;
  const int cb = (int)(u % (unsigned)ncb); u /= (unsigned)ncb;
  const int band = (int)(u % (unsigned)nbands);
  const int n = (int)(u / (unsigned)nbands);
  const int cq = threadIdx.x % CQT, oxl = threadIdx.x / CQT;
  const int ox = cb * OXT + oxl;
  if (ox >= OW) return;
  const int c4 = (ct * CQT + cq) * 4;
  int x0, x1; float lx0, lx1;
  lerp_coord(ox, sx, W, x0, x1, lx0, lx1);
  const float* p0 = x + ((size_t)n * H * W + x0) * ldx + c4;
  const float* p1 = x + ((size_t)n * H * W + x1) * ldx + c4;
  const size_t rs = (size_t)W * ldx;
  float* out = y + ((size_t)n * OH * OW + ox) * ldy + c4;
  const int o0 = band * BIL_FWD_BAND, o1 = min(o0 + BIL_FWD_BAND, OH);
  auto hrow = [&](int r) { return lx0 * *(const f32x4*)(p0 + (size_t)r * rs) + lx1 * *(const f32x4*)(p1 + (size_t)r * rs); };
  int y0, y1; float ly0, ly1;
  lerp_coord(o0, sy, H, y0, y1, ly0, ly1);
  int cur = y0;
  f32x4 h0 = hrow(cur), h1 = hrow(min(cur + 1, H - 1));
  for (int oy = o0; oy < o1; ++oy) {
    lerp_coord(oy, sy, H, y0, y1, ly0, ly1);
    if (y0 > cur) { cur = y0; h0 = h1; h1 = hrow(min(cur + 1, H - 1)); }      // y0 == cur + 1: the scale is below one
    *(f32x4*)(out + (size_t)oy * OW * ldy) = ly0 * h0 + ly1 * h1;               // y1 == y0 only on the last row, where h1 == h0
  }
}

template <int CQT>
__global__ __launch_bounds__(256) void bilinear_bwd_stream_kernel(const float* __restrict__ dy, int lddy, int N, int H, int W, int C,
                                                                  float* __restrict__ dx, int lddx, float sy, float sx,
                                                                  int nct, int ncb, int nbands) {
  constexpr int IXT = 256 / CQT;
  const int OH = 2 * H, OW = 2 * W;
  unsigned u = blockIdx.x;
  const int ct = (int)(u % (unsigned)nct); u /= (unsigned)nct;
  const int cb = (int)(u % (unsigned)ncb); u /= (unsigned)ncb;
  const int band = (int)(u % (unsigned)nbands);
  const int n = (int)(u / (unsigned)nbands);
  const int cq = threadIdx.x % CQT, ixl = threadIdx.x / CQT;
  const int ix = cb * IXT + ixl;
  if (ix >= W) return;
  const int c4 = (ct * CQT + cq) * 4;
  // the output columns that read input column ix: a run of <= 5 (ox * sx in [ix - 1, ix + 1), sx >= 0.4: the launcher's condition)
  int xlo, xhi;
  cand_range(ix, sx, OW, xlo, xhi);
  int oxf = -1, cnt = 0; float wx[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int ox = xlo; ox <= xhi; ++ox) {
    int x0, x1; float lx0, lx1;
    lerp_coord(ox, sx, W, x0, x1, lx0, lx1);
    if (x0 != ix && x1 != ix) continue;
    if (oxf < 0) oxf = ox;
    const float wgt = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
    const int k = ox - oxf;
    if (k < 5) cnt = k + 1;
#pragma unroll
    for (int q = 0; q < 5; ++q) if (q == k) wx[q] = wgt;
  }
  const int b0 = band * BIL_BWD_BAND, b1 = min(b0 + BIL_BWD_BAND, H);
  int oylo, oyhi, t;
  cand_range(b0, sy, OH, oylo, t);
  cand_range(b1 - 1, sy, OH, t, oyhi);
  const float* src = dy + ((size_t)n * OH * OW + (oxf < 0 ? 0 : oxf)) * lddy + c4;
  float* dst = dx + ((size_t)n * H * W + ix) * lddx + c4;
  auto flush = [&](int r, const f32x4& v) { if (r >= b0 && r < b1) *(f32x4*)(dst + (size_t)r * W * lddx) = v; };
  int y0, y1; float ly0, ly1;
  lerp_coord(oylo, sy, H, y0, y1, ly0, ly1);
  int cur = y0;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int oy = oylo; oy <= oyhi; ++oy) {
    lerp_coord(oy, sy, H, y0, y1, ly0, ly1);
    const float* row = src + (size_t)oy * OW * lddy;
    f32x4 h = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 5; ++q)
      if (q < cnt) h += wx[q] * *(const f32x4*)(row + (size_t)q * lddy);
    if (y0 > cur) { flush(cur, acc0); acc0 = acc1; acc1 = f32x4{0, 0, 0, 0}; cur = y0; }
    acc0 += ly0 * h;
    if (y1 != y0) acc1 += ly1 * h; else acc0 += ly1 * h;
  }
  flush(cur, acc0);
  flush(cur + 1, acc1);
}

// ---------------------------------------------------------------- nearest x2
__global__ __launch_bounds__(256) void nearest_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ldx,
                                                          float* __restrict__ y, int ldy) {
  const int OH = 2 * H, OW = 2 * W, CQ = C / 4;
  const long long total = (long long)N * OH * OW * CQ;
  GRID_STRIDE(i, total) {
    const int cq = (int)(i % CQ); long long r = i / CQ;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH); const int n = (int)(r / OH);
    *(f32x4*)(y + ((size_t)(n * OH + oy) * OW + ox) * ldy + 4 * cq) =
        *(const f32x4*)(x + ((size_t)(n * H + (oy >> 1)) * W + (ox >> 1)) * ldx + 4 * cq);
  }
}
__global__ __launch_bounds__(256) void nearest_bwd_kernel(const float* __restrict__ dy, int lddy, int N, int H, int W, int C,
                                                          float* __restrict__ dx, int lddx) {
  const int OW = 2 * W, CQ = C / 4;
  const long long total = (long long)N * H * W * CQ;
  GRID_STRIDE(i, total) {
    const int cq = (int)(i % CQ); long long r = i / CQ;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H); const int n = (int)(r / H);
    const float* b = dy + ((size_t)(n * 2 * H + 2 * iy) * OW + 2 * ix) * lddy + 4 * cq;
    const f32x4 s = (*(const f32x4*)b + *(const f32x4*)(b + lddy)) + (*(const f32x4*)(b + (size_t)OW * lddy) + *(const f32x4*)(b + (size_t)(OW + 1) * lddy));
    *(f32x4*)(dx + ((size_t)(n * H + iy) * W + ix) * lddx + 4 * cq) = s;
  }
}

// ---------------------------------------------------------------- adaptive avg pool -> flat NCHW order
__device__ __forceinline__ int bin_lo(int o, int in, int out) { return (o * in) / out; }
__device__ __forceinline__ int bin_hi(int o, int in, int out) { return ((o + 1) * in + out - 1) / out; }

__global__ __launch_bounds__(256) void avgpool_flat_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ldx,
                                                               int O, float* __restrict__ y) {
  const long long total = (long long)N * O * O * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C); long long r = i / C;
    const int ox = (int)(r % O); r /= O;
    const int oy = (int)(r % O); const int n = (int)(r / O);
    const int y0 = bin_lo(oy, H, O), y1 = bin_hi(oy, H, O), x0 = bin_lo(ox, W, O), x1 = bin_hi(ox, W, O);
    float s = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) s += x[((size_t)(n * H + yy) * W + xx) * ldx + c];
    y[(size_t)n * C * O * O + (size_t)c * O * O + oy * O + ox] = s / (float)((y1 - y0) * (x1 - x0));
  }
}
__global__ __launch_bounds__(256) void avgpool_flat_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int C, int O,
                                                               float* __restrict__ dx, int lddx) {
  const long long total = (long long)N * H * W * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C); long long r = i / C;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H); const int n = (int)(r / H);
    float s = 0.f;
    for (int oy = 0; oy < O; ++oy) {
      const int y0 = bin_lo(oy, H, O), y1 = bin_hi(oy, H, O);
      if (yy < y0 || yy >= y1) continue;
      for (int ox = 0; ox < O; ++ox) {
        const int x0 = bin_lo(ox, W, O), x1 = bin_hi(ox, W, O);
        if (xx < x0 || xx >= x1) continue;
        s += dy[(size_t)n * C * O * O + (size_t)c * O * O + oy * O + ox] / (float)((y1 - y0) * (x1 - x0));
      }
    }
    dx[((size_t)(n * H + yy) * W + xx) * lddx + c] = s;
  }
}

// ---------------------------------------------------------------- NCHW <-> NHWC(ld)
// 32x32 LDS transpose per (n, 32 pixels, 32 channels) tile so both sides stay coalesced.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, int C, long long S, float* __restrict__ dst, int ld) {
  // grid: x = pixel tiles of 64, y = n; threads: 64 pixels x 4 channel lanes; small C (3..) fast path
  const long long p = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
  const int n = blockIdx.y;
  if (p >= S) return;
  for (int c = threadIdx.x >> 6; c < ld; c += 4) {
    const float v = c < C ? src[((size_t)n * C + c) * S + p] : 0.f;
    dst[((size_t)n * S + p) * ld + c] = v;
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, int ld, int C, long long S, float* __restrict__ dst) {
  const long long p = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
  const int n = blockIdx.y;
  if (p >= S) return;
  for (int c = threadIdx.x >> 6; c < C; c += 4) dst[((size_t)n * C + c) * S + p] = src[((size_t)n * S + p) * ld + c];
}

// streaming bilinear kernels: 16-channel tiles, scales in [0.4, 0.5) on both axes (H, W >= 3), a grid that fits 32 bits
static bool bilinear_stream_ok(int N, int H, int W, int C) {
  static const int on = [] { const char* e = getenv("SSG_BILINEAR_STREAM"); return e ? atoi(e) : 1; }();
  return on && C % 16 == 0 && H >= 3 && W >= 3 && (long long)N * (2 * H / BIL_FWD_BAND + 1) * (2 * W / 16 + 1) * (C / 16) < (1ll << 31);
}
#define REQ_Q(C, ...) SSG_REQUIRE((C) > 0 && (C) % 4 == 0, SSG_EINVAL, __VA_ARGS__)

}  // namespace

extern "C" int ssg_maxpool2x2_fwd_f32(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy, uint8_t* idx, void* stream) {
  SSG_REQUIRE(x && y && idx && N > 0 && H >= 2 && W >= 2, SSG_EINVAL, "maxpool: bad args");
  REQ_Q(C, "maxpool: C %% 4");
  SSG_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, SSG_EALIGN, "maxpool: strides");
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(elem_grid(total)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ldx, y, ldy, idx);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_maxpool2x2_bwd_f32(const float* dy, int lddy, const uint8_t* idx, int N, int H, int W, int C, float* dx, int lddx, void* stream) {
  SSG_REQUIRE(dy && dx && idx && N > 0 && H % 2 == 0 && W % 2 == 0, SSG_EINVAL, "maxpool_bwd: needs even H, W");
  REQ_Q(C, "maxpool_bwd: C %% 4");
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(scatter2x2_kernel, dim3(elem_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, lddy, idx, N, H / 2, W / 2, C, dx, lddx, (const float*)nullptr, 0);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_maxpool2x2_bwd_add_f32(const float* dy, int lddy, const uint8_t* idx, const float* res, int ldr, int N, int H, int W, int C,
                                         float* dx, int lddx, void* stream) {
  SSG_REQUIRE(dy && dx && idx && res && N > 0 && H % 2 == 0 && W % 2 == 0 && ldr % 4 == 0, SSG_EINVAL, "maxpool_bwd_add: bad args");
  REQ_Q(C, "maxpool_bwd_add: C %% 4");
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(scatter2x2_kernel, dim3(elem_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, lddy, idx, N, H / 2, W / 2, C, dx, lddx, res, ldr);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_maxunpool2x2_fwd_f32(const float* x, int ldx, const uint8_t* idx, int N, int OH, int OW, int C, float* y, int ldy, void* stream) {
  SSG_REQUIRE(x && y && idx && N > 0 && OH % 2 == 0 && OW % 2 == 0, SSG_EINVAL, "maxunpool: bad args");
  REQ_Q(C, "maxunpool: C %% 4");
  const long long total = (long long)N * (OH / 2) * (OW / 2) * (C / 4);
  hipLaunchKernelGGL(scatter2x2_kernel, dim3(elem_grid(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, idx, N, OH / 2, OW / 2, C, y, ldy, (const float*)nullptr, 0);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_maxunpool2x2_bwd_f32(const float* dy, int lddy, const uint8_t* idx, int N, int OH, int OW, int C, float* dx, int lddx, void* stream) {
  SSG_REQUIRE(dy && dx && idx && N > 0 && OH % 2 == 0 && OW % 2 == 0, SSG_EINVAL, "maxunpool_bwd: bad args");
  REQ_Q(C, "maxunpool_bwd: C %% 4");
  const long long total = (long long)N * (OH / 2) * (OW / 2) * (C / 4);
  hipLaunchKernelGGL(gather2x2_kernel, dim3(elem_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, lddy, idx, N, OH / 2, OW / 2, C, dx, lddx);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_upsample2x_bilinear_fwd_f32(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy, void* stream) {
  SSG_REQUIRE(x && y && N > 0 && H > 0 && W > 0, SSG_EINVAL, "bilinear: bad args");
  REQ_Q(C, "bilinear: C %% 4");
  const float sy = (2 * H > 1) ? (float)(H - 1) / (float)(2 * H - 1) : 0.f, sx = (2 * W > 1) ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
  SSG_REQUIRE((long long)N * 2 * H < (1ll << 31) && (long long)2 * W * (C / 4) < (1ll << 31), SSG_EINVAL, "bilinear: too large");
  if (bilinear_stream_ok(N, H, W, C)) {
    const int cqt = C % 64 == 0 ? 16 : C % 32 == 0 ? 8 : 4;
    const int nct = C / (4 * cqt), ncb = (2 * W + 256 / cqt - 1) / (256 / cqt), nb = (2 * H + BIL_FWD_BAND - 1) / BIL_FWD_BAND;
    const dim3 grid((unsigned)((long long)N * nb * ncb * nct));
    if (cqt == 16) hipLaunchKernelGGL(bilinear_fwd_stream_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ldx, y, ldy, sy, sx, nct, ncb, nb);
    else if (cqt == 8) hipLaunchKernelGGL(bilinear_fwd_stream_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ldx, y, ldy, sy, sx, nct, ncb, nb);
    else hipLaunchKernelGGL(bilinear_fwd_stream_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ldx, y, ldy, sy, sx, nct, ncb, nb);
    SSG_LAUNCH_CHECK();
    return SSG_OK;
  }
  const unsigned cols = (unsigned)(2 * W) * (unsigned)(C / 4);
  const unsigned gy = cols / 1024 ? (cols / 1024 > 64 ? 64 : cols / 1024) : 1;              // ~4 column steps per thread
  SSG_REQUIRE((long long)N * 2 * H * gy < (1ll << 31), SSG_EINVAL, "bilinear: too large");
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3((unsigned)(N * 2 * H) * gy), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ldx, y, ldy, sy, sx, gy);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_upsample2x_bilinear_bwd_f32(const float* dy, int lddy, int N, int H, int W, int C, float* dx, int lddx, void* stream) {
  SSG_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0, SSG_EINVAL, "bilinear_bwd: bad args");
  REQ_Q(C, "bilinear_bwd: C %% 4");
  const float sy = (2 * H > 1) ? (float)(H - 1) / (float)(2 * H - 1) : 0.f, sx = (2 * W > 1) ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
  SSG_REQUIRE((long long)N * H < (1ll << 31) && (long long)W * (C / 4) < (1ll << 31), SSG_EINVAL, "bilinear_bwd: too large");
  if (bilinear_stream_ok(N, H, W, C)) {
    const int cqt = C % 64 == 0 ? 16 : C % 32 == 0 ? 8 : 4;
    const int nct = C / (4 * cqt), ncb = (W + 256 / cqt - 1) / (256 / cqt), nb = (H + BIL_BWD_BAND - 1) / BIL_BWD_BAND;
    const dim3 grid((unsigned)((long long)N * nb * ncb * nct));
    if (cqt == 16) hipLaunchKernelGGL(bilinear_bwd_stream_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, dy, lddy, N, H, W, C, dx, lddx, sy, sx, nct, ncb, nb);
    else if (cqt == 8) hipLaunchKernelGGL(bilinear_bwd_stream_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, dy, lddy, N, H, W, C, dx, lddx, sy, sx, nct, ncb, nb);
    else hipLaunchKernelGGL(bilinear_bwd_stream_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, dy, lddy, N, H, W, C, dx, lddx, sy, sx, nct, ncb, nb);
    SSG_LAUNCH_CHECK();
    return SSG_OK;
  }
  const unsigned cols = (unsigned)W * (unsigned)(C / 4);
  const unsigned gy = (cols + 255) / 256 > 64 ? 64 : (cols + 255) / 256;
  SSG_REQUIRE((long long)N * H * gy < (1ll << 31), SSG_EINVAL, "bilinear_bwd: too large");
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3((unsigned)(N * H) * gy), dim3(256), 0, (hipStream_t)stream, dy, lddy, N, H, W, C, dx, lddx, sy, sx, gy);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_upsample2x_nearest_fwd_f32(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy, void* stream) {
  SSG_REQUIRE(x && y && N > 0 && H > 0 && W > 0, SSG_EINVAL, "nearest: bad args");
  REQ_Q(C, "nearest: C %% 4");
  const long long total = (long long)N * 4 * H * W * (C / 4);
  hipLaunchKernelGGL(nearest_fwd_kernel, dim3(elem_grid(total)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ldx, y, ldy);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_upsample2x_nearest_bwd_f32(const float* dy, int lddy, int N, int H, int W, int C, float* dx, int lddx, void* stream) {
  SSG_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0, SSG_EINVAL, "nearest_bwd: bad args");
  REQ_Q(C, "nearest_bwd: C %% 4");
  const long long total = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(nearest_bwd_kernel, dim3(elem_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, lddy, N, H, W, C, dx, lddx);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_adaptive_avgpool_flat_fwd_f32(const float* x, int N, int H, int W, int C, int ldx, int OHW, float* y, void* stream) {
  SSG_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && OHW > 0, SSG_EINVAL, "avgpool: bad args");
  const long long total = (long long)N * OHW * OHW * C;
  hipLaunchKernelGGL(avgpool_flat_fwd_kernel, dim3(elem_grid(total)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ldx, OHW, y);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_adaptive_avgpool_flat_bwd_f32(const float* dy, int N, int H, int W, int C, int OHW, float* dx, int lddx, void* stream) {
  SSG_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && OHW > 0, SSG_EINVAL, "avgpool_bwd: bad args");
  const long long total = (long long)N * H * W * C;
  hipLaunchKernelGGL(avgpool_flat_bwd_kernel, dim3(elem_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, N, H, W, C, OHW, dx, lddx);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_nchw_to_nhwc_f32(const float* src, int N, int C, int H, int W, float* dst, int ld, void* stream) {
  SSG_REQUIRE(src && dst && N > 0 && C > 0 && ld >= C, SSG_EINVAL, "nchw_to_nhwc: bad args");
  const long long S = (long long)H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)ssg_cdiv(S, 64), (unsigned)N), dim3(256), 0, (hipStream_t)stream, src, C, S, dst, ld);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
extern "C" int ssg_nhwc_to_nchw_f32(const float* src, int ld, int N, int C, int H, int W, float* dst, void* stream) {
  SSG_REQUIRE(src && dst && N > 0 && C > 0 && ld >= C, SSG_EINVAL, "nhwc_to_nchw: bad args");
  const long long S = (long long)H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)ssg_cdiv(S, 64), (unsigned)N), dim3(256), 0, (hipStream_t)stream, src, ld, C, S, dst);
  SSG_LAUNCH_CHECK();
  return SSG_OK;
}
