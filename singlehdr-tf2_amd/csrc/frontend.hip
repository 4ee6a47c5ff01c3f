// Linearization-Net front end on gfx950: spatial-aware soft histogram and the
// fused [img | sobel | hist4 | hist8 | hist16] feature tensor.
//
// Both kernels are HBM-write bound (12 B read, 4*3B resp. 384 B written per
// pixel): one thread produces one 16-byte quad of output channels so that a
// wavefront writes 1 KiB contiguously; the 3 input floats of a pixel are
// shared through L1 by the 24 threads that expand it.
//
// Bit-exactness: the bin value is evaluated exactly as
// the reference does -- centre = fp32(2i-1)/fp32(2B) (IEEE divide),
// d = |x - centre|, h = d < fp32(1/B) ? 1 - d*B : 0 with the multiply and the
// subtract rounded separately (__fmul_rn/__fsub_rn forbid FMA contraction).
#include <stdlib.h>

#include "shdr_internal.h"

namespace {
typedef float nt4 __attribute__((ext_vector_type(4)));       // for __builtin_nontemporal_store


__device__ __forceinline__ float soft_bin(float x, int i /*1..B*/, float two_b, float nb, float thr) {
#pragma clang fp contract(off)  // this file is also built with -ffp-contract=off
  const float centre = __fdiv_rn((float)(2 * i - 1), two_b);
  const float d = fabsf(__fsub_rn(x, centre));
  return d < thr ? __fsub_rn(1.0f, __fmul_rn(d, nb)) : 0.0f;
}

// y [npix, B*C], channel ch = (bin-1)*C + c.  Scalar-element version (any B*C).
__global__ __launch_bounds__(256) void soft_hist_scalar_kernel(const float* __restrict__ x,
                                                               float* __restrict__ y, long npix,
                                                               int C, int B, float thr) {
  const int CO = B * C;
  const long total = npix * CO;
  const float two_b = (float)(2 * B), nb = (float)B;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long p = e / CO;
    const int ch = (int)(e - p * CO);
    const int bin = ch / C, c = ch - bin * C;
    y[e] = soft_bin(x[p * C + c], bin + 1, two_b, nb, thr);
  }
}

// Quad version: (B*C) % 4 == 0; one float4 store per thread.
__global__ __launch_bounds__(256) void soft_hist_quad_kernel(const float* __restrict__ x,
                                                             float* __restrict__ y, long npix,
                                                             int C, int B, float thr) {
  const int Q = (B * C) >> 2;
  const long total = npix * Q;
  const float two_b = (float)(2 * B), nb = (float)B;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long p = e / Q;
    const int ch0 = (int)(e - p * Q) * 4;
    const float* xp = x + p * C;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ch = ch0 + k;
      const int bin = ch / C, c = ch - bin * C;
      v[k] = soft_bin(xp[c], bin + 1, two_b, nb, thr);
    }
    *reinterpret_cast<float4*>(y + e * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// Same output, HBM-bound form: the grid stride is a multiple of the quads per pixel, so a thread keeps ONE output quad
// position (4 channels = 4 (bin, colour) pairs) for all its pixels; the four bin centres -- the IEEE divides of soft_bin --
// and the channel indices are computed once per thread instead of once per element, and the 64-bit e / Q per store is gone.
// The per-element arithmetic is soft_bin's, bit for bit.
__global__ __launch_bounds__(256) void soft_hist_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long npix,
                                                             int C, int B, float thr) {
#pragma clang fp contract(off)
  const int Q = (B * C) >> 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int q = (int)(t % Q);
  const long dp = ((long)gridDim.x * 256) / Q;              // exact: the launcher makes the stride a multiple of Q
  const float two_b = (float)(2 * B), nb = (float)B;
  int cc[4];
  float centre[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ch = 4 * q + k;
    const int bin = ch / C;
    cc[k] = ch - bin * C;
    centre[k] = __fdiv_rn((float)(2 * (bin + 1) - 1), two_b);
  }
  auto one = [&](long p, const float* xv) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = fabsf(__fsub_rn(xv[k], centre[k]));
      v[k] = d < thr ? __fsub_rn(1.0f, __fmul_rn(d, nb)) : 0.0f;
    }
    __builtin_nontemporal_store((nt4){v[0], v[1], v[2], v[3]}, reinterpret_cast<nt4*>(y + (p * Q + q) * 4));      // written once, read by another kernel
  };
  long p = t / Q;
  for (; p + 3 * dp < npix; p += 4 * dp) {                  // four pixels in flight
    float xv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) xv[u][k] = x[(p + u * dp) * C + cc[k]];
#pragma unroll
    for (int u = 0; u < 4; ++u) one(p + u * dp, xv[u]);
  }
  for (; p < npix; p += dp) {
    float xv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = x[p * C + cc[k]];
    one(p, xv);
  }
}

__device__ __forceinline__ int reflect(int i, int n) {  // REFLECT pad by 1
  return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
}

// Feature channel `ch` (0..92) of pixel (h, w); rgb = centre pixel.
__device__ __forceinline__ float frontend_channel(const float* __restrict__ img, long ibase, int h,
                                                  int w, int H, int W, const float* rgb, int ch) {
  if (ch < 3) return rgb[ch];
  if (ch < 9) {  // sobel: channel = 3 + c*2 + {0: dy, 1: dx}   (linearization_net.py:312-314)
    const int c = (ch - 3) >> 1, dir = (ch - 3) & 1;
    const int hm = reflect(h - 1, H), hp = reflect(h + 1, H);
    const int wm = reflect(w - 1, W), wp = reflect(w + 1, W);
    auto at = [&](int hh, int ww) { return img[(ibase + (long)hh * W + ww) * 3 + c]; };
    if (dir == 0)  // dy: -(row above) + (row below), weights 1 2 1
      return (at(hp, wm) - at(hm, wm)) + 2.0f * (at(hp, w) - at(hm, w)) + (at(hp, wp) - at(hm, wp));
    return (at(hm, wp) - at(hm, wm)) + 2.0f * (at(h, wp) - at(h, wm)) + (at(hp, wp) - at(hp, wm));
  }
  int idx, B;
  if (ch < 21) { idx = ch - 9; B = 4; }
  else if (ch < 45) { idx = ch - 21; B = 8; }
  else if (ch < 93) { idx = ch - 45; B = 16; }
  else return 0.0f;
  const int bin = idx / 3, c = idx - bin * 3;
  // B is 4, 8 or 16 here: 1/(2B) is a power of two, so (2i-1) * (1/(2B)) IS the correctly rounded quotient (2i-1)/(2B) -- the
  // IEEE divide of soft_bin (needed for arbitrary B) is replaced by one exact multiply; d*B and 1/B are exact as well
  const float centre = (float)(2 * bin + 1) * (0.5f / (float)B);
  const float d = fabsf(__fsub_rn(rgb[c], centre));
  return d < 1.0f / (float)B ? __fsub_rn(1.0f, __fmul_rn(d, (float)B)) : 0.0f;
}

// grid.y = image row (n*H + h), grid.x covers the W * QPP quads of that row: no runtime division
// per thread (QPP is a compile-time constant), the 3 rows of the sobel stencil are block-uniform.
template <int YC>
__global__ __launch_bounds__(256) void lin_frontend_kernel(const float* __restrict__ img,
                                                           float* __restrict__ y, int N, int H, int W) {
  constexpr int QPP = (YC + 3) / 4;   // quads per pixel
  const int row = blockIdx.y;
  const int h = row % H;
  const long ibase = (long)(row / H) * H * W;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= W * QPP) return;
  const int w = idx / QPP, ch0 = (idx - w * QPP) * 4;
  const long p = ibase + (long)h * W + w;
  const float rgb[3] = {img[p * 3], img[p * 3 + 1], img[p * 3 + 2]};
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = frontend_channel(img, ibase, h, w, H, W, rgb, ch0 + k);
  float* yp = y + p * YC + ch0;
  if ((YC & 3) == 0) {
    *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (ch0 + k < YC) yp[k] = v[k];
  }
}

// Row-segment version of the fused front end: a block owns 64 consecutive pixels of one image row and builds their 96-channel
// feature rows in LDS -- wave 0 the 9 image / sobel channels (one pixel per lane), all four waves the 84 histogram channels
// (thread = (pixel, channel) pairs; bin centre, bin count and colour index come from the channel number by arithmetic, no
// divergent paths) -- then streams the 24 KB tile to HBM with fully coalesced float4 stores.  The per-quad kernel above mixes
// the sobel and histogram code paths in every wave and reaches 1.9 TB/s of stores; this one is store-bound.
template <int YC>
__global__ __launch_bounds__(256) void lin_frontend_rows_kernel(const float* __restrict__ img, float* __restrict__ y, int N, int H,
                                                                int W) {
  constexpr int PX = 64;
  __shared__ __attribute__((aligned(16))) float tile[PX * 96];
  __shared__ float rgbs[PX * 3];
  const int row = blockIdx.y;
  const int h = row % H;
  const long ibase = (long)(row / H) * H * W;
  const int w0 = blockIdx.x * PX;
  const int tid = threadIdx.x;
  if (tid < PX) {                                    // wave 0: image + sobel (linearization_net.py:312-314)
    const int w = min(w0 + tid, W - 1);
    const long p = ibase + (long)h * W + w;
    const int hm = reflect(h - 1, H), hp = reflect(h + 1, H), wm = reflect(w - 1, W), wp = reflect(w + 1, W);
    float* t = tile + tid * 96;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      auto at = [&](int hh, int ww) { return img[(ibase + (long)hh * W + ww) * 3 + c]; };
      const float v = img[p * 3 + c];
      t[c] = v;
      rgbs[tid * 3 + c] = v;
      t[3 + 2 * c] = (at(hp, wm) - at(hm, wm)) + 2.0f * (at(hp, w) - at(hm, w)) + (at(hp, wp) - at(hm, wp));
      t[4 + 2 * c] = (at(hm, wp) - at(hm, wm)) + 2.0f * (at(h, wp) - at(h, wm)) + (at(hp, wp) - at(hp, wm));
    }
    t[93] = 0.0f; t[94] = 0.0f; t[95] = 0.0f;
  }
  __syncthreads();
  // histogram channels 9 .. 92: 84 per pixel.  Thread = (pixel phase g of 3, channel k): the channel's bin centre, bin count and
  // colour index are computed ONCE per thread (the element-major loop spent 25 VALU operations per value on e / 84, idx / 3 and
  // the level select: the kernel was VALU-bound at 4.3 TB/s); consecutive threads = consecutive channels of one pixel
  if (tid < 252) {
    const int g = tid / 84, k = tid - g * 84;         // k: 0..11 -> B = 4, 12..35 -> B = 8, 36..83 -> B = 16
    const int lvl = k < 12 ? 0 : (k < 36 ? 1 : 2);
    const int idx = k - (lvl == 0 ? 0 : (lvl == 1 ? 12 : 36));
    const int B = 4 << lvl;
    const int bin = idx / 3, c = idx - bin * 3;
    // 1/(2B) is a power of two: (2i-1) * (1/(2B)) is the correctly rounded quotient of soft_bin's IEEE divide
    const float centre = (float)(2 * bin + 1) * (0.5f / (float)B);
    const float fB = (float)B, w = 1.0f / fB;
#pragma unroll 4
    for (int px = g; px < PX; px += 3) {
      const float d = fabsf(__fsub_rn(rgbs[px * 3 + c], centre));
      tile[px * 96 + 9 + k] = d < w ? __fsub_rn(1.0f, __fmul_rn(d, fB)) : 0.0f;
    }
  }
  __syncthreads();
  const int npx = min(PX, W - w0);
  float* yo = y + (ibase + (long)h * W + w0) * YC;
  if (YC == 96) {
    const int nq = npx * 24;
    for (int q = tid; q < nq; q += 256) __builtin_nontemporal_store(*reinterpret_cast<const nt4*>(tile + 4 * q), reinterpret_cast<nt4*>(yo + 4 * q));
  } else {
    const int n = npx * 93;
    for (int e = tid; e < n; e += 256) yo[e] = tile[(e / 93) * 96 + e % 93];
  }
}

}  // namespace

namespace {
// dx[p][c] = sum_bins dy[p][(bin-1)*C + c] * dh/dx,  h = 1 - |x - centre| * B inside the support |x - centre| < 1/B (slope -+B), 0 outside
__global__ __launch_bounds__(256) void soft_hist_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                            long n, int C, int B) {
  const float fb = (float)B, inv_b = 1.0f / fb;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const long p = e / C;
    const int c = (int)(e - p * C);
    const float xv = x[e];
    const float* g = dy + p * (long)B * C + c;
    float acc = 0.0f;
    for (int i = 1; i <= B; ++i) {
      const float centre = (float)(2 * i - 1) / (float)(2 * B);
      const float d = xv - centre;
      if (fabsf(d) < inv_b) acc += g[(long)(i - 1) * C] * (d > 0.0f ? -fb : (d < 0.0f ? fb : 0.0f));
    }
    dx[e] = acc;
  }
}
}  // namespace

extern "C" int shdr_soft_hist_bwd_f32(const float* x, const float* dy, float* dx, int64_t npix, int C, int B, void* stream) {
  SHDR_REQUIRE(x && dy && dx, SHDR_E_NULL, "soft_hist_bwd: null pointer");
  SHDR_REQUIRE(npix > 0 && C > 0 && B > 0, SHDR_E_SHAPE, "soft_hist_bwd: npix, C, B must be positive");
  hipLaunchKernelGGL(soft_hist_bwd_kernel, dim3(shdr::stream_grid(npix * C)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, dy, dx,
                     (long)npix * C, C, B);
  return shdr::check_launch("soft_hist_bwd");
}

extern "C" int shdr_soft_hist_fwd_f32(const float* x, float* y, int64_t npix, int C, int B,
                                      void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "soft_hist: null pointer");
  SHDR_REQUIRE(npix >= 0 && C > 0 && B > 0 && B <= 4096, SHDR_E_SHAPE, "soft_hist: bad shape");
  if (npix == 0) return SHDR_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float thr = (float)(1.0 / (double)B);  // Python double 1./max_bin cast to fp32
  if (((B * C) & 3) == 0 && shdr::aligned16(y)) {
    const int Q = (B * C) >> 2;
    const long total = (long)npix * Q;
    // smallest grid unit whose 256-thread blocks make the stride a multiple of Q: Q / gcd(Q, 256) blocks
    int g = Q, h = 256;
    while (h) { const int r = g % h; g = h; h = r; }
    const long unit = Q / g;
    if (unit <= 64 && SHDR_ENV("SHDR_FRONTEND_QUADS") == nullptr) {
      long grid = shdr::stream_grid(total);
      grid = (grid + unit - 1) / unit * unit;
      hipLaunchKernelGGL(soft_hist_rows_kernel, dim3((unsigned)grid), dim3(256), 0, st, x, y, (long)npix, C, B, thr);
    } else {
      hipLaunchKernelGGL(soft_hist_quad_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0, st, x, y,
                         (long)npix, C, B, thr);
    }
  } else {
    const long total = (long)npix * B * C;
    hipLaunchKernelGGL(soft_hist_scalar_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0, st, x,
                       y, (long)npix, C, B, thr);
  }
  return shdr::check_launch("soft_hist");
}

extern "C" int shdr_lin_frontend_fwd_f32(const float* img, float* y, int N, int H, int W,
                                         int y_channels, void* stream) {
  SHDR_REQUIRE(img && y, SHDR_E_NULL, "lin_frontend: null pointer");
  SHDR_REQUIRE(N > 0 && H >= 2 && W >= 2, SHDR_E_SHAPE, "lin_frontend: need N>0, H,W>=2 (REFLECT pad)");
  SHDR_REQUIRE(y_channels >= 93, SHDR_E_SHAPE, "lin_frontend: y_channels must be >= 93");
  SHDR_REQUIRE((y_channels & 3) != 0 || shdr::aligned16(y), SHDR_E_ALIGN, "lin_frontend: y not 16-byte aligned");
  SHDR_REQUIRE(y_channels == 93 || y_channels == 96, SHDR_E_SHAPE, "lin_frontend: y_channels must be 93 or 96");
  SHDR_REQUIRE((long)N * H <= 65535, SHDR_E_SHAPE, "lin_frontend: N*H must be <= 65535");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (SHDR_ENV("SHDR_FRONTEND_QUADS") != nullptr) {          // the per-quad kernel (kept for comparison)
    const dim3 grid((unsigned)((W * 24 + 255) / 256), (unsigned)(N * H));
    if (y_channels == 96) hipLaunchKernelGGL(lin_frontend_kernel<96>, grid, dim3(256), 0, st, img, y, N, H, W);
    else hipLaunchKernelGGL(lin_frontend_kernel<93>, grid, dim3(256), 0, st, img, y, N, H, W);
    return shdr::check_launch("lin_frontend");
  }
  const dim3 grid((unsigned)((W + 63) / 64), (unsigned)(N * H));
  if (y_channels == 96) hipLaunchKernelGGL(lin_frontend_rows_kernel<96>, grid, dim3(256), 0, st, img, y, N, H, W);
  else hipLaunchKernelGGL(lin_frontend_rows_kernel<93>, grid, dim3(256), 0, st, img, y, N, H, W);
  return shdr::check_launch("lin_frontend");
}
