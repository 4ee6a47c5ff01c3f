// fp16 twins of the HBM-bound tape ops, for the native-fp16 activation layout of BASELINE configs[4] (csrc/conv_f16.hip): every
// feature map of the fine-tuning chain is NHWC fp16 with C % 8 == 0, so one thread moves 16 bytes = 8 channels; all arithmetic
// is fp32 in registers, reductions are fp32 / fp64, only the stored tensors are fp16 (half the bytes of the fp32 kernels in
// bwd.hip / pool.hip / glue.hip / frontend.hip, whose semantics -- tie-breaking of the max-pools, border handling of the
// bilinear resize, BatchNorm statistics in double -- they restate).  Image-like tensors (3 channels) stay fp32.
// Replaces the same TF op call sites as their _f32 counterparts (include/shdr.h).
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "shdr_internal.h"

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef _Float16 hf;

struct V8 {
  float v[8];
};
__device__ __forceinline__ V8 ldh(const hf* p) {
  const f16x8 t = *reinterpret_cast<const f16x8*>(p);
  V8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (float)t[i];
  return r;
}
__device__ __forceinline__ void sth(hf* p, const V8& a) {
  f16x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (hf)a.v[i];
  *reinterpret_cast<f16x8*>(p) = t;
}
__device__ __forceinline__ V8 zero8() {
  V8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0.f;
  return r;
}
__device__ __forceinline__ V8 ld8f(const float* p) {          // 8 consecutive floats (32-byte aligned source)
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  V8 r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// e -> (n, h, w, o) of a [N, Hd, Wd, 8*O] tensor; 32-bit divisions while the index fits (see pool.hip)
#define DECODE8(e, O, Wd, Hd, o, w, h, n)                              \
  int o, w, h;                                                          \
  long n;                                                               \
  if ((unsigned long)(e) <= 0xffffffffUL) {                             \
    unsigned _t = (unsigned)(e);                                        \
    o = (int)(_t % (unsigned)(O)); _t /= (unsigned)(O);                 \
    w = (int)(_t % (unsigned)(Wd)); _t /= (unsigned)(Wd);               \
    h = (int)(_t % (unsigned)(Hd)); n = (long)(_t / (unsigned)(Hd));    \
  } else {                                                              \
    long _t = (e);                                                      \
    o = (int)(_t % (O)); _t /= (O);                                     \
    w = (int)(_t % (Wd)); _t /= (Wd);                                   \
    h = (int)(_t % (Hd)); n = _t / (Hd);                                \
  }

__device__ __forceinline__ float act_grad(float g, float y, int act) {
  switch (act) {
    case SHDR_ACT_RELU: return y > 0.f ? g : 0.f;
    case SHDR_ACT_LRELU: return y > 0.f ? g : 0.1f * g;
    case SHDR_ACT_TANH: return g * (1.0f - y * y);
    default: return g;
  }
}

// ---- casts / packing ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float* __restrict__ x, hf* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = (hf)x[i];
}
__global__ __launch_bounds__(256) void cast_f16_f32_kernel(const hf* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = (float)x[i];
}
// y[p][c] = c < Cin ? x[p][c] : 0 with a dtype change (fp32 [npix, Cin] -> fp16 [npix, Cout] and back, Cout % 8 == 0 on the fp16 side)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void pad_cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, long npix, int Cin, int Cout) {
  const long total = npix * Cout;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long p = e / Cout;
    const int c = (int)(e - p * Cout);
    y[e] = c < Cin ? (TO)(float)x[p * Cin + c] : (TO)0.f;
  }
}
struct Src4 {
  const float* s[4];
};
// concat of up to four 3-channel fp32 images -> fp16 [npix, OC] (zero-padded), optionally with the VGG preprocessing of
// hallucination_net.py:149-153 on source 0 (x*255, RGB -> BGR, minus mean)
__global__ __launch_bounds__(256) void pack3_h_kernel(Src4 src, int nsrc, hf* __restrict__ y, int OC, long npix, int vgg) {
  const int O = OC >> 3;
  const long total = npix * O;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long p = e / O;
    const int o = (int)(e - p * O);
    V8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = 8 * o + j, s = ch / 3, c = ch - 3 * s;
      float v = 0.f;
      if (s < nsrc) {
        const float* sp = s == 0 ? src.s[0] : (s == 1 ? src.s[1] : (s == 2 ? src.s[2] : src.s[3]));
        if (vgg) {
          const float mean = c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f);
          v = sp[3 * p + (2 - c)] * 255.0f - mean;
        } else {
          v = sp[3 * p + c];
        }
      }
      r.v[j] = v;
    }
    sth(y + e * 8, r);
  }
}
struct Dst4 {
  float* d[4];
};
// inverse: the first `nout` 3-channel slices of an fp16 [npix, C] tensor -> fp32 images; vgg: the backward of the preprocessing
__global__ __launch_bounds__(256) void unpack3_h_kernel(const hf* __restrict__ y, Dst4 out, int nout, int C, long npix, int vgg) {
  const long total = npix * 3 * nout;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long p = e / (3 * nout);
    const int ch = (int)(e - p * 3 * nout), s = ch / 3, c = ch - 3 * s;
    float* op = s == 0 ? out.d[0] : (s == 1 ? out.d[1] : (s == 2 ? out.d[2] : out.d[3]));
    if (vgg) op[3 * p + c] = (float)y[p * C + (2 - c)] * 255.0f;
    else op[3 * p + c] = (float)y[p * C + ch];
  }
}

// ---- activation backward + bias gradient -----------------------------------------------------------------------------------
// dz = dy * act'(y) (skipped when act == NONE: dz == dy), db[c] += sum_p dz[p][c] (skipped when db == null).  A thread's channel
// octet is fixed over its grid-stride loop (the launcher makes gridDim * 256 a multiple of O), partial sums go through LDS
// atomics, one global atomic per block and channel.
__global__ __launch_bounds__(256) void act_bwd_bias_h_kernel(const hf* __restrict__ dy, const hf* __restrict__ y, hf* __restrict__ dz,
                                                             float* __restrict__ db, float* __restrict__ ws, long nvec, int O, int act) {
  extern __shared__ float sdb[];                              // [8 * O]
  if (db) {
    for (int i = threadIdx.x; i < 8 * O; i += 256) sdb[i] = 0.f;
    __syncthreads();
  }
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long step = (long)gridDim.x * 256;
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  const int o = (int)(e % O);
  auto one = [&](long i, const V8& g, const V8& yv) {
    V8 d = g;
    if (act != SHDR_ACT_NONE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) d.v[j] = act_grad(g.v[j], yv.v[j], act);
      sth(dz + 8 * i, d);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += d.v[j];
  };
  for (; e + 3 * step < nvec; e += 4 * step) {                // four vectors per operand in flight
    V8 g[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      g[u] = ldh(dy + 8 * (e + u * step));
      yv[u] = g[u];
      if (act != SHDR_ACT_NONE) yv[u] = ldh(y + 8 * (e + u * step));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(e + u * step, g[u], yv[u]);
  }
  for (; e < nvec; e += step) {
    const V8 g0 = ldh(dy + 8 * e);
    V8 y0 = g0;
    if (act != SHDR_ACT_NONE) y0 = ldh(y + 8 * e);
    one(e, g0, y0);
  }
  if (db) {
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&sdb[8 * o + j], s[j]);
    __syncthreads();
    if (ws) {                                                  // this block's partial row (folded by shdr::col_fold_kernel)
      for (int i = threadIdx.x; i < 8 * O; i += 256) ws[(size_t)blockIdx.x * 8 * O + i] = sdb[i];
    } else {
      for (int i = threadIdx.x; i < 8 * O; i += 256) atomicAdd(db + i, sdb[i]);
    }
  }
}

// y = act(a + b): the fan-in joins of the tape (residual adds, gradient accumulation at fan-out points)
__global__ __launch_bounds__(256) void add_h_kernel(const hf* __restrict__ a, const hf* __restrict__ b, hf* __restrict__ y, long nvec,
                                                    int relu) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) {
    const V8 x0 = ldh(a + 8 * e), x1 = ldh(b + 8 * e);
    V8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      r.v[j] = x0.v[j] + x1.v[j];
      if (relu) r.v[j] = fmaxf(r.v[j], 0.f);
    }
    sth(y + 8 * e, r);
  }
}

// ---- pooling / resize ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool2_h_kernel(const hf* __restrict__ x, hf* __restrict__ y, int N, int H, int W, int C, int is_max) {
  const int Ho = H >> 1, Wo = W >> 1, O = C >> 3;
  const long total = (long)N * Ho * Wo * O;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    DECODE8(e, O, Wo, Ho, o, ow, oh, n)
    const hf* p = x + (((n * H + 2 * oh) * W + 2 * ow) * (long)C + 8 * o);
    const V8 a = ldh(p), b = ldh(p + C), c = ldh(p + (long)W * C), d = ldh(p + (long)W * C + C);
    V8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      r.v[j] = is_max ? fmaxf(fmaxf(a.v[j], b.v[j]), fmaxf(c.v[j], d.v[j])) : ((a.v[j] + b.v[j]) + (c.v[j] + d.v[j])) * 0.25f;
    sth(y + e * 8, r);
  }
}
__global__ __launch_bounds__(256) void avgpool2_bwd_h_kernel(const hf* __restrict__ dy, hf* __restrict__ dx, int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, O = C >> 3;
  const long total = (long)N * H * W * O;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    DECODE8(e, O, W, H, o, w, h, n)
    V8 g = zero8();
    if ((h >> 1) < Ho && (w >> 1) < Wo) {
      g = ldh(dy + (((n * Ho + (h >> 1)) * Wo + (w >> 1)) * (long)C + 8 * o));
#pragma unroll
      for (int j = 0; j < 8; ++j) g.v[j] *= 0.25f;
    }
    sth(dx + e * 8, g);
  }
}
// the gradient goes to the FIRST maximum of the window in row-major scan order (as maxpool2_bwd_kernel)
__global__ __launch_bounds__(256) void maxpool2_bwd_h_kernel(const hf* __restrict__ x, const hf* __restrict__ dy, hf* __restrict__ dx,
                                                             int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, O = C >> 3;
  const long total = (long)N * Ho * Wo * O;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    DECODE8(e, O, Wo, Ho, o, ow, oh, n)
    const long base = ((n * H + 2 * oh) * W + 2 * ow) * (long)C + 8 * o;
    const long off[4] = {0, (long)C, (long)W * C, (long)W * C + C};
    V8 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ldh(x + base + off[k]);
    const V8 g = ldh(dy + e * 8);
    V8 r[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int arg = 0;
      float m = v[0].v[c];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k].v[c] > m) { m = v[k].v[c]; arg = k; }
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k].v[c] = (k == arg) ? g.v[c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sth(dx + base + off[k], r[k]);
  }
}
__global__ __launch_bounds__(256) void maxpool3s2_h_kernel(const hf* __restrict__ x, hf* __restrict__ y, int N, int H, int W, int C, int Ho,
                                                           int Wo, int pt, int pl) {
  const int O = C >> 3;
  const long total = (long)N * Ho * Wo * O;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    DECODE8(e, O, Wo, Ho, o, ow, oh, n)
    V8 m;
#pragma unroll
    for (int j = 0; j < 8; ++j) m.v[j] = -__builtin_huge_valf();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ih = 2 * oh - pt + i;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int iw = 2 * ow - pl + k;
        if ((unsigned)iw >= (unsigned)W) continue;
        const V8 t = ldh(x + (((n * H + ih) * W + iw) * (long)C + 8 * o));
#pragma unroll
        for (int j = 0; j < 8; ++j) m.v[j] = fmaxf(m.v[j], t.v[j]);
      }
    }
    sth(y + e * 8, m);
  }
}
// candidates are the elements equal to the window's maximum `y`; only earlier equals can take the gradient (maxpool3s2_bwd_kernel)
__global__ __launch_bounds__(256) void maxpool3s2_bwd_h_kernel(const hf* __restrict__ x, const hf* __restrict__ y, const hf* __restrict__ dy,
                                                               hf* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo, int pt,
                                                               int pl) {
  const int O = C >> 3;
  const long total = (long)N * H * W * O;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    DECODE8(e, O, W, H, o, w, h, n)
    const V8 mine = ldh(x + e * 8);
    V8 acc = zero8();
    for (int oh = max(0, (h + pt - 1) / 2); oh <= min(Ho - 1, (h + pt) / 2); ++oh) {
      for (int ow = max(0, (w + pl - 1) / 2); ow <= min(Wo - 1, (w + pl) / 2); ++ow) {
        const int h0 = 2 * oh - pt, w0 = 2 * ow - pl;
        if (h < h0 || h > h0 + 2 || w < w0 || w > w0 + 2) continue;
        const long widx = ((n * Ho + oh) * Wo + ow) * (long)C + 8 * o;
        const V8 m = ldh(y + widx);
        unsigned win = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) win |= (unsigned)(mine.v[c] == m.v[c]) << c;
        if (!win) continue;
        for (int ih = max(h0, 0); ih <= h; ++ih) {
          const int wend = (ih == h) ? w : min(w0 + 3, W);
          for (int iw = max(w0, 0); iw < wend; ++iw) {
            const V8 t = ldh(x + (((n * H + ih) * W + iw) * (long)C + 8 * o));
#pragma unroll
            for (int c = 0; c < 8; ++c) win &= ~((unsigned)(t.v[c] == mine.v[c]) << c);
          }
        }
        if (!win) continue;
        const V8 g = ldh(dy + widx);
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (win >> c & 1) acc.v[c] += g.v[c];
      }
    }
    sth(dx + e * 8, acc);
  }
}
// tf.image.resize(2x, BILINEAR), half-pixel centres: one thread owns one INPUT octet and writes its 2x2 outputs (resize2x_kernel)
__global__ __launch_bounds__(256) void resize2x_h_kernel(const hf* __restrict__ x, hf* __restrict__ y, int N, int H, int W, int C) {
  const int O = C >> 3;
  const long total = (long)N * H * W * O;
  const long orow = (long)2 * W * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    DECODE8(e, O, W, H, o, w, h, n)
    const int hm = max(h - 1, 0), hp = min(h + 1, H - 1), wm = max(w - 1, 0), wp = min(w + 1, W - 1);
    const hf* b = x + (n * H * (long)W) * C + 8 * o;
    const int rows[3] = {hm, h, hp};
    V8 lo[3], hi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const hf* rp = b + (long)rows[r] * W * C;
      const V8 l = ldh(rp + (long)wm * C), c = ldh(rp + (long)w * C), rr = ldh(rp + (long)wp * C);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        lo[r].v[j] = l.v[j] + (c.v[j] - l.v[j]) * 0.75f;
        hi[r].v[j] = c.v[j] + (rr.v[j] - c.v[j]) * 0.25f;
      }
    }
    hf* op = y + ((n * 2 * H + 2 * h) * (long)(2 * W) + 2 * w) * C + 8 * o;
    V8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t.v[j] = lo[0].v[j] + (lo[1].v[j] - lo[0].v[j]) * 0.75f;
    sth(op, t);
#pragma unroll
    for (int j = 0; j < 8; ++j) t.v[j] = hi[0].v[j] + (hi[1].v[j] - hi[0].v[j]) * 0.75f;
    sth(op + C, t);
#pragma unroll
    for (int j = 0; j < 8; ++j) t.v[j] = lo[1].v[j] + (lo[2].v[j] - lo[1].v[j]) * 0.25f;
    sth(op + orow, t);
#pragma unroll
    for (int j = 0; j < 8; ++j) t.v[j] = hi[1].v[j] + (hi[2].v[j] - hi[1].v[j]) * 0.25f;
    sth(op + orow + C, t);
  }
}
__device__ __forceinline__ void resize_taps(int m, int n_in, int* idx, float* wt, int* cnt) {
  int k = 0;
  if (m > 0) { idx[k] = 2 * m - 1; wt[k] = 0.25f; ++k; }
  idx[k] = 2 * m; wt[k] = (m == 0) ? 1.0f : 0.75f; ++k;
  idx[k] = 2 * m + 1; wt[k] = (m == n_in - 1) ? 1.0f : 0.75f; ++k;
  if (m < n_in - 1) { idx[k] = 2 * m + 2; wt[k] = 0.25f; ++k; }
  *cnt = k;
}
__global__ __launch_bounds__(256) void resize2x_bwd_h_kernel(const hf* __restrict__ dy, hf* __restrict__ dx, int N, int H, int W, int C) {
  const int O = C >> 3, Ho = 2 * H, Wo = 2 * W;
  const long total = (long)N * H * W * O;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    DECODE8(e, O, W, H, o, w, h, n)
    int yi[4], xi[4], ny, nx;
    float yw[4], xw[4];
    resize_taps(h, H, yi, yw, &ny);
    resize_taps(w, W, xi, xw, &nx);
    V8 s = zero8();
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        const V8 g = ldh(dy + (((n * Ho + yi[a]) * Wo + xi[b]) * (long)C + 8 * o));
        const float wgt = yw[a] * xw[b];
#pragma unroll
        for (int j = 0; j < 8; ++j) s.v[j] += wgt * g.v[j];
      }
    sth(dx + e * 8, s);
  }
}
__global__ __launch_bounds__(256) void upsample_zero2_h_kernel(const hf* __restrict__ dy, hf* __restrict__ dx, int N, int H, int W, int C) {
  const int O = C >> 3, Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
  const long total = (long)N * H * W * O;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    DECODE8(e, O, W, H, o, w, h, n)
    V8 g = zero8();
    if (((h | w) & 1) == 0) g = ldh(dy + (((n * Ho + (h >> 1)) * Wo + (w >> 1)) * (long)C + 8 * o));
    sth(dx + e * 8, g);
  }
}
// tf.reduce_mean(x, [1,2]): fp16 [N, HW, C] -> fp32 [N, C]; grid (ceil(C / 128), N), thread = (octet 0..15, pixel lane 0..15)
__global__ __launch_bounds__(256) void gap_h_kernel(const hf* __restrict__ x, float* __restrict__ y, int HW, int C) {
  __shared__ float part[16][16][8];
  const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 128 + 8 * q;
  const long n = blockIdx.y;
  V8 s = zero8();
  if (c0 < C) {
    const hf* xb = x + n * HW * (long)C + c0;
    V8 s1 = s, s2 = s, s3 = s;
    int p = g;
    for (; p + 48 < HW; p += 64) {
      const V8 a = ldh(xb + (long)p * C), b = ldh(xb + (long)(p + 16) * C), c = ldh(xb + (long)(p + 32) * C), d = ldh(xb + (long)(p + 48) * C);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s.v[j] += a.v[j]; s1.v[j] += b.v[j]; s2.v[j] += c.v[j]; s3.v[j] += d.v[j]; }
    }
    for (; p < HW; p += 16) {
      const V8 a = ldh(xb + (long)p * C);
#pragma unroll
      for (int j = 0; j < 8; ++j) s.v[j] += a.v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] = (s.v[j] + s1.v[j]) + (s2.v[j] + s3.v[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[g][q][j] = s.v[j];
  __syncthreads();
  if (g == 0 && c0 < C) {
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += part[i][q][j];
      y[n * C + c0 + j] = t * inv;
    }
  }
}
__global__ __launch_bounds__(256) void gap_bwd_h_kernel(const float* __restrict__ dy, hf* __restrict__ dx, long total_v, int HW, int C) {
  const int O = C >> 3;
  const float inv = 1.0f / (float)HW;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total_v; e += (long)gridDim.x * 256) {
    const int o = (int)(e % O);
    const long n = e / ((long)O * HW);
    V8 g = ld8f(dy + n * C + 8 * o);
#pragma unroll
    for (int j = 0; j < 8; ++j) g.v[j] *= inv;
    sth(dx + e * 8, g);
  }
}

// ---- BatchNormalization, training mode (statistics in double, as bn_reduce4_kernel) -------------------------------------------
//   mode 0 : ws[c] += sum x,   ws[C + c] += sum x*x
//   mode 1 : ws[c] += sum dy', ws[C + c] += sum dy' * (x - mean),   dy' = dy masked by y > 0 when y != null
__global__ __launch_bounds__(256) void bn_reduce_h_kernel(const hf* __restrict__ a, const hf* __restrict__ x, const hf* __restrict__ y,
                                                          const float* __restrict__ mean, double* __restrict__ ws, long npix, int C, int mode) {
  __shared__ double part[16][256];
  const int O = C >> 3;
  int OL = 1;
  while (OL < O && OL < 256) OL <<= 1;
  const int PL = 256 / OL;
  const int ol = threadIdx.x % OL, pl = threadIdx.x / OL;
  for (int o0 = 0; o0 < O; o0 += OL) {
    const int o = o0 + ol;
    double s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.0; s2[j] = 0.0; }
    if (o < O) {
      V8 mu = zero8();
      if (mode) mu = ld8f(mean + 8 * o);
      const long step = (long)gridDim.x * PL;
      // fp32 partial sums over short runs (<= 16 pixels: their rounding is 2^-24 relative, far below the fp16 input), folded into
      // the double accumulators once per run -- the double adds of every element made the kernel VALU-bound at ~2 TB/s
      float f1[8], f2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { f1[j] = 0.f; f2[j] = 0.f; }
      int run = 0;
      auto acc = [&](const V8& av, const V8& xv, const V8& yv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (mode == 0) {
            f1[j] += av.v[j]; f2[j] += av.v[j] * av.v[j];
          } else {
            const float gg = (y && !(yv.v[j] > 0.f)) ? 0.f : av.v[j];
            f1[j] += gg; f2[j] += gg * (xv.v[j] - mu.v[j]);
          }
        }
      };
      auto fold = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += (double)f1[j]; s2[j] += (double)f2[j]; f1[j] = 0.f; f2[j] = 0.f; }
        run = 0;
      };
      long p = (long)blockIdx.x * PL + pl;
      if (mode == 0) {
        for (; p + 7 * step < npix; p += 8 * step) {         // statistics pass: eight 16-byte loads in flight per thread
          V8 av[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) av[u] = ldh(a + (p + u * step) * C + 8 * o);
#pragma unroll
          for (int u = 0; u < 8; ++u) acc(av[u], av[u], av[u]);
          run += 8;
          if (run >= 16) fold();
        }
      }
      for (; p + 3 * step < npix; p += 4 * step) {           // four vectors per operand in flight
        V8 av[4], xv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long i = (p + u * step) * C + 8 * o;
          av[u] = ldh(a + i);
          xv[u] = av[u]; yv[u] = av[u];
          if (mode) {
            xv[u] = ldh(x + i);
            if (y) yv[u] = ldh(y + i);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc(av[u], xv[u], yv[u]);
        run += 4;
        if (run >= 16) fold();
      }
      for (; p < npix; p += step) {
        const long i = p * C + 8 * o;
        const V8 a0 = ldh(a + i);
        V8 x0 = a0, y0 = a0;
        if (mode) {
          x0 = ldh(x + i);
          if (y) y0 = ldh(y + i);
        }
        acc(a0, x0, y0);
      }
      fold();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { part[j][threadIdx.x] = s1[j]; part[8 + j][threadIdx.x] = s2[j]; }
    __syncthreads();
    if (pl == 0 && o < O) {
      double* wp = ws + (size_t)(1 + blockIdx.x) * 2 * C;       // this block's partial row; summed by bn_fold_h_kernel (no atomics, no memset)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        double t1 = s1[j], t2 = s2[j];
        for (int k = 1; k < PL; ++k) { t1 += part[j][k * OL + ol]; t2 += part[8 + j][k * OL + ol]; }
        wp[8 * o + j] = t1;
        wp[C + 8 * o + j] = t2;
      }
    }
    __syncthreads();
  }
}
// ws[col] = sum over the g partial rows ws[(1 + b) * 2C + col]: one wave per column, lanes stride over the rows
__global__ __launch_bounds__(256) void bn_fold_h_kernel(double* __restrict__ ws, int C, int g) {
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (col >= 2 * C) return;
  double t = 0.0;
  for (int b = lane; b < g; b += 64) t += ws[(size_t)(1 + b) * 2 * C + col];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  if (lane == 0) ws[col] = t;
}
__global__ void bn_finalize_h_kernel(const double* __restrict__ ws, float* __restrict__ mean, float* __restrict__ var,
                                     float* __restrict__ mov_mean, float* __restrict__ mov_var, long npix, int C, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = ws[c] / (double)npix;
  double v = ws[C + c] / (double)npix - mu * mu;
  if (v < 0.0) v = 0.0;
  mean[c] = (float)mu;
  var[c] = (float)v;
  if (mov_mean) {
    const double unbiased = npix > 1 ? v * (double)npix / (double)(npix - 1) : v;
    mov_mean[c] = mov_mean[c] * momentum + (float)mu * (1.0f - momentum);
    mov_var[c] = mov_var[c] * momentum + (float)unbiased * (1.0f - momentum);
  }
}
__global__ void bn_bwd_finalize_h_kernel(const double* __restrict__ ws, const float* __restrict__ var, float* __restrict__ dgamma,
                                         float* __restrict__ dbeta, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] += (float)ws[c];
  dgamma[c] += (float)(ws[C + c] * (double)rsqrtf(var[c] + eps));
}
// forward apply (mode 0): y = [relu]((x - mean) * rstd * gamma + beta);  backward apply (mode 1):
// dx = gamma * rstd * (dy' - mean(dy') - xhat * mean(dy' * xhat)).  The launcher makes gridDim * 256 a multiple of O, so the
// per-channel factors are computed once per thread.
__global__ __launch_bounds__(256) void bn_apply_h_kernel(const hf* __restrict__ a, const hf* __restrict__ x, const hf* __restrict__ y,
                                                         const float* __restrict__ mean, const float* __restrict__ var,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const double* __restrict__ ws, hf* __restrict__ out, long nvec, long npix, int C,
                                                         float eps, int relu, int mode) {
  const int O = C >> 3;
  const int o = (int)(((long)blockIdx.x * 256 + threadIdx.x) % O);
  float rstd[8], mu[8], k0[8], k1[8], k2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 8 * o + j;
    rstd[j] = rsqrtf(var[c] + eps);
    mu[j] = mean[c];
    if (mode == 0) {
      k0[j] = rstd[j] * gamma[c];
      k1[j] = beta[c];
      k2[j] = 0.f;
    } else {
      k0[j] = gamma[c] * rstd[j];
      k1[j] = (float)(ws[c] / (double)npix);
      k2[j] = (float)(ws[C + c] / (double)npix) * rstd[j];
    }
  }
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) {
    V8 r;
    if (mode == 0) {
      const V8 xv = ldh(a + 8 * e);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r.v[j] = (xv.v[j] - mu[j]) * k0[j] + k1[j];
        if (relu) r.v[j] = fmaxf(r.v[j], 0.f);
      }
    } else {
      V8 g = ldh(a + 8 * e);
      const V8 xv = ldh(x + 8 * e);
      if (y) {
        const V8 yv = ldh(y + 8 * e);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (!(yv.v[j] > 0.f)) g.v[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (xv.v[j] - mu[j]) * rstd[j];
        r.v[j] = k0[j] * (g.v[j] - k1[j] - xh * k2[j]);
      }
    }
    sth(out + 8 * e, r);
  }
}

// ---- Linearization-Net front end (linearization_net.py:310-322, 336-350) -> fp16 [N,H,W,96] and its backward --------------------
// channel layout as lin_frontend_rows_kernel: [img 3 | sobel 6 (c*2 + {dy,dx}) | hist4 12 | hist8 24 | hist16 48 | 0 0 0]
__device__ __forceinline__ float lin_feature(const float* __restrict__ img, long n, int h, int w, int H, int W, int ch) {
  const long base = (n * H + h) * (long)W + w;
  if (ch < 3) return img[3 * base + ch];
  if (ch < 9) {
    const int c = (ch - 3) >> 1, k = (ch - 3) & 1;            // k = 0: dy kernel [[-1,-2,-1],[0,0,0],[1,2,1]], 1: its transpose
    auto at = [&](int hh, int ww) {                           // REFLECT padding by one
      hh = hh < 0 ? -hh : (hh >= H ? 2 * H - 2 - hh : hh);
      ww = ww < 0 ? -ww : (ww >= W ? 2 * W - 2 - ww : ww);
      return img[3 * ((n * H + hh) * (long)W + ww) + c];
    };
    if (k == 0)
      return (at(h + 1, w - 1) + 2.f * at(h + 1, w) + at(h + 1, w + 1)) - (at(h - 1, w - 1) + 2.f * at(h - 1, w) + at(h - 1, w + 1));
    return (at(h - 1, w + 1) + 2.f * at(h, w + 1) + at(h + 1, w + 1)) - (at(h - 1, w - 1) + 2.f * at(h, w - 1) + at(h + 1, w - 1));
  }
  if (ch >= 93) return 0.f;
  int B, r;
  if (ch < 21) { B = 4; r = ch - 9; } else if (ch < 45) { B = 8; r = ch - 21; } else { B = 16; r = ch - 45; }
  const int bin = r / 3 + 1, c = r - 3 * (r / 3);
  const float xv = img[3 * base + c];
  const float centre = (float)(2 * bin - 1) / (float)(2 * B);
  const float d = fabsf(xv - centre);
  return d < 1.0f / (float)B ? 1.0f - d * (float)B : 0.0f;
}
// Block roles (no divergence inside a wave): every sixth block writes the two leading octets (image, sobel, first histogram channels:
// the general per-channel form, 18 neighbour loads per pixel and colour), the others the ten pure-histogram octets -- a thread owns ONE
// of them for all its pixels, so bin centre, bin count and colour index of its eight channels are computed once, then 3 loads + 8 hat
// functions + one 16-byte store per pixel (the element-major form decoded the channel per value: 0.92 ms for 4 x 1024^2).  YC = 96.
__global__ __launch_bounds__(256) void lin_frontend_h_kernel(const float* __restrict__ img, hf* __restrict__ y, int N, int H, int W, int YC) {
  const int O = YC >> 3;
  const long npix = (long)N * H * W;
  const bool lead = blockIdx.x % 6 == 0;
  if (!lead) {
    const int OH = O - 2;                                      // histogram octets 2 .. O-1
    const long bid = blockIdx.x - blockIdx.x / 6 - 1, nb = gridDim.x - (gridDim.x + 5) / 6;       // index among the histogram blocks
    const long t0 = bid * 256 + threadIdx.x, nthr = nb * 256;  // nthr is a multiple of OH (launcher)
    const int o = 2 + (int)(t0 % OH);
    int cj[8];
    float centre[8], fB[8], wB[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = 8 * o + j;
      int B = 4, r = ch - 9;
      if (ch >= 45) { B = 16; r = ch - 45; } else if (ch >= 21) { B = 8; r = ch - 21; }
      const int bin = r / 3 + 1;
      cj[j] = ch < 93 ? r - 3 * (r / 3) : -1;
      centre[j] = (float)(2 * bin - 1) / (float)(2 * B);
      fB[j] = (float)B;
      wB[j] = 1.0f / (float)B;
    }
    for (long p = t0 / OH; p < npix; p += nthr / OH) {
      const float x3[3] = {img[3 * p], img[3 * p + 1], img[3 * p + 2]};
      V8 r;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = cj[j] == 1 ? x3[1] : (cj[j] == 2 ? x3[2] : x3[0]);
        const float d = fabsf(xv - centre[j]);
        r.v[j] = (cj[j] >= 0 && d < wB[j]) ? 1.0f - d * fB[j] : 0.0f;
      }
      sth(y + (p * O + o) * 8, r);
    }
  } else {
    const long bid = blockIdx.x / 6, nb = (gridDim.x + 5) / 6;
    const long t0 = bid * 256 + threadIdx.x, nthr = nb * 256;
    const int o = (int)(t0 & 1);
    for (long p = t0 >> 1; p < npix; p += nthr >> 1) {
      const int w = (int)(p % W);
      const long t = p / W;
      const int h = (int)(t % H);
      const long n = t / H;
      V8 r;
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[j] = lin_feature(img, n, h, w, H, W, 8 * o + j);
      sth(y + (p * O + o) * 8, r);
    }
  }
}
// dimg[n,h,w,c] = dF[c] + sum of the sobel transposes + sum_bins dF[hist] * (-+B inside the bin support); gather form, no atomics
__global__ __launch_bounds__(256) void lin_frontend_bwd_h_kernel(const float* __restrict__ img, const hf* __restrict__ dF,
                                                                 float* __restrict__ dimg, int N, int H, int W, int YC) {
  const long total = (long)N * H * W * 3;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % 3);
    const long pix = e / 3;
    const int w = (int)(pix % W);
    const long t = pix / W;
    const int h = (int)(t % H);
    const long n = t / H;
    const hf* g = dF + pix * YC;
    float acc = (float)g[c];
    const float xv = img[e];
    // histogram slopes: h = 1 - |x - centre| * B inside the support
    const int Bs[3] = {4, 8, 16}, offs[3] = {9, 21, 45};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int B = Bs[s];
      for (int bin = 1; bin <= B; ++bin) {
        const float centre = (float)(2 * bin - 1) / (float)(2 * B);
        const float d = xv - centre;
        if (fabsf(d) < 1.0f / (float)B) acc += (float)g[offs[s] + 3 * (bin - 1) + c] * (d > 0.f ? -(float)B : (d < 0.f ? (float)B : 0.f));
      }
    }
    // sobel transpose: output pixel (hh, ww) read input (reflect(hh+i-1), reflect(ww+j-1)) with weight k[i][j]; gather every
    // output pixel whose reflected tap lands on (h, w)
    for (int hh = h - 2; hh <= h + 2; ++hh) {
      if (hh < 0 || hh >= H) continue;
      for (int ww = w - 2; ww <= w + 2; ++ww) {
        if (ww < 0 || ww >= W) continue;
        float wy = 0.f, wx = 0.f;
#pragma unroll
        for (int i = -1; i <= 1; ++i) {
          int rh = hh + i;
          rh = rh < 0 ? -rh : (rh >= H ? 2 * H - 2 - rh : rh);
          if (rh != h) continue;
#pragma unroll
          for (int j = -1; j <= 1; ++j) {
            int rw = ww + j;
            rw = rw < 0 ? -rw : (rw >= W ? 2 * W - 2 - rw : rw);
            if (rw != w) continue;
            wy += (float)i * (j == 0 ? 2.f : 1.f);           // dy kernel: rows -1 / +1 weighted (1,2,1)
            wx += (float)j * (i == 0 ? 2.f : 1.f);           // dx kernel: its transpose
          }
        }
        if (wy != 0.f || wx != 0.f) {
          const hf* go = dF + ((n * H + hh) * (long)W + ww) * YC + 3 + 2 * c;
          acc += wy * (float)go[0] + wx * (float)go[1];
        }
      }
    }
    dimg[e] = acc;
  }
}

int chk8(const char* op, const void* a, const void* b, int N, int H, int W, int C) {
  SHDR_REQUIRE(a && b, SHDR_E_NULL, "%s: null pointer", op);
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, SHDR_E_SHAPE, "%s: non-positive dimension", op);
  SHDR_REQUIRE((C & 7) == 0, SHDR_E_ALIGN, "%s: C=%d must be a multiple of 8 (16-byte fp16 channel groups)", op, C);
  SHDR_REQUIRE(shdr::aligned16(a) && shdr::aligned16(b), SHDR_E_ALIGN, "%s: tensors must be 16-byte aligned", op);
  return SHDR_OK;
}
// grid whose thread count is a multiple of O (a thread keeps its channel octet), at most `cap` blocks
inline int octet_grid(long nvec, int O, int cap) {
  long g = (nvec + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  int unit = O;
  for (int d = 256; d > 1; d >>= 1)
    if (unit % 2 == 0 && d > 1) { unit /= 2; } else break;      // unit = O / gcd(O, 256)
  g = (g + unit - 1) / unit * unit;
  return (int)g;
}
typedef const hf* HP;
typedef hf* HM;

}  // namespace

#define H_(p) reinterpret_cast<HP>(p)
#define HM_(p) reinterpret_cast<HM>(p)

extern "C" int shdr_cast_f32_to_f16(const float* x, void* y, int64_t n, void* stream) {
  SHDR_REQUIRE(x && y && n > 0, SHDR_E_NULL, "cast: bad arguments");
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(shdr::stream_grid(n)), dim3(256), 0, S(stream), x, HM_(y), (long)n);
  return shdr::check_launch("cast_f32_to_f16");
}
extern "C" int shdr_cast_f16_to_f32(const void* x, float* y, int64_t n, void* stream) {
  SHDR_REQUIRE(x && y && n > 0, SHDR_E_NULL, "cast: bad arguments");
  hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(shdr::stream_grid(n)), dim3(256), 0, S(stream), H_(x), y, (long)n);
  return shdr::check_launch("cast_f16_to_f32");
}
extern "C" int shdr_pad_channels_f32_to_f16(const float* x, void* y, int64_t npix, int Cin, int Cout, void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "pad_channels: null pointer");
  SHDR_REQUIRE(npix > 0 && Cin > 0 && Cout >= Cin, SHDR_E_SHAPE, "pad_channels: need Cout >= Cin > 0");
  hipLaunchKernelGGL((pad_cast_kernel<float, hf>), dim3(shdr::stream_grid(npix * Cout)), dim3(256), 0, S(stream), x, HM_(y), (long)npix, Cin, Cout);
  return shdr::check_launch("pad_channels_f32_to_f16");
}
extern "C" int shdr_pack3_f16(const float* s0, const float* s1, const float* s2, const float* s3, int nsrc, void* y, int out_channels,
                              int64_t npix, int vgg_preprocess, void* stream) {
  SHDR_REQUIRE(y && s0, SHDR_E_NULL, "pack3_f16: null pointer");
  SHDR_REQUIRE(nsrc >= 1 && nsrc <= 4 && out_channels >= 3 * nsrc && out_channels % 8 == 0 && npix > 0, SHDR_E_SHAPE,
               "pack3_f16: need 1 <= nsrc <= 4, out_channels >= 3 * nsrc, out_channels %% 8 == 0");
  SHDR_REQUIRE(!vgg_preprocess || nsrc == 1, SHDR_E_SHAPE, "pack3_f16: the VGG preprocessing takes one source");
  SHDR_REQUIRE(shdr::aligned16(y), SHDR_E_ALIGN, "pack3_f16: y must be 16-byte aligned");
  const float* s[4] = {s0, s1, s2, s3};
  for (int i = 0; i < nsrc; ++i) SHDR_REQUIRE(s[i], SHDR_E_NULL, "pack3_f16: source %d is null", i);
  Src4 src{{s0, s1, s2, s3}};
  hipLaunchKernelGGL(pack3_h_kernel, dim3(shdr::stream_grid(npix * (out_channels / 8))), dim3(256), 0, S(stream), src, nsrc, HM_(y),
                     out_channels, (long)npix, vgg_preprocess);
  return shdr::check_launch("pack3_f16");
}
extern "C" int shdr_unpack3_f16(const void* y, float* o0, float* o1, float* o2, float* o3, int nout, int channels, int64_t npix,
                                int vgg_preprocess_bwd, void* stream) {
  SHDR_REQUIRE(y && o0, SHDR_E_NULL, "unpack3_f16: null pointer");
  SHDR_REQUIRE(nout >= 1 && nout <= 4 && channels >= 3 * nout && npix > 0, SHDR_E_SHAPE, "unpack3_f16: need 1 <= nout <= 4, channels >= 3 * nout");
  float* o[4] = {o0, o1, o2, o3};
  for (int i = 0; i < nout; ++i) SHDR_REQUIRE(o[i], SHDR_E_NULL, "unpack3_f16: output %d is null", i);
  Dst4 dst{{o0, o1, o2, o3}};
  hipLaunchKernelGGL(unpack3_h_kernel, dim3(shdr::stream_grid(npix * 3 * nout)), dim3(256), 0, S(stream), H_(y), dst, nout, channels,
                     (long)npix, vgg_preprocess_bwd);
  return shdr::check_launch("unpack3_f16");
}
extern "C" int shdr_act_bwd_bias_f16(const void* dy, const void* y, void* dz, float* db, float* ws, int64_t npix, int C, int act, void* stream) {
  SHDR_REQUIRE(dy, SHDR_E_NULL, "act_bwd_bias_f16: null pointer");
  SHDR_REQUIRE(act == SHDR_ACT_NONE || (y && dz), SHDR_E_NULL, "act_bwd_bias_f16: y and dz are needed with an activation");
  SHDR_REQUIRE(npix > 0 && C > 0 && C % 8 == 0 && C <= 4096 && act >= 0 && act <= 3, SHDR_E_SHAPE, "act_bwd_bias_f16: bad arguments (C %% 8 == 0)");
  SHDR_REQUIRE(shdr::aligned16(dy) && (!y || shdr::aligned16(y)) && (!dz || shdr::aligned16(dz)), SHDR_E_ALIGN,
               "act_bwd_bias_f16: tensors must be 16-byte aligned");
  const int O = C / 8;
  const long nvec = (long)npix * O;
  const bool rows = db && ws && nvec >= (1L << 22);   // partial rows + fold instead of atomics on the same C addresses (pays from ~64 MB on)
  const int cap = db && !rows ? (nvec >= (1L << 23) ? 512 : 256) : 2048;
  int grid = octet_grid(nvec, O, cap);
  if (rows && grid > shdr::kBiasMaxBlocks) grid = octet_grid(nvec, O, shdr::kBiasMaxBlocks / 2);       // the rounding up to the octet unit stays below the row count
  hipLaunchKernelGGL(act_bwd_bias_h_kernel, dim3(grid), dim3(256), db ? 8 * O * sizeof(float) : 0, S(stream), H_(dy),
                     H_(y), HM_(dz), db, rows ? ws : (float*)nullptr, nvec, O, act);
  if (rows) shdr::launch_col_fold(ws, db, grid, C, S(stream));
  return shdr::check_launch("act_bwd_bias_f16");
}
extern "C" int shdr_add_f16(const void* a, const void* b, void* y, int64_t n, int relu, void* stream) {
  SHDR_REQUIRE(a && b && y, SHDR_E_NULL, "add_f16: null pointer");
  SHDR_REQUIRE(n > 0 && n % 8 == 0, SHDR_E_SHAPE, "add_f16: n must be a positive multiple of 8");
  SHDR_REQUIRE(shdr::aligned16(a) && shdr::aligned16(b) && shdr::aligned16(y), SHDR_E_ALIGN, "add_f16: tensors must be 16-byte aligned");
  hipLaunchKernelGGL(add_h_kernel, dim3(shdr::stream_grid(n / 8)), dim3(256), 0, S(stream), H_(a), H_(b), HM_(y), (long)(n / 8), relu);
  return shdr::check_launch("add_f16");
}
extern "C" int shdr_avgpool2_fwd_f16(const void* x, void* y, int N, int H, int W, int C, void* stream) {
  if (int rc = chk8("avgpool2_f16", x, y, N, H, W, C)) return rc;
  SHDR_REQUIRE(H >= 2 && W >= 2, SHDR_E_SHAPE, "avgpool2_f16: H, W must be >= 2");
  hipLaunchKernelGGL(pool2_h_kernel, dim3(shdr::stream_grid((long)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, S(stream), H_(x), HM_(y), N, H, W, C, 0);
  return shdr::check_launch("avgpool2_f16");
}
extern "C" int shdr_maxpool2_fwd_f16(const void* x, void* y, int N, int H, int W, int C, void* stream) {
  if (int rc = chk8("maxpool2_f16", x, y, N, H, W, C)) return rc;
  SHDR_REQUIRE((H & 1) == 0 && (W & 1) == 0, SHDR_E_SHAPE, "maxpool2_f16: H, W must be even");
  hipLaunchKernelGGL(pool2_h_kernel, dim3(shdr::stream_grid((long)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, S(stream), H_(x), HM_(y), N, H, W, C, 1);
  return shdr::check_launch("maxpool2_f16");
}
extern "C" int shdr_avgpool2_bwd_f16(const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
  if (int rc = chk8("avgpool2_bwd_f16", dy, dx, N, H, W, C)) return rc;
  hipLaunchKernelGGL(avgpool2_bwd_h_kernel, dim3(shdr::stream_grid((long)N * H * W * (C / 8))), dim3(256), 0, S(stream), H_(dy), HM_(dx), N, H, W, C);
  return shdr::check_launch("avgpool2_bwd_f16");
}
extern "C" int shdr_maxpool2_bwd_f16(const void* x, const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
  if (int rc = chk8("maxpool2_bwd_f16", x, dx, N, H, W, C)) return rc;
  SHDR_REQUIRE(dy && shdr::aligned16(dy), SHDR_E_NULL, "maxpool2_bwd_f16: dy null or unaligned");
  SHDR_REQUIRE((H & 1) == 0 && (W & 1) == 0, SHDR_E_SHAPE, "maxpool2_bwd_f16: H, W must be even");
  hipLaunchKernelGGL(maxpool2_bwd_h_kernel, dim3(shdr::stream_grid((long)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, S(stream), H_(x),
                     H_(dy), HM_(dx), N, H, W, C);
  return shdr::check_launch("maxpool2_bwd_f16");
}
extern "C" int shdr_maxpool3s2_fwd_f16(const void* x, void* y, int N, int H, int W, int C, void* stream) {
  if (int rc = chk8("maxpool3s2_f16", x, y, N, H, W, C)) return rc;
  int Ho, Wo, pt, pl;
  shdr_same_pad(H, 3, 2, &Ho, &pt);
  shdr_same_pad(W, 3, 2, &Wo, &pl);
  hipLaunchKernelGGL(maxpool3s2_h_kernel, dim3(shdr::stream_grid((long)N * Ho * Wo * (C / 8))), dim3(256), 0, S(stream), H_(x), HM_(y), N, H,
                     W, C, Ho, Wo, pt, pl);
  return shdr::check_launch("maxpool3s2_f16");
}
extern "C" int shdr_maxpool3s2_bwd_f16(const void* x, const void* y, const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
  if (int rc = chk8("maxpool3s2_bwd_f16", x, dx, N, H, W, C)) return rc;
  SHDR_REQUIRE(dy && shdr::aligned16(dy), SHDR_E_NULL, "maxpool3s2_bwd_f16: dy null or unaligned");
  SHDR_REQUIRE(y && shdr::aligned16(y), SHDR_E_NULL, "maxpool3s2_bwd_f16: y (the pooled output) null or unaligned");
  int Ho, Wo, pt, pl;
  shdr_same_pad(H, 3, 2, &Ho, &pt);
  shdr_same_pad(W, 3, 2, &Wo, &pl);
  hipLaunchKernelGGL(maxpool3s2_bwd_h_kernel, dim3(shdr::stream_grid((long)N * H * W * (C / 8))), dim3(256), 0, S(stream), H_(x), H_(y),
                     H_(dy), HM_(dx), N, H, W, C, Ho, Wo, pt, pl);
  return shdr::check_launch("maxpool3s2_bwd_f16");
}
extern "C" int shdr_resize2x_fwd_f16(const void* x, void* y, int N, int H, int W, int C, void* stream) {
  if (int rc = chk8("resize2x_f16", x, y, N, H, W, C)) return rc;
  hipLaunchKernelGGL(resize2x_h_kernel, dim3(shdr::stream_grid((long)N * H * W * (C / 8))), dim3(256), 0, S(stream), H_(x), HM_(y), N, H, W, C);
  return shdr::check_launch("resize2x_f16");
}
extern "C" int shdr_resize2x_bwd_f16(const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
  if (int rc = chk8("resize2x_bwd_f16", dy, dx, N, H, W, C)) return rc;
  hipLaunchKernelGGL(resize2x_bwd_h_kernel, dim3(shdr::stream_grid((long)N * H * W * (C / 8))), dim3(256), 0, S(stream), H_(dy), HM_(dx), N, H, W, C);
  return shdr::check_launch("resize2x_bwd_f16");
}
extern "C" int shdr_upsample_zero2_f16(const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
  if (int rc = chk8("upsample_zero2_f16", dy, dx, N, H, W, C)) return rc;
  hipLaunchKernelGGL(upsample_zero2_h_kernel, dim3(shdr::stream_grid((long)N * H * W * (C / 8))), dim3(256), 0, S(stream), H_(dy), HM_(dx), N, H, W, C);
  return shdr::check_launch("upsample_zero2_f16");
}
extern "C" int shdr_gap_fwd_f16(const void* x, float* y, int N, int HW, int C, void* stream) {
  if (int rc = chk8("gap_f16", x, y, N, HW, 1, C)) return rc;
  hipLaunchKernelGGL(gap_h_kernel, dim3((C + 127) / 128, N), dim3(256), 0, S(stream), H_(x), y, HW, C);
  return shdr::check_launch("gap_f16");
}
extern "C" int shdr_gap_bwd_f16(const float* dy, void* dx, int N, int HW, int C, void* stream) {
  if (int rc = chk8("gap_bwd_f16", dy, dx, N, HW, 1, C)) return rc;
  const long tv = (long)N * HW * (C / 8);
  hipLaunchKernelGGL(gap_bwd_h_kernel, dim3(shdr::stream_grid(tv)), dim3(256), 0, S(stream), dy, HM_(dx), tv, HW, C);
  return shdr::check_launch("gap_bwd_f16");
}
namespace {
inline void launch_bn_reduce_h(hipStream_t st, HP a, HP x, HP y, const float* mean, double* ws, long npix, int C, int mode) {
  const int O = C / 8;
  int OL = 1;
  while (OL < O && OL < 256) OL <<= 1;
  const long PL = 256 / OL;
  long g = (npix + PL * 16 - 1) / (PL * 16);
  g = g < 1 ? 1 : (g > SHDR_BN_MAX_BLOCKS ? SHDR_BN_MAX_BLOCKS : g);       // every block owns a partial row of the workspace
  hipLaunchKernelGGL(bn_reduce_h_kernel, dim3((unsigned)g), dim3(256), 0, st, a, x, y, mean, ws, npix, C, mode);
  hipLaunchKernelGGL(bn_fold_h_kernel, dim3((unsigned)((2 * C + 3) / 4)), dim3(256), 0, st, ws, C, (int)g);
}
}  // namespace
extern "C" int shdr_bn_stats_f16(const void* x, double* ws, float* mean, float* var, float* moving_mean, float* moving_var, int64_t npix,
                                 int C, float momentum, void* stream) {
  SHDR_REQUIRE(x && ws && mean && var, SHDR_E_NULL, "bn_stats_f16: null pointer");
  SHDR_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), SHDR_E_NULL, "bn_stats_f16: moving stats come in pairs");
  SHDR_REQUIRE(npix > 0 && C > 0 && C % 8 == 0, SHDR_E_SHAPE, "bn_stats_f16: bad shape (C %% 8 == 0)");
  hipStream_t st = S(stream);
  launch_bn_reduce_h(st, H_(x), nullptr, nullptr, nullptr, ws, (long)npix, C, 0);
  hipLaunchKernelGGL(bn_finalize_h_kernel, dim3((C + 255) / 256), dim3(256), 0, st, ws, mean, var, moving_mean, moving_var, (long)npix, C, momentum);
  return shdr::check_launch("bn_stats_f16");
}
extern "C" int shdr_bn_train_apply_f16(const void* x, const float* mean, const float* var, const float* gamma, const float* beta, void* y,
                                       int64_t npix, int C, float eps, int relu, void* stream) {
  SHDR_REQUIRE(x && mean && var && gamma && beta && y, SHDR_E_NULL, "bn_train_apply_f16: null pointer");
  SHDR_REQUIRE(npix > 0 && C > 0 && C % 8 == 0, SHDR_E_SHAPE, "bn_train_apply_f16: bad shape (C %% 8 == 0)");
  const long nvec = (long)npix * (C / 8);
  hipLaunchKernelGGL(bn_apply_h_kernel, dim3(octet_grid(nvec, C / 8, 2048)), dim3(256), 0, S(stream), H_(x), (HP) nullptr, (HP) nullptr, mean, var,
                     gamma, beta, (const double*)nullptr, HM_(y), nvec, (long)npix, C, eps, relu, 0);
  return shdr::check_launch("bn_train_apply_f16");
}
extern "C" int shdr_bn_bwd_f16(const void* dy, const void* x, const void* y_relu, const float* mean, const float* var, const float* gamma,
                               double* ws, float* dgamma, float* dbeta, void* dx, int64_t npix, int C, float eps, void* stream) {
  SHDR_REQUIRE(dy && x && mean && var && gamma && ws && dgamma && dbeta && dx, SHDR_E_NULL, "bn_bwd_f16: null pointer");
  SHDR_REQUIRE(npix > 0 && C > 0 && C % 8 == 0, SHDR_E_SHAPE, "bn_bwd_f16: bad shape (C %% 8 == 0)");
  hipStream_t st = S(stream);
  launch_bn_reduce_h(st, H_(dy), H_(x), H_(y_relu), mean, ws, (long)npix, C, 1);
  hipLaunchKernelGGL(bn_bwd_finalize_h_kernel, dim3((C + 255) / 256), dim3(256), 0, st, ws, var, dgamma, dbeta, C, eps);
  const long nvec = (long)npix * (C / 8);
  hipLaunchKernelGGL(bn_apply_h_kernel, dim3(octet_grid(nvec, C / 8, 2048)), dim3(256), 0, st, H_(dy), H_(x), H_(y_relu), mean, var, gamma,
                     (const float*)nullptr, ws, HM_(dx), nvec, (long)npix, C, eps, 0, 1);
  return shdr::check_launch("bn_bwd_f16");
}
extern "C" int shdr_lin_frontend_fwd_f16(const float* img, void* y, int N, int H, int W, int y_channels, void* stream) {
  SHDR_REQUIRE(img && y, SHDR_E_NULL, "lin_frontend_f16: null pointer");
  SHDR_REQUIRE(N > 0 && H >= 2 && W >= 2 && y_channels >= 93 && y_channels % 8 == 0, SHDR_E_SHAPE,
               "lin_frontend_f16: need H, W >= 2 and y_channels >= 93, a multiple of 8");
  SHDR_REQUIRE(shdr::aligned16(y), SHDR_E_ALIGN, "lin_frontend_f16: y must be 16-byte aligned");
  SHDR_REQUIRE(y_channels == 96, SHDR_E_SHAPE, "lin_frontend_f16: built for the 96-channel (93 + 3 zero) layout");
  // 6 blocks per group: one leading-octet block + five histogram blocks (5 * 256 threads = a multiple of the 10 histogram octets)
  long groups = ((long)N * H * W * 12 + 6 * 256 - 1) / (6 * 256);
  groups = groups < 1 ? 1 : (groups > 342 ? 342 : groups);
  hipLaunchKernelGGL(lin_frontend_h_kernel, dim3((unsigned)(6 * groups)), dim3(256), 0, S(stream), img, HM_(y), N, H, W, y_channels);
  return shdr::check_launch("lin_frontend_f16");
}
extern "C" int shdr_lin_frontend_bwd_f16(const float* img, const void* dF, float* dimg, int N, int H, int W, int y_channels, void* stream) {
  SHDR_REQUIRE(img && dF && dimg, SHDR_E_NULL, "lin_frontend_bwd_f16: null pointer");
  SHDR_REQUIRE(N > 0 && H >= 2 && W >= 2 && y_channels >= 93, SHDR_E_SHAPE, "lin_frontend_bwd_f16: bad shape");
  hipLaunchKernelGGL(lin_frontend_bwd_h_kernel, dim3(shdr::stream_grid((long)N * H * W * 3)), dim3(256), 0, S(stream), img, H_(dF), dimg, N, H, W,
                     y_channels);
  return shdr::check_launch("lin_frontend_bwd_f16");
}
