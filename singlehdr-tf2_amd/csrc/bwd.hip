// Backward / training kernels of the SingleHDR hot path on gfx950 (everything except the conv
// weight gradient, which lives in wgrad.hip).  All HBM-bound or tiny; NHWC fp32.
//
// Replaces the GradientTape.gradient / Adam.apply_gradients call sites of
// joint_training.py:185-186, train.py:175-176,195-196,242-243, finetune_real_dataset.py:177-178
// for the ops of SURVEY.md section 2.2 rows T3, T5-T8, T12-T18.
#include <stdlib.h>

#include "shdr_internal.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// block-wide sum (256 threads); result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sred) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  __syncthreads();
  return r;
}

// ---- activation backward: dx = dy * act'(.) evaluated from the activation OUTPUT y ------------
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dx, long n, int act) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float yv = y[i], g = dy[i];
    float d;
    switch (act) {
      case SHDR_ACT_RELU: d = yv > 0.f ? g : 0.f; break;
      case SHDR_ACT_LRELU: d = yv > 0.f ? g : 0.1f * g; break;   // y and the pre-activation share their sign
      case SHDR_ACT_TANH: d = g * (1.0f - yv * yv); break;
      default: d = g;
    }
    dx[i] = d;
  }
}

// Fused dz = dy * act'(y) and db[c] += sum_p dz[p][c]: one pass over dy / y instead of act_bwd + bias_grad (three tensor
// passes instead of four, float4 accesses).  C / 4 is a power of two <= 256, so a thread's channel quad is fixed over its
// grid-stride loop and the four sums stay in registers; one LDS tree + one atomic per block and channel.
// ws != null: every block writes its C sums to row blockIdx.x of ws instead (folded by shdr::col_fold_kernel) -- no atomics on the
// same C addresses, so the grid can be as large as the stream wants (3.7 -> 5 TB/s).
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           float* __restrict__ dz, float* __restrict__ db, float* __restrict__ ws,
                                                           long nquads, int Q, int act) {
  __shared__ float4 part[256];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  auto one = [&](long e, const float4 g, const float4 yv) {
    float4 d = g;
    if (act != SHDR_ACT_NONE) {
      if (act == SHDR_ACT_RELU) {
        d.x = yv.x > 0.f ? g.x : 0.f; d.y = yv.y > 0.f ? g.y : 0.f; d.z = yv.z > 0.f ? g.z : 0.f; d.w = yv.w > 0.f ? g.w : 0.f;
      } else if (act == SHDR_ACT_LRELU) {
        d.x = yv.x > 0.f ? g.x : 0.1f * g.x; d.y = yv.y > 0.f ? g.y : 0.1f * g.y;
        d.z = yv.z > 0.f ? g.z : 0.1f * g.z; d.w = yv.w > 0.f ? g.w : 0.1f * g.w;
      } else {
        d.x = g.x * (1.0f - yv.x * yv.x); d.y = g.y * (1.0f - yv.y * yv.y);
        d.z = g.z * (1.0f - yv.z * yv.z); d.w = g.w * (1.0f - yv.w * yv.w);
      }
      *reinterpret_cast<float4*>(dz + 4 * e) = d;
    }
    s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
  };
  // four quads per trip: the grid is capped (one atomic per block and channel on the same C addresses), so the bytes in
  // flight have to come from the loop
  const long step = (long)gridDim.x * 256;
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  for (; e + 3 * step < nquads; e += 4 * step) {
    float4 g[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      g[u] = *reinterpret_cast<const float4*>(dy + 4 * (e + u * step));
      yv[u] = g[u];
      if (act != SHDR_ACT_NONE) yv[u] = *reinterpret_cast<const float4*>(y + 4 * (e + u * step));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(e + u * step, g[u], yv[u]);
  }
  for (; e < nquads; e += step) {
    const float4 g0 = *reinterpret_cast<const float4*>(dy + 4 * e);
    float4 y0 = g0;
    if (act != SHDR_ACT_NONE) y0 = *reinterpret_cast<const float4*>(y + 4 * e);
    one(e, g0, y0);
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= Q; off >>= 1) {            // threads t and t + k*Q hold the same channel quad
    if ((int)threadIdx.x < off) {
      const float4 o = part[threadIdx.x + off];
      float4 m = part[threadIdx.x];
      m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
      part[threadIdx.x] = m;
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < Q) {
    const float4 m = part[threadIdx.x];
    if (ws) {
      *reinterpret_cast<float4*>(ws + (size_t)blockIdx.x * 4 * Q + 4 * threadIdx.x) = m;
    } else {
      float* o = db + 4 * threadIdx.x;
      atomicAdd(o, m.x); atomicAdd(o + 1, m.y); atomicAdd(o + 2, m.z); atomicAdd(o + 3, m.w);
    }
  }
}

// clip backward: gradient passes inside the closed interval [lo, hi] (tf.clip_by_value)
__global__ __launch_bounds__(256) void clip_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       float* __restrict__ dx, long n, float lo, float hi) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float xv = x[i];
    dx[i] = (xv >= lo && xv <= hi) ? dy[i] : 0.f;
  }
}

// yr: range slot of y or null (the sums of a backward pass -- forked tensors, residual joins -- feed split-operand input / weight
// gradients: tracking max |y| here saves their measuring pass)
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long n, unsigned* __restrict__ yr) {
  float ym = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = a[i] + b[i];
    y[i] = v;
    ym = fmaxf(ym, fabsf(v));
  }
  if (yr) shdr::range_out_block256(yr, ym);
}

// (e < 2^32 for every tensor below 64 GB: three 32-bit divisions -- about 25 instructions each -- instead of three emulated
//  64-bit ones, which cost more than the whole rest of these HBM-bound kernels; wider indices keep the 64-bit path)
#define SHDR_DECODE_QUAD(e, Q, Wd, Hd, q, w, h, n)                     \
  int q, w, h;                                                          \
  long n;                                                               \
  if ((unsigned long)(e) <= 0xffffffffUL) {                             \
    unsigned _t = (unsigned)(e);                                        \
    q = (int)(_t % (unsigned)(Q)); _t /= (unsigned)(Q);                 \
    w = (int)(_t % (unsigned)(Wd)); _t /= (unsigned)(Wd);               \
    h = (int)(_t % (unsigned)(Hd)); n = (long)(_t / (unsigned)(Hd));    \
  } else {                                                              \
    long _t = (e);                                                      \
    q = (int)(_t % (Q)); _t /= (Q);                                     \
    w = (int)(_t % (Wd)); _t /= (Wd);                                   \
    h = (int)(_t % (Hd)); n = _t / (Hd);                                \
  }

// ---- pooling / resize backward (gather form: one thread per INPUT quad) ------------------------
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                           int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, Q = C >> 2;
  const long total = (long)N * H * W * Q;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    SHDR_DECODE_QUAD(e, Q, W, H, q, w, h, n)
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((h >> 1) < Ho && (w >> 1) < Wo) {
      g = ld4(dy + (((n * Ho + (h >> 1)) * Wo + (w >> 1)) * (long)C + 4 * q));
      g.x *= 0.25f; g.y *= 0.25f; g.z *= 0.25f; g.w *= 0.25f;
    }
    st4(dx + e * 4, g);
  }
}

// MaxPool 2x2/2: the gradient goes to the FIRST maximum of the window in row-major scan order
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, Q = C >> 2;
  const long total = (long)N * Ho * Wo * Q;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    SHDR_DECODE_QUAD(e, Q, Wo, Ho, q, ow, oh, n)
    const long base = ((n * H + 2 * oh) * W + 2 * ow) * (long)C + 4 * q;
    const long off[4] = {0, (long)C, (long)W * C, (long)W * C + C};
    float v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 t = ld4(x + base + off[k]);
      v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
    }
    const float4 g4 = ld4(dy + e * 4);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    float o[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int arg = 0;
      float m = v[0][c];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][c] > m) { m = v[k][c]; arg = k; }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][c] = (k == arg) ? g[c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) st4(dx + base + off[k], make_float4(o[k][0], o[k][1], o[k][2], o[k][3]));
  }
}

// MaxPool 3x3/2 SAME (overlapping windows): one thread per input quad gathers from every window
// that contains it and whose first maximum it is.  With the pooled output at hand an element is a
// candidate only where it equals the window's maximum (one load per window), and only the elements
// BEFORE it in the window's scan order can still take the gradient from it (the tie rule).
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy, float* __restrict__ dx, int N, int H,
                                                             int W, int C, int Ho, int Wo, int pt, int pl) {
  const int Q = C >> 2;
  const long total = (long)N * H * W * Q;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    SHDR_DECODE_QUAD(e, Q, W, H, q, w, h, n)
    const float4 mine4 = ld4(x + e * 4);
    const float mine[4] = {mine4.x, mine4.y, mine4.z, mine4.w};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oh = max(0, (h + pt - 1) / 2); oh <= min(Ho - 1, (h + pt) / 2); ++oh) {
      for (int ow = max(0, (w + pl - 1) / 2); ow <= min(Wo - 1, (w + pl) / 2); ++ow) {
        const int h0 = 2 * oh - pt, w0 = 2 * ow - pl;
        if (h < h0 || h > h0 + 2 || w < w0 || w > w0 + 2) continue;
        const long widx = ((n * Ho + oh) * Wo + ow) * (long)C + 4 * q;
        const float4 m4 = ld4(y + widx);
        bool win[4] = {mine[0] == m4.x, mine[1] == m4.y, mine[2] == m4.z, mine[3] == m4.w};
        if (!(win[0] | win[1] | win[2] | win[3])) continue;
        for (int ih = max(h0, 0); ih <= h; ++ih) {
          const int wend = (ih == h) ? w : min(w0 + 3, W);
          for (int iw = max(w0, 0); iw < wend; ++iw) {
            const float4 o4 = ld4(x + (((n * H + ih) * W + iw) * (long)C + 4 * q));
            win[0] &= o4.x != mine[0]; win[1] &= o4.y != mine[1]; win[2] &= o4.z != mine[2]; win[3] &= o4.w != mine[3];
          }
        }
        if (!(win[0] | win[1] | win[2] | win[3])) continue;
        const float4 g4 = ld4(dy + widx);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (win[c]) acc[c] += g[c];
      }
    }
    st4(dx + e * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
  }
}

// bilinear 2x backward: per axis, input m receives {.25,.75,.75,.25} from outputs {2m-1,2m,2m+1,2m+2};
// the clamped borders fold the missing tap onto output 0 / 2H-1 (weight 1.0).
__device__ __forceinline__ void resize_taps(int m, int n_in, int* idx, float* wt, int* cnt) {
  int k = 0;
  if (m > 0) { idx[k] = 2 * m - 1; wt[k] = 0.25f; ++k; }
  idx[k] = 2 * m; wt[k] = (m == 0) ? 1.0f : 0.75f; ++k;
  idx[k] = 2 * m + 1; wt[k] = (m == n_in - 1) ? 1.0f : 0.75f; ++k;
  if (m < n_in - 1) { idx[k] = 2 * m + 2; wt[k] = 0.25f; ++k; }
  *cnt = k;
}
__global__ __launch_bounds__(256) void resize2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                           int N, int H, int W, int C, unsigned* __restrict__ xr) {
  const int Q = C >> 2, Ho = 2 * H, Wo = 2 * W;
  const long total = (long)N * H * W * Q;
  float dm = 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    SHDR_DECODE_QUAD(e, Q, W, H, q, w, h, n)
    int yi[4], xi[4], ny, nx;
    float yw[4], xw[4];
    resize_taps(h, H, yi, yw, &ny);
    resize_taps(w, W, xi, xw, &nx);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        const float4 g = ld4(dy + (((n * Ho + yi[a]) * Wo + xi[b]) * (long)C + 4 * q));
        const float wgt = yw[a] * xw[b];
        s.x += wgt * g.x; s.y += wgt * g.y; s.z += wgt * g.z; s.w += wgt * g.w;
      }
    st4(dx + e * 4, s);
    dm = fmaxf(fmaxf(fmaxf(fmaxf(dm, fabsf(s.x)), fabsf(s.y)), fabsf(s.z)), fabsf(s.w));
  }
  if (xr) shdr::range_out_block256(xr, dm);                    // (a sum of up to nine weighted taps: not bounded by max |dy|)
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                      long total_q, int HW, int C) {
  const int Q = C >> 2;
  const float inv = 1.0f / (float)HW;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total_q; e += (long)gridDim.x * 256) {
    const int q = (int)(e % Q);
    const long n = e / ((long)Q * HW);
    float4 g = ld4(dy + n * C + 4 * q);
    g.x *= inv; g.y *= inv; g.z *= inv; g.w *= inv;
    st4(dx + e * 4, g);
  }
}

// dx[n, 2*oh, 2*ow, :] = dy[n, oh, ow, :], zero elsewhere (input gradient of a 1x1 stride-2 conv
// after the 1x1 dgrad has been evaluated on the coarse grid)
__global__ __launch_bounds__(256) void upsample_zero2_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                             int N, int H, int W, int C) {
  const int Q = C >> 2, Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
  const long total = (long)N * H * W * Q;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    SHDR_DECODE_QUAD(e, Q, W, H, q, w, h, n)
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (((h | w) & 1) == 0) g = ld4(dy + (((n * Ho + (h >> 1)) * Wo + (w >> 1)) * (long)C + 4 * q));
    st4(dx + e * 4, g);
  }
}

// ---- BatchNormalization, training mode ----------------------------------------------------------
// per-channel double-precision partial sums: ws[c] += sum v1, ws[C + c] += sum v2
//   mode 0 (forward stats) : v1 = x,   v2 = x*x
//   mode 1 (backward)      : v1 = dy', v2 = dy' * (x - mean)   with dy' = dy masked by y > 0 when y != NULL
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                        const float* __restrict__ y, const float* __restrict__ mean,
                                                        double* __restrict__ ws, long npix, int C, int mode) {
  __shared__ double p1[256], p2[256];
  int CL = 1;
  while (CL < C && CL < 256) CL <<= 1;
  const int PL = 256 / CL;
  const int cl = threadIdx.x % CL, pl = threadIdx.x / CL;
  for (int c0 = 0; c0 < C; c0 += CL) {
    const int c = c0 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
      const float mu = mode ? mean[c] : 0.f;
      for (long p = (long)blockIdx.x * PL + pl; p < npix; p += (long)gridDim.x * PL) {
        const long i = p * C + c;
        if (mode == 0) {
          const double v = (double)a[i];
          s1 += v; s2 += v * v;
        } else {
          float g = a[i];
          if (y && !(y[i] > 0.f)) g = 0.f;
          s1 += (double)g; s2 += (double)g * (double)(x[i] - mu);
        }
      }
    }
    p1[threadIdx.x] = s1; p2[threadIdx.x] = s2;
    __syncthreads();
    if (pl == 0 && c < C) {
      for (int j = 1; j < PL; ++j) { s1 += p1[j * CL + cl]; s2 += p2[j * CL + cl]; }
      atomicAdd(ws + c, s1);
      atomicAdd(ws + C + c, s2);
    }
    __syncthreads();
  }
}

// The same sums with 16-byte accesses (C % 4 == 0): a thread owns one channel quad (QL = quads per pixel row handled side by
// side, a power of two <= 256) and walks the pixels two at a time, so two independent float4 loads per operand are in flight
// (the scalar kernel above runs at 2 TB/s, this one at the rate of a copy).  Double accumulators as above; fp32 only in
// the products of the loaded values.
__global__ __launch_bounds__(256) void bn_reduce4_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                         const float* __restrict__ y, const float* __restrict__ mean,
                                                         double* __restrict__ ws, long npix, int C, int mode) {
  __shared__ double part[8][256];
  const int Q = C >> 2;
  int QL = 1;
  while (QL < Q && QL < 256) QL <<= 1;
  const int PL = 256 / QL;
  const int ql = threadIdx.x % QL, pl = threadIdx.x / QL;
  for (int q0 = 0; q0 < Q; q0 += QL) {
    const int q = q0 + ql;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    if (q < Q) {
      float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mode) mu = *reinterpret_cast<const float4*>(mean + 4 * q);
      auto acc = [&](const float4 av, const float4 xv, const float4 yv) {
        const float g[4] = {av.x, av.y, av.z, av.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w}, yy[4] = {yv.x, yv.y, yv.z, yv.w};
        const float m[4] = {mu.x, mu.y, mu.z, mu.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (mode == 0) {
            const double v = (double)g[e];
            s1[e] += v; s2[e] += v * v;
          } else {
            const float gg = (y && !(yy[e] > 0.f)) ? 0.f : g[e];
            s1[e] += (double)gg; s2[e] += (double)gg * (double)(xx[e] - m[e]);
          }
        }
      };
      const long step = (long)gridDim.x * PL;
      long p = (long)blockIdx.x * PL + pl;
      const float4 z4 = make_float4(1.f, 1.f, 1.f, 1.f);
      for (; p + 3 * step < npix; p += 4 * step) {            // four independent loads per operand in flight
        float4 av[4], xv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long i = (p + u * step) * C + 4 * q;
          av[u] = *reinterpret_cast<const float4*>(a + i);
          xv[u] = z4; yv[u] = z4;
          if (mode) {
            xv[u] = *reinterpret_cast<const float4*>(x + i);
            if (y) yv[u] = *reinterpret_cast<const float4*>(y + i);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc(av[u], xv[u], yv[u]);
      }
      for (; p < npix; p += step) {
        const long i0 = p * C + 4 * q;
        const float4 a0 = *reinterpret_cast<const float4*>(a + i0);
        float4 x0 = z4, y0 = z4;
        if (mode) {
          x0 = *reinterpret_cast<const float4*>(x + i0);
          if (y) y0 = *reinterpret_cast<const float4*>(y + i0);
        }
        acc(a0, x0, y0);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { part[e][threadIdx.x] = s1[e]; part[4 + e][threadIdx.x] = s2[e]; }
    __syncthreads();
    if (pl == 0 && q < Q) {
      double* wp = ws + (size_t)(1 + blockIdx.x) * 2 * C;       // this block's partial row, summed by bn_fold_kernel: no atomics, no memset
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double t1 = s1[e], t2 = s2[e];
        for (int j = 1; j < PL; ++j) { t1 += part[e][j * QL + ql]; t2 += part[4 + e][j * QL + ql]; }
        wp[4 * q + e] = t1;
        wp[C + 4 * q + e] = t2;
      }
    }
    __syncthreads();
  }
}
// ws[col] = sum over the g partial rows ws[(1 + b) * 2C + col]: one wave per column, lanes stride over the rows
__global__ __launch_bounds__(256) void bn_fold_kernel(double* __restrict__ ws, int C, int g) {
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (col >= 2 * C) return;
  double t = 0.0;
  for (int b = lane; b < g; b += 64) t += ws[(size_t)(1 + b) * 2 * C + col];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  if (lane == 0) ws[col] = t;
}

// mean/var (biased) from the double sums; optional Keras moving-average update
// (momentum m: moving = moving*m + batch*(1-m); the variance update uses the unbiased estimate)
__global__ void bn_finalize_kernel(const double* __restrict__ ws, float* __restrict__ mean, float* __restrict__ var,
                                   float* __restrict__ mov_mean, float* __restrict__ mov_var, long npix, int C,
                                   float momentum) {
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

__global__ __launch_bounds__(256) void bn_train_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ var, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             long total, int C, float eps, int relu, unsigned* __restrict__ yr) {
  float ym = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    float v = (x[i] - mean[c]) * rsqrtf(var[c] + eps) * gamma[c] + beta[c];
    if (relu) v = fmaxf(v, 0.f);
    y[i] = v;
    ym = fmaxf(ym, fabsf(v));
  }
  if (yr) shdr::range_out_block256(yr, ym);                     // range slot of y: the consumer is usually a split-operand conv
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ ws, const float* __restrict__ var,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] += (float)ws[c];
  dgamma[c] += (float)(ws[C + c] * (double)rsqrtf(var[c] + eps));
}

// dx = gamma*invstd * (dy' - mean(dy') - xhat * mean(dy'*xhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ var, const float* __restrict__ gamma,
                                                           const double* __restrict__ ws, float* __restrict__ dx,
                                                           long total, long npix, int C, float eps, unsigned* __restrict__ xr) {
  float dm = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    float g = dy[i];
    if (y && !(y[i] > 0.f)) g = 0.f;
    const float invstd = rsqrtf(var[c] + eps);
    const float xh = (x[i] - mean[c]) * invstd;
    const float m1 = (float)(ws[c] / (double)npix);
    const float m2 = (float)(ws[C + c] / (double)npix) * invstd;   // mean(dy' * xhat)
    const float o = gamma[c] * invstd * (g - m1 - xh * m2);
    dx[i] = o;
    dm = fmaxf(dm, fabsf(o));
  }
  if (xr) shdr::range_out_block256(xr, dm);
}

// float4 form (C / 4 a power of two, grid * 256 a multiple of it): a thread's channel quad is fixed over its grid-stride
// loop, so the per-channel factors (two fp64 divisions, rsqrt) are computed once per thread instead of once per element
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ var, const float* __restrict__ gamma,
                                                            const double* __restrict__ ws, float* __restrict__ dx,
                                                            long nquads, long npix, int C, float eps, unsigned* __restrict__ xr) {
  const int Q = C >> 2;
  float dm = 0.f;
  const int q = (int)(((long)blockIdx.x * 256 + threadIdx.x) % Q);
  float invstd[4], mu[4], m1[4], m2[4], ga[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * q + e;
    invstd[e] = rsqrtf(var[c] + eps);
    mu[e] = mean[c];
    m1[e] = (float)(ws[c] / (double)npix);
    m2[e] = (float)(ws[C + c] / (double)npix) * invstd[e];
    ga[e] = gamma[c] * invstd[e];
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nquads; i += (long)gridDim.x * 256) {
    const float4 g4 = *reinterpret_cast<const float4*>(dy + 4 * i);
    const float4 x4 = *reinterpret_cast<const float4*>(x + 4 * i);
    float g[4] = {g4.x, g4.y, g4.z, g4.w};
    const float xx[4] = {x4.x, x4.y, x4.z, x4.w};
    if (y) {
      const float4 y4 = *reinterpret_cast<const float4*>(y + 4 * i);
      if (!(y4.x > 0.f)) g[0] = 0.f;
      if (!(y4.y > 0.f)) g[1] = 0.f;
      if (!(y4.z > 0.f)) g[2] = 0.f;
      if (!(y4.w > 0.f)) g[3] = 0.f;
    }
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xx[e] - mu[e]) * invstd[e];
      o[e] = ga[e] * (g[e] - m1[e] - xh * m2[e]);
    }
    *reinterpret_cast<float4*>(dx + 4 * i) = make_float4(o[0], o[1], o[2], o[3]);
    dm = fmaxf(fmaxf(fmaxf(fmaxf(dm, fabsf(o[0])), fabsf(o[1])), fabsf(o[2])), fabsf(o[3]));
  }
  if (xr) shdr::range_out_block256(xr, dm);
}

// ---- inverse-CRF head backward --------------------------------------------------------------------
constexpr int NPCA = 11;
// one block per batch row: dw11 = hinv^T dinv; dfeat = Wfc dw11; dWfc += feat (x) dw11; dbfc += dw11
__global__ __launch_bounds__(256) void invcrf_decode_bwd_kernel(const float* __restrict__ dinv, const float* __restrict__ feat,
                                                                const float* __restrict__ wfc, const float* __restrict__ table,
                                                                float* __restrict__ dfeat, float* __restrict__ dwfc,
                                                                float* __restrict__ dbfc, int F, int K) {
  __shared__ float red[4][NPCA];
  __shared__ float dwv[NPCA];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float part[NPCA];
#pragma unroll
  for (int j = 0; j < NPCA; ++j) part[j] = 0.f;
  for (int k = tid; k < K; k += 256) {
    const float g = dinv[(long)b * K + k];
    const float* row = table + (long)k * (NPCA + 1);
#pragma unroll
    for (int j = 0; j < NPCA; ++j) part[j] = fmaf(row[1 + j], g, part[j]);
  }
#pragma unroll
  for (int j = 0; j < NPCA; ++j) {
    const float s = wave_sum(part[j]);
    if (lane == 0) red[wave][j] = s;
  }
  __syncthreads();
  if (tid < NPCA) {
    const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    dwv[tid] = v;
    atomicAdd(dbfc + tid, v);
  }
  __syncthreads();
  for (int f = tid; f < F; f += 256) {
    const float xv = feat[(long)b * F + f];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NPCA; ++j) {
      s = fmaf(wfc[f * NPCA + j], dwv[j], s);
      atomicAdd(dwfc + f * NPCA + j, xv * dwv[j]);
    }
    dfeat[(long)b * F + f] = s;
  }
}

// _increase backward, one block per row (K <= 4096), sequential parts run by thread 0
__global__ __launch_bounds__(256) void increase_bwd_kernel(const float* __restrict__ rf, const float* __restrict__ dout,
                                                           float* __restrict__ drf, int K) {
  extern __shared__ float sm[];   // ng[G] | dn[G]
  __shared__ float sred[4];
  __shared__ float s_total, s_dot, s_dr, s_min;
  __shared__ int s_arg;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int G = K - 1;
  float* ng = sm;
  float* dn = sm + G;
  const float* r = rf + (long)b * K;
  const float* go = dout + (long)b * K;
  for (int k = tid; k < G; k += 256) ng[k] = r[k + 1] - r[k];
  __syncthreads();
  if (tid == 0) {          // first minimum (TF reduce_min gradient is spread evenly over ties; ties are measure-zero)
    float m = ng[0];
    int arg = 0;
    for (int k = 1; k < G; ++k)
      if (ng[k] < m) { m = ng[k]; arg = k; }
    s_min = m; s_arg = arg;
    float run = 0.f;       // dn[k] = sum_{j >= k} dout[j+1]  (reverse cumsum)
    for (int k = G - 1; k >= 0; --k) { run += go[k + 1]; dn[k] = run; }
  }
  __syncthreads();
  const float rr = fmaxf(-s_min, 0.f);
  float loc = 0.f, dot = 0.f;
  for (int k = tid; k < G; k += 256) {
    const float v = ng[k] + rr;
    ng[k] = v;
    loc += v;
  }
  const float total = block_sum(loc, sred);
  if (tid == 0) s_total = total;
  __syncthreads();
  for (int k = tid; k < G; k += 256) dot += dn[k] * ng[k];
  const float dsum = block_sum(dot, sred);
  if (tid == 0) s_dot = dsum;
  __syncthreads();
  const float S = s_total, D = s_dot;
  float dr = 0.f;
  for (int k = tid; k < G; k += 256) {
    const float d = dn[k] / S - D / (S * S);   // d loss / d ng[k]
    dn[k] = d;
    dr += d;
  }
  const float drs = block_sum(dr, sred);
  if (tid == 0) {
    s_dr = drs;
    if (s_min < 0.f) dn[s_arg] -= drs;          // r = -min(g): d r / d g[argmin] = -1
  }
  __syncthreads();
  (void)s_dr;
  float* o = drf + (long)b * K;
  for (int k = tid; k < K; k += 256) {
    float v = 0.f;
    if (k >= 1) v += dn[k - 1];
    if (k < G) v -= dn[k];
    o[k] = v;
  }
}

// apply_rf backward: d rf (LDS histogram, then global atomics) and optionally d x
__global__ __launch_bounds__(256) void apply_rf_bwd_kernel(const float* __restrict__ x, const float* __restrict__ rf,
                                                           const float* __restrict__ dy, float* __restrict__ drf,
                                                           float* __restrict__ dx, long n_per_batch, int K) {
  extern __shared__ float sm[];   // lut[K] | acc[K]
  float* lut = sm;
  float* acc = sm + K;
  const int b = blockIdx.y;
  for (int k = threadIdx.x; k < K; k += 256) { lut[k] = rf[(long)b * K + k]; acc[k] = 0.f; }
  __syncthreads();
  const float km1 = (float)(K - 1);
  const float* xb = x + (long)b * n_per_batch;
  const float* gb = dy + (long)b * n_per_batch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_per_batch; i += (long)gridDim.x * 256) {
    const float yv = km1 * xb[i];
    const float y0 = floorf(yv), y1 = y0 + 1.0f;
    const int i0 = min(max((int)y0, 0), K - 1), i1 = min(max((int)y1, 0), K - 1);
    const float g = gb[i];
    atomicAdd(acc + i0, g * (y1 - yv));
    atomicAdd(acc + i1, g * (yv - y0));
    if (dx) dx[(long)b * n_per_batch + i] = g * km1 * (lut[i1] - lut[i0]);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256)
    if (acc[k] != 0.f) atomicAdd(drf + (long)b * K + k, acc[k]);
}

// ---- losses -------------------------------------------------------------------------------------
// out[b] += mean over the sample of (a-b)^2 (mode 0) or |a-b| (mode 1); grid (blocks, B)
__global__ __launch_bounds__(256) void diff_loss_kernel(const float* __restrict__ a, const float* __restrict__ bb,
                                                        float* __restrict__ out, long n_per, int mode) {
  __shared__ float sred[4];
  const int b = blockIdx.y;
  const float* pa = a + (long)b * n_per;
  const float* pb = bb + (long)b * n_per;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_per; i += (long)gridDim.x * 256) {
    const float d = pa[i] - pb[i];
    s += mode ? fabsf(d) : d * d;
  }
  const float t = block_sum(s, sred);
  if (threadIdx.x == 0) atomicAdd(out + b, t / (float)n_per);
}

// da = g[b] * d/da mean-loss: 2(a-b)/n (mode 0) or sign(a-b)/n (mode 1); `accumulate` adds into da
__global__ __launch_bounds__(256) void diff_loss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ bb,
                                                            const float* __restrict__ g, float* __restrict__ da,
                                                            long n_per, int B, int mode, int accumulate) {
  const long total = n_per * B;
  const float inv = 1.0f / (float)n_per;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float d = a[i] - bb[i];
    const float gb = g[i / n_per] * inv;
    const float v = mode ? (d > 0.f ? gb : (d < 0.f ? -gb : 0.f)) : 2.0f * d * gb;
    da[i] = accumulate ? da[i] + v : v;
  }
}

// TV loss (joint_training.py:175-179): out[0] += mean|y[h+1]-y[h]| (over N*H*W*C incl. the zero last row)
//                                               + mean|y[w+1]-y[w]|
__global__ __launch_bounds__(256) void tv_loss_kernel(const float* __restrict__ y, float* __restrict__ out, int N, int H,
                                                      int W, int C) {
  __shared__ float sred[4];
  const long total = (long)N * H * W * C;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / C;
    const int w = (int)(p % W);
    const int h = (int)((p / W) % H);
    const float v = y[i];
    if (h + 1 < H) s += fabsf(y[i + (long)W * C] - v);
    if (w + 1 < W) s += fabsf(y[i + C] - v);
  }
  const float t = block_sum(s, sred);
  if (threadIdx.x == 0) atomicAdd(out, t / (float)total);
}

// dy += g * d tv / d y
__global__ __launch_bounds__(256) void tv_loss_bwd_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                          float* __restrict__ dy, int N, int H, int W, int C,
                                                          int accumulate) {
  const long total = (long)N * H * W * C;
  const float gs = g[0] / (float)total;
  auto sgn = [](float d) { return d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f); };
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / C;
    const int w = (int)(p % W);
    const int h = (int)((p / W) % H);
    const float v = y[i];
    float d = 0.f;
    if (h + 1 < H) d -= sgn(y[i + (long)W * C] - v);
    if (h > 0) d += sgn(v - y[i - (long)W * C]);
    if (w + 1 < W) d -= sgn(y[i + C] - v);
    if (w > 0) d += sgn(v - y[i - C]);
    dy[i] = accumulate ? dy[i] + gs * d : gs * d;
  }
}

// logc backward: dx = dy * 10 / ((1 + 10 x) ln 11)
__global__ __launch_bounds__(256) void logc_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       float* __restrict__ dx, long n) {
  const float k = 10.0f / logf(11.0f);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    dx[i] = dy[i] * k / (1.0f + 10.0f * x[i]);
}

// A = B + alpha * reverse3(hal), alpha constant:  d hal = reverse3(alpha * dA)
__global__ __launch_bounds__(256) void alpha_blend_bwd_kernel(const float* __restrict__ dA, const float* __restrict__ alpha,
                                                              float* __restrict__ dhal, long npix) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const float al = alpha[p];
    dhal[3 * p] = al * dA[3 * p + 2];
    dhal[3 * p + 1] = al * dA[3 * p + 1];
    dhal[3 * p + 2] = al * dA[3 * p];
  }
}

// alpha mask alone (joint_training.py:141-145): alpha[p] = clamp((max_c x - 1 + thr)/thr, 0, 1)
__global__ __launch_bounds__(256) void alpha_mask_kernel(const float* __restrict__ x, float* __restrict__ alpha, long npix,
                                                         float thr) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const float mx = fmaxf(fmaxf(x[3 * p], x[3 * p + 1]), x[3 * p + 2]);
    alpha[p] = fminf(1.0f, fmaxf(0.0f, mx - 1.0f + thr) / thr);
  }
}

// vgg_preprocess backward: d rgb = 255 * reverse3(d bgr[:3])
__global__ __launch_bounds__(256) void vgg_preprocess_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                 long npix, int ic) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    dx[3 * p] = 255.0f * dy[ic * p + 2];
    dx[3 * p + 1] = 255.0f * dy[ic * p + 1];
    dx[3 * p + 2] = 255.0f * dy[ic * p];
  }
}

// ---- Keras Adam (joint_training.py:186): theta -= lr_t * m / (sqrt(v) + eps) ---------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr_t, float b1, float b2,
                                                   float eps, float gscale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}

int nhwc4(const char* op, const void* a, const void* b, int N, int H, int W, int C) {
  SHDR_REQUIRE(a && b, SHDR_E_NULL, "%s: null pointer", op);
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, SHDR_E_SHAPE, "%s: non-positive dimension", op);
  SHDR_REQUIRE((C & 3) == 0, SHDR_E_ALIGN, "%s: C=%d must be a multiple of 4", op, C);
  SHDR_REQUIRE(shdr::aligned16(a) && shdr::aligned16(b), SHDR_E_ALIGN, "%s: tensors must be 16-byte aligned", op);
  return SHDR_OK;
}
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int reduce_grid(long npix, int C) {
  int CL = 1;
  while (CL < C && CL < 256) CL <<= 1;
  const long PL = 256 / CL;
  long g = (npix + PL * 64 - 1) / (PL * 64);
  return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}

inline void launch_bn_reduce(hipStream_t st, const float* a, const float* x, const float* y, const float* mean, double* ws,
                             long npix, int C, int mode) {
  const bool vec = C % 4 == 0 && shdr::aligned16(a) && (!x || shdr::aligned16(x)) && (!y || shdr::aligned16(y)) &&
                   (!mean || shdr::aligned16(mean)) && SHDR_ENV("SHDR_BN_SCALAR") == nullptr;
  if (vec) {
    const int Q = C / 4;
    int QL = 1;
    while (QL < Q && QL < 256) QL <<= 1;
    const long PL = 256 / QL;
    long g = (npix + PL * 16 - 1) / (PL * 16);          // >= 16 pixels per thread before the grid is capped
    // every block owns a partial row of the workspace (2 fp64 atomics per channel and block on the SAME addresses made 1024 blocks
    // 2x slower than 256 on a 134 MB tensor; the rows are summed by bn_fold_kernel)
    g = g < 1 ? 1 : (g > SHDR_BN_MAX_BLOCKS ? SHDR_BN_MAX_BLOCKS : g);
    hipLaunchKernelGGL(bn_reduce4_kernel, dim3((unsigned)g), dim3(256), 0, st, a, x, y, mean, ws, npix, C, mode);
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)((2 * C + 3) / 4)), dim3(256), 0, st, ws, C, (int)g);
  } else {
    (void)hipMemsetAsync(ws, 0, sizeof(double) * 2 * C, st);
    hipLaunchKernelGGL(bn_reduce_kernel, dim3(reduce_grid(npix, C)), dim3(256), 0, st, a, x, y, mean, ws, npix, C, mode);
  }
}

}  // namespace

extern "C" int shdr_act_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, int act, void* stream) {
  SHDR_REQUIRE(dy && y && dx, SHDR_E_NULL, "act_bwd: null pointer");
  SHDR_REQUIRE(n > 0 && act >= 0 && act <= 3, SHDR_E_SHAPE, "act_bwd: bad arguments");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(shdr::stream_grid(n)), dim3(256), 0, S(stream), dy, y, dx, (long)n, act);
  return shdr::check_launch("act_bwd");
}
extern "C" int shdr_act_bwd_bias_f32(const float* dy, const float* y, float* dz, float* db, float* ws, int64_t npix, int C, int act,
                                     void* stream) {
  SHDR_REQUIRE(dy && db, SHDR_E_NULL, "act_bwd_bias: null pointer");
  SHDR_REQUIRE(act == SHDR_ACT_NONE || (y && dz), SHDR_E_NULL, "act_bwd_bias: y and dz are needed with an activation");
  SHDR_REQUIRE(npix > 0 && act >= 0 && act <= 3, SHDR_E_SHAPE, "act_bwd_bias: bad arguments");
  const int Q = C / 4;
  SHDR_REQUIRE(C % 4 == 0 && Q >= 1 && Q <= 256 && (Q & (Q - 1)) == 0, SHDR_E_SHAPE,
               "act_bwd_bias: C / 4 must be a power of two <= 256 (got C = %d)", C);
  SHDR_REQUIRE(shdr::aligned16(dy) && (!y || shdr::aligned16(y)) && (!dz || shdr::aligned16(dz)), SHDR_E_ALIGN,
               "act_bwd_bias: tensors must be 16-byte aligned");
  const long nquads = (long)npix * Q;
  // <= 512 blocks: every block ends with one atomic per channel on the SAME C addresses (2048 blocks measured slower than
  // the unfused pair)
  int grid = shdr::stream_grid(nquads);
  if (ws && nquads >= (1L << 22)) {                   // partial rows + fold: the grid of a plain streaming kernel (pays from ~64 MB on)
    SHDR_REQUIRE(shdr::aligned16(ws), SHDR_E_ALIGN, "act_bwd_bias: ws must be 16-byte aligned");
    if (grid > shdr::kBiasMaxBlocks) grid = shdr::kBiasMaxBlocks;
    hipLaunchKernelGGL(act_bwd_bias_kernel, dim3(grid), dim3(256), 0, S(stream), dy, y, dz, db, ws, nquads, Q, act);
    shdr::launch_col_fold(ws, db, grid, C, S(stream));
    return shdr::check_launch("act_bwd_bias");
  }
  const int cap = nquads >= (1L << 24) ? 512 : (nquads >= (1L << 23) ? 384 : 256);     // measured per tensor size
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(act_bwd_bias_kernel, dim3(grid), dim3(256), 0, S(stream), dy, y, dz, db, (float*)nullptr, nquads, Q, act);
  return shdr::check_launch("act_bwd_bias");
}
extern "C" int shdr_clip_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, float lo, float hi, void* stream) {
  SHDR_REQUIRE(dy && x && dx, SHDR_E_NULL, "clip_bwd: null pointer");
  SHDR_REQUIRE(n > 0, SHDR_E_SHAPE, "clip_bwd: n must be positive");
  hipLaunchKernelGGL(clip_bwd_kernel, dim3(shdr::stream_grid(n)), dim3(256), 0, S(stream), dy, x, dx, (long)n, lo, hi);
  return shdr::check_launch("clip_bwd");
}
extern "C" int shdr_add_ranged_f32(const float* a, const float* b, float* y, int64_t n, float* y_range, void* stream) {
  SHDR_REQUIRE(a && b && y, SHDR_E_NULL, "add: null pointer");
  SHDR_REQUIRE(n > 0, SHDR_E_SHAPE, "add: n must be positive");
  hipLaunchKernelGGL(add_kernel, dim3(shdr::stream_grid(n)), dim3(256), 0, S(stream), a, b, y, (long)n, reinterpret_cast<unsigned*>(y_range));
  return shdr::check_launch("add");
}
extern "C" int shdr_add_f32(const float* a, const float* b, float* y, int64_t n, void* stream) {
  return shdr_add_ranged_f32(a, b, y, n, nullptr, stream);
}
extern "C" int shdr_avgpool2_bwd_f32(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
  if (int rc = nhwc4("avgpool2_bwd", dy, dx, N, H, W, C)) return rc;
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(shdr::stream_grid((long)N * H * W * (C / 4))), dim3(256), 0, S(stream),
                     dy, dx, N, H, W, C);
  return shdr::check_launch("avgpool2_bwd");
}
extern "C" int shdr_maxpool2_bwd_f32(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
  if (int rc = nhwc4("maxpool2_bwd", x, dx, N, H, W, C)) return rc;
  SHDR_REQUIRE(dy && shdr::aligned16(dy), SHDR_E_NULL, "maxpool2_bwd: dy null or unaligned");
  SHDR_REQUIRE((H & 1) == 0 && (W & 1) == 0, SHDR_E_SHAPE, "maxpool2_bwd: H, W must be even");
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(shdr::stream_grid((long)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     S(stream), x, dy, dx, N, H, W, C);
  return shdr::check_launch("maxpool2_bwd");
}
extern "C" int shdr_maxpool3s2_bwd_f32(const float* x, const float* y, const float* dy, float* dx, int N, int H, int W, int C,
                                       void* stream) {
  if (int rc = nhwc4("maxpool3s2_bwd", x, dx, N, H, W, C)) return rc;
  SHDR_REQUIRE(dy && shdr::aligned16(dy), SHDR_E_NULL, "maxpool3s2_bwd: dy null or unaligned");
  SHDR_REQUIRE(y && shdr::aligned16(y), SHDR_E_NULL, "maxpool3s2_bwd: y (the pooled output) null or unaligned");
  int Ho, Wo, pt, pl;
  shdr_same_pad(H, 3, 2, &Ho, &pt);
  shdr_same_pad(W, 3, 2, &Wo, &pl);
  hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(shdr::stream_grid((long)N * H * W * (C / 4))), dim3(256), 0, S(stream),
                     x, y, dy, dx, N, H, W, C, Ho, Wo, pt, pl);
  return shdr::check_launch("maxpool3s2_bwd");
}
extern "C" int shdr_resize2x_bwd_ranged_f32(const float* dy, float* dx, int N, int H, int W, int C, float* dx_range, void* stream) {
  if (int rc = nhwc4("resize2x_bwd", dy, dx, N, H, W, C)) return rc;
  hipLaunchKernelGGL(resize2x_bwd_kernel, dim3(shdr::stream_grid((long)N * H * W * (C / 4))), dim3(256), 0, S(stream),
                     dy, dx, N, H, W, C, reinterpret_cast<unsigned*>(dx_range));
  return shdr::check_launch("resize2x_bwd");
}
extern "C" int shdr_resize2x_bwd_f32(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
  return shdr_resize2x_bwd_ranged_f32(dy, dx, N, H, W, C, nullptr, stream);
}
extern "C" int shdr_gap_bwd_f32(const float* dy, float* dx, int N, int HW, int C, void* stream) {
  if (int rc = nhwc4("gap_bwd", dy, dx, N, HW, 1, C)) return rc;
  const long tq = (long)N * HW * (C / 4);
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(shdr::stream_grid(tq)), dim3(256), 0, S(stream), dy, dx, tq, HW, C);
  return shdr::check_launch("gap_bwd");
}
extern "C" int shdr_upsample_zero2_f32(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
  if (int rc = nhwc4("upsample_zero2", dy, dx, N, H, W, C)) return rc;
  hipLaunchKernelGGL(upsample_zero2_kernel, dim3(shdr::stream_grid((long)N * H * W * (C / 4))), dim3(256), 0, S(stream),
                     dy, dx, N, H, W, C);
  return shdr::check_launch("upsample_zero2");
}

extern "C" int shdr_bn_stats_f32(const float* x, double* ws, float* mean, float* var, float* moving_mean,
                                 float* moving_var, int64_t npix, int C, float momentum, void* stream) {
  SHDR_REQUIRE(x && ws && mean && var, SHDR_E_NULL, "bn_stats: null pointer");
  SHDR_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), SHDR_E_NULL, "bn_stats: moving stats come in pairs");
  SHDR_REQUIRE(npix > 0 && C > 0, SHDR_E_SHAPE, "bn_stats: bad shape");
  hipStream_t st = S(stream);
  launch_bn_reduce(st, x, nullptr, nullptr, nullptr, ws, (long)npix, C, 0);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, ws, mean, var, moving_mean, moving_var,
                     (long)npix, C, momentum);
  return shdr::check_launch("bn_stats");
}
extern "C" int shdr_bn_train_apply_ranged_f32(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                                              float* y, int64_t npix, int C, float eps, int relu, float* y_range, void* stream) {
  SHDR_REQUIRE(x && mean && var && gamma && beta && y, SHDR_E_NULL, "bn_train_apply: null pointer");
  SHDR_REQUIRE(npix > 0 && C > 0, SHDR_E_SHAPE, "bn_train_apply: bad shape");
  hipLaunchKernelGGL(bn_train_apply_kernel, dim3(shdr::stream_grid(npix * C)), dim3(256), 0, S(stream), x, mean, var,
                     gamma, beta, y, (long)npix * C, C, eps, relu, reinterpret_cast<unsigned*>(y_range));
  return shdr::check_launch("bn_train_apply");
}
extern "C" int shdr_bn_train_apply_f32(const float* x, const float* mean, const float* var, const float* gamma,
                                       const float* beta, float* y, int64_t npix, int C, float eps, int relu, void* stream) {
  return shdr_bn_train_apply_ranged_f32(x, mean, var, gamma, beta, y, npix, C, eps, relu, nullptr, stream);
}
extern "C" int shdr_bn_bwd_ranged_f32(const float* dy, const float* x, const float* y_relu, const float* mean, const float* var,
                                      const float* gamma, double* ws, float* dgamma, float* dbeta, float* dx, int64_t npix,
                                      int C, float eps, float* dx_range, void* stream) {
  SHDR_REQUIRE(dy && x && mean && var && gamma && ws && dgamma && dbeta && dx, SHDR_E_NULL, "bn_bwd: null pointer");
  SHDR_REQUIRE(npix > 0 && C > 0, SHDR_E_SHAPE, "bn_bwd: bad shape");
  hipStream_t st = S(stream);
  launch_bn_reduce(st, dy, x, y_relu, mean, ws, (long)npix, C, 1);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, ws, var, dgamma, dbeta, C, eps);
  const int Q = C / 4;
  if (C % 4 == 0 && (Q & (Q - 1)) == 0 && Q <= 4096 && shdr::aligned16(dy) && shdr::aligned16(x) && shdr::aligned16(dx) &&
      (!y_relu || shdr::aligned16(y_relu)) && SHDR_ENV("SHDR_BN_SCALAR") == nullptr) {
    long grid = shdr::stream_grid(npix * Q);
    const long unit = Q > 256 ? Q / 256 : 1;              // grid * 256 must be a multiple of Q
    grid = (grid + unit - 1) / unit * unit;
    hipLaunchKernelGGL(bn_bwd_apply4_kernel, dim3((unsigned)grid), dim3(256), 0, st, dy, x, y_relu, mean, var, gamma, ws, dx,
                       (long)npix * Q, (long)npix, C, eps, reinterpret_cast<unsigned*>(dx_range));
  } else {
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(shdr::stream_grid(npix * C)), dim3(256), 0, st, dy, x, y_relu, mean, var,
                       gamma, ws, dx, (long)npix * C, (long)npix, C, eps, reinterpret_cast<unsigned*>(dx_range));
  }
  return shdr::check_launch("bn_bwd");
}
extern "C" int shdr_bn_bwd_f32(const float* dy, const float* x, const float* y_relu, const float* mean, const float* var,
                               const float* gamma, double* ws, float* dgamma, float* dbeta, float* dx, int64_t npix,
                               int C, float eps, void* stream) {
  return shdr_bn_bwd_ranged_f32(dy, x, y_relu, mean, var, gamma, ws, dgamma, dbeta, dx, npix, C, eps, nullptr, stream);
}

extern "C" int shdr_invcrf_decode_bwd_f32(const float* dinv, const float* feat, const float* wfc, const float* table,
                                          float* dfeat, float* dwfc, float* dbfc, int B, int F, int K, void* stream) {
  SHDR_REQUIRE(dinv && feat && wfc && table && dfeat && dwfc && dbfc, SHDR_E_NULL, "invcrf_decode_bwd: null pointer");
  SHDR_REQUIRE(B > 0 && F > 0 && K > 0, SHDR_E_SHAPE, "invcrf_decode_bwd: bad shape");
  hipLaunchKernelGGL(invcrf_decode_bwd_kernel, dim3(B), dim3(256), 0, S(stream), dinv, feat, wfc, table, dfeat, dwfc,
                     dbfc, F, K);
  return shdr::check_launch("invcrf_decode_bwd");
}
extern "C" int shdr_increase_bwd_f32(const float* rf, const float* dout, float* drf, int B, int K, void* stream) {
  SHDR_REQUIRE(rf && dout && drf, SHDR_E_NULL, "increase_bwd: null pointer");
  SHDR_REQUIRE(B > 0 && K >= 2 && K <= 4096, SHDR_E_SHAPE, "increase_bwd: need 2 <= K <= 4096");
  hipLaunchKernelGGL(increase_bwd_kernel, dim3(B), dim3(256), (size_t)2 * (K - 1) * sizeof(float), S(stream), rf, dout,
                     drf, K);
  return shdr::check_launch("increase_bwd");
}
extern "C" int shdr_apply_rf_bwd_f32(const float* x, const float* rf, const float* dy, float* drf, float* dx, int B,
                                     int64_t n_per_batch, int K, void* stream) {
  SHDR_REQUIRE(x && rf && dy && drf, SHDR_E_NULL, "apply_rf_bwd: null pointer");
  SHDR_REQUIRE(B > 0 && B <= 65535 && n_per_batch > 0 && K >= 2 && K <= 8192, SHDR_E_SHAPE, "apply_rf_bwd: bad shape");
  int gx = shdr::stream_grid(n_per_batch);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(apply_rf_bwd_kernel, dim3(gx, B), dim3(256), (size_t)2 * K * sizeof(float), S(stream), x, rf, dy,
                     drf, dx, (long)n_per_batch, K);
  return shdr::check_launch("apply_rf_bwd");
}

extern "C" int shdr_diff_loss_f32(const float* a, const float* b, float* out, int B, int64_t n_per_sample, int mode, void* stream) {
  SHDR_REQUIRE(a && b && out, SHDR_E_NULL, "diff_loss: null pointer");
  SHDR_REQUIRE(B > 0 && B <= 65535 && n_per_sample > 0 && (mode == 0 || mode == 1), SHDR_E_SHAPE, "diff_loss: bad arguments");
  hipStream_t st = S(stream);
  if (hipMemsetAsync(out, 0, sizeof(float) * B, st) != hipSuccess) return shdr::fail(SHDR_E_LAUNCH, "diff_loss: memset");
  int gx = shdr::stream_grid(n_per_sample);
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(diff_loss_kernel, dim3(gx, B), dim3(256), 0, st, a, b, out, (long)n_per_sample, mode);
  return shdr::check_launch("diff_loss");
}
extern "C" int shdr_diff_loss_bwd_f32(const float* a, const float* b, const float* g, float* da, int B,
                                      int64_t n_per_sample, int mode, int accumulate, void* stream) {
  SHDR_REQUIRE(a && b && g && da, SHDR_E_NULL, "diff_loss_bwd: null pointer");
  SHDR_REQUIRE(B > 0 && n_per_sample > 0 && (mode == 0 || mode == 1), SHDR_E_SHAPE, "diff_loss_bwd: bad arguments");
  hipLaunchKernelGGL(diff_loss_bwd_kernel, dim3(shdr::stream_grid(n_per_sample * B)), dim3(256), 0, S(stream), a, b, g,
                     da, (long)n_per_sample, B, mode, accumulate);
  return shdr::check_launch("diff_loss_bwd");
}
extern "C" int shdr_tv_loss_f32(const float* y, float* out, int N, int H, int W, int C, void* stream) {
  SHDR_REQUIRE(y && out, SHDR_E_NULL, "tv_loss: null pointer");
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, SHDR_E_SHAPE, "tv_loss: bad shape");
  hipStream_t st = S(stream);
  if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) return shdr::fail(SHDR_E_LAUNCH, "tv_loss: memset");
  int gx = shdr::stream_grid((long)N * H * W * C);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(tv_loss_kernel, dim3(gx), dim3(256), 0, st, y, out, N, H, W, C);
  return shdr::check_launch("tv_loss");
}
extern "C" int shdr_tv_loss_bwd_f32(const float* y, const float* g, float* dy, int N, int H, int W, int C, int accumulate, void* stream) {
  SHDR_REQUIRE(y && g && dy, SHDR_E_NULL, "tv_loss_bwd: null pointer");
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, SHDR_E_SHAPE, "tv_loss_bwd: bad shape");
  hipLaunchKernelGGL(tv_loss_bwd_kernel, dim3(shdr::stream_grid((long)N * H * W * C)), dim3(256), 0, S(stream), y, g, dy,
                     N, H, W, C, accumulate);
  return shdr::check_launch("tv_loss_bwd");
}
extern "C" int shdr_logc_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
  SHDR_REQUIRE(dy && x && dx, SHDR_E_NULL, "logc_bwd: null pointer");
  SHDR_REQUIRE(n > 0, SHDR_E_SHAPE, "logc_bwd: n must be positive");
  hipLaunchKernelGGL(logc_bwd_kernel, dim3(shdr::stream_grid(n)), dim3(256), 0, S(stream), dy, x, dx, (long)n);
  return shdr::check_launch("logc_bwd");
}
extern "C" int shdr_alpha_mask_f32(const float* x, float* alpha, int64_t npix, float thr, void* stream) {
  SHDR_REQUIRE(x && alpha, SHDR_E_NULL, "alpha_mask: null pointer");
  SHDR_REQUIRE(npix > 0 && thr > 0.f, SHDR_E_SHAPE, "alpha_mask: bad arguments");
  hipLaunchKernelGGL(alpha_mask_kernel, dim3(shdr::stream_grid(npix)), dim3(256), 0, S(stream), x, alpha, (long)npix, thr);
  return shdr::check_launch("alpha_mask");
}
extern "C" int shdr_alpha_blend_bwd_f32(const float* dA, const float* alpha, float* dhal, int64_t npix, void* stream) {
  SHDR_REQUIRE(dA && alpha && dhal, SHDR_E_NULL, "alpha_blend_bwd: null pointer");
  SHDR_REQUIRE(npix > 0, SHDR_E_SHAPE, "alpha_blend_bwd: npix must be positive");
  hipLaunchKernelGGL(alpha_blend_bwd_kernel, dim3(shdr::stream_grid(npix)), dim3(256), 0, S(stream), dA, alpha, dhal, (long)npix);
  return shdr::check_launch("alpha_blend_bwd");
}
extern "C" int shdr_vgg_preprocess_bwd_f32(const float* dy, float* dx, int64_t npix, int in_channels, void* stream) {
  SHDR_REQUIRE(dy && dx, SHDR_E_NULL, "vgg_preprocess_bwd: null pointer");
  SHDR_REQUIRE(npix > 0 && (in_channels == 3 || in_channels == 4), SHDR_E_SHAPE, "vgg_preprocess_bwd: bad arguments");
  hipLaunchKernelGGL(vgg_preprocess_bwd_kernel, dim3(shdr::stream_grid(npix)), dim3(256), 0, S(stream), dy, dx, (long)npix, in_channels);
  return shdr::check_launch("vgg_preprocess_bwd");
}
extern "C" int shdr_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                             float eps, float grad_scale, void* stream) {
  SHDR_REQUIRE(p && g && m && v, SHDR_E_NULL, "adam: null pointer");
  SHDR_REQUIRE(n > 0, SHDR_E_SHAPE, "adam: n must be positive");
  hipLaunchKernelGGL(adam_kernel, dim3(shdr::stream_grid(n)), dim3(256), 0, S(stream), p, g, m, v, (long)n, lr_t, beta1,
                     beta2, eps, grad_scale);
  return shdr::check_launch("adam");
}
