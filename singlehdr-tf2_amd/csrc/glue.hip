// Elementwise glue arithmetic of the step closures (gfx950, HBM-bound).
// Replaces the tf.* elementwise call sites cited per entry point in include/shdr.h.
#include "shdr_internal.h"

namespace {

constexpr float kVggMean0 = 103.939f, kVggMean1 = 116.779f, kVggMean2 = 123.68f;

__global__ __launch_bounds__(256) void clip_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                   long n, float lo, float hi) {
  const long nq = n >> 2;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
    float4 v = *reinterpret_cast<const float4*>(x + 4 * q);
    v.x = fminf(fmaxf(v.x, lo), hi); v.y = fminf(fmaxf(v.y, lo), hi);
    v.z = fminf(fmaxf(v.z, lo), hi); v.w = fminf(fmaxf(v.w, lo), hi);
    *reinterpret_cast<float4*>(y + 4 * q) = v;
  }
  for (long i = (nq << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = fminf(fmaxf(x[i], lo), hi);
}

__global__ __launch_bounds__(256) void logc_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  const float inv = 1.0f / logf(11.0f);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = logf(1.0f + 10.0f * x[i]) * inv;
}

// 3-channel pixels: one thread per pixel.
__global__ __launch_bounds__(256) void vgg_preprocess_kernel(const float* __restrict__ x,
                                                             float* __restrict__ y, long npix, int oc) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const float r = x[3 * p], g = x[3 * p + 1], b = x[3 * p + 2];
    const float v0 = b * 255.0f - kVggMean0, v1 = g * 255.0f - kVggMean1, v2 = r * 255.0f - kVggMean2;
    if (oc == 4) {
      *reinterpret_cast<float4*>(y + 4 * p) = make_float4(v0, v1, v2, 0.0f);
    } else {
      y[3 * p] = v0; y[3 * p + 1] = v1; y[3 * p + 2] = v2;
    }
  }
}

__global__ __launch_bounds__(256) void reverse3_kernel(const float* __restrict__ x, float* __restrict__ y, long npix) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const float a = x[3 * p], b = x[3 * p + 1], c = x[3 * p + 2];
    y[3 * p] = c; y[3 * p + 1] = b; y[3 * p + 2] = a;
  }
}

__global__ __launch_bounds__(256) void alpha_blend_kernel(const float* __restrict__ bp,
                                                          const float* __restrict__ hal,
                                                          float* __restrict__ a, float* __restrict__ alpha_out,
                                                          long npix, float thr) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const float b0 = bp[3 * p], b1 = bp[3 * p + 1], b2 = bp[3 * p + 2];
    const float mx = fmaxf(fmaxf(b0, b1), b2);
    const float al = fminf(1.0f, fmaxf(0.0f, mx - 1.0f + thr) / thr);
    a[3 * p] = b0 + al * hal[3 * p + 2];
    a[3 * p + 1] = b1 + al * hal[3 * p + 1];
    a[3 * p + 2] = b2 + al * hal[3 * p];
    if (alpha_out) alpha_out[p] = al;
  }
}

struct Pack3Args { const float* s[4]; };

__global__ __launch_bounds__(256) void pack3_kernel(Pack3Args src, int nsrc, float* __restrict__ y,
                                                    int OC, long npix) {
  const long total = npix * OC;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long p = e / OC;
    const int ch = (int)(e - p * OC);
    const int s = ch / 3, c = ch - 3 * s;
    float v = 0.f;
    if (s < nsrc) {
      const float* sp = s == 0 ? src.s[0] : (s == 1 ? src.s[1] : (s == 2 ? src.s[2] : src.s[3]));
      v = sp[3 * p + c];
    }
    y[e] = v;
  }
}

// y[p][c] = c < Cin ? x[p][c] : 0, c < Cout (channel zero-padding so that narrow tensors fit the
// 16-byte DMA quads / 16-channel MFMA tiles)
__global__ __launch_bounds__(256) void pad_channels_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           long npix, int Cin, int Cout) {
  const long total = npix * Cout;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long p = e / Cout;
    const int c = (int)(e - p * Cout);
    y[e] = c < Cin ? x[p * Cin + c] : 0.f;
  }
}

// y = act(x * scale[c] + shift[c] + residual): the folded-BatchNorm / residual / second-activation epilogue of the inference
// convs as a stand-alone op -- used when such a layer runs under a gradient tape (frozen BatchNorm statistics), where the
// conv, the affine map and the join have to be separate tape entries.  C % 4 == 0, float4 per thread.
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ res,
                                                         float* __restrict__ y, long nquads, int cq, int act) {
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nquads; q += (long)gridDim.x * 256) {
    const int c = 4 * (int)(q % cq);
    float4 v = *reinterpret_cast<const float4*>(x + 4 * q);
    if (scale) {
      const float4 s4 = *reinterpret_cast<const float4*>(scale + c);
      v.x *= s4.x; v.y *= s4.y; v.z *= s4.z; v.w *= s4.w;
    }
    if (shift) {
      const float4 t4 = *reinterpret_cast<const float4*>(shift + c);
      v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
    }
    if (res) {
      const float4 r4 = *reinterpret_cast<const float4*>(res + 4 * q);
      v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
    }
    v.x = shdr::act_apply(v.x, act); v.y = shdr::act_apply(v.y, act);
    v.z = shdr::act_apply(v.z, act); v.w = shdr::act_apply(v.w, act);
    *reinterpret_cast<float4*>(y + 4 * q) = v;
  }
}

// any channel count (the 3-channel heads): one element per thread
__global__ __launch_bounds__(256) void affine_act_scalar_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ res,
                                                                float* __restrict__ y, long n, int C, int act) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    float v = x[e];
    if (scale) v *= scale[c];
    if (shift) v += shift[c];
    if (res) v += res[e];
    y[e] = shdr::act_apply(v, act);
  }
}

}  // namespace

extern "C" int shdr_affine_act_f32(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                                   int64_t npix, int C, int act, void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "affine_act: null pointer");
  SHDR_REQUIRE(npix > 0 && C > 0, SHDR_E_SHAPE, "affine_act: need npix > 0 and C > 0");
  const bool vec = C % 4 == 0 && shdr::aligned16(x) && shdr::aligned16(y) && (!scale || shdr::aligned16(scale)) &&
                   (!shift || shdr::aligned16(shift)) && (!residual || shdr::aligned16(residual));
  if (!vec) {
    hipLaunchKernelGGL(affine_act_scalar_kernel, dim3(shdr::stream_grid((long)npix * C)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, scale, shift, residual, y, (long)npix * C, C, act);
    return shdr::check_launch("affine_act");
  }
  const long nquads = (long)npix * (C / 4);
  hipLaunchKernelGGL(affine_act_kernel, dim3(shdr::stream_grid(nquads)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     scale, shift, residual, y, nquads, C / 4, act);
  return shdr::check_launch("affine_act");
}

extern "C" int shdr_pad_channels_f32(const float* x, float* y, int64_t npix, int Cin, int Cout, void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "pad_channels: null pointer");
  SHDR_REQUIRE(npix > 0 && Cin > 0 && Cout >= Cin, SHDR_E_SHAPE, "pad_channels: need Cout >= Cin > 0");
  hipLaunchKernelGGL(pad_channels_kernel, dim3(shdr::stream_grid(npix * Cout)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, (long)npix, Cin, Cout);
  return shdr::check_launch("pad_channels");
}

extern "C" int shdr_clip_fwd_f32(const float* x, float* y, int64_t n, float lo, float hi, void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "clip: null pointer");
  SHDR_REQUIRE(n >= 0, SHDR_E_SHAPE, "clip: negative size");
  SHDR_REQUIRE(shdr::aligned16(x) && shdr::aligned16(y), SHDR_E_ALIGN, "clip: tensors must be 16-byte aligned");
  if (n == 0) return SHDR_OK;
  hipLaunchKernelGGL(clip_kernel, dim3(shdr::stream_grid((n + 3) / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, (long)n, lo, hi);
  return shdr::check_launch("clip");
}

extern "C" int shdr_logc_fwd_f32(const float* x, float* y, int64_t n, void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "logc: null pointer");
  SHDR_REQUIRE(n >= 0, SHDR_E_SHAPE, "logc: negative size");
  if (n == 0) return SHDR_OK;
  hipLaunchKernelGGL(logc_kernel, dim3(shdr::stream_grid(n)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, (long)n);
  return shdr::check_launch("logc");
}

extern "C" int shdr_vgg_preprocess_fwd_f32(const float* x, float* y, int64_t npix, int out_channels,
                                           void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "vgg_preprocess: null pointer");
  SHDR_REQUIRE(npix > 0, SHDR_E_SHAPE, "vgg_preprocess: npix must be positive");
  SHDR_REQUIRE(out_channels == 3 || out_channels == 4, SHDR_E_SHAPE, "vgg_preprocess: out_channels must be 3 or 4");
  SHDR_REQUIRE(out_channels == 3 || shdr::aligned16(y), SHDR_E_ALIGN, "vgg_preprocess: y not 16-byte aligned");
  hipLaunchKernelGGL(vgg_preprocess_kernel, dim3(shdr::stream_grid(npix)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, (long)npix, out_channels);
  return shdr::check_launch("vgg_preprocess");
}

extern "C" int shdr_reverse3_fwd_f32(const float* x, float* y, int64_t npix, void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "reverse3: null pointer");
  SHDR_REQUIRE(npix > 0, SHDR_E_SHAPE, "reverse3: npix must be positive");
  SHDR_REQUIRE(x != y, SHDR_E_SHAPE, "reverse3: in-place not supported");
  hipLaunchKernelGGL(reverse3_kernel, dim3(shdr::stream_grid(npix)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, (long)npix);
  return shdr::check_launch("reverse3");
}

extern "C" int shdr_alpha_blend_fwd_f32(const float* b, const float* hal, float* a, float* alpha_out,
                                        int64_t npix, float thr, void* stream) {
  SHDR_REQUIRE(b && hal && a, SHDR_E_NULL, "alpha_blend: null pointer");
  SHDR_REQUIRE(npix > 0 && thr > 0.f, SHDR_E_SHAPE, "alpha_blend: npix and thr must be positive");
  hipLaunchKernelGGL(alpha_blend_kernel, dim3(shdr::stream_grid(npix)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), b, hal, a, alpha_out, (long)npix, thr);
  return shdr::check_launch("alpha_blend");
}

extern "C" int shdr_pack3_fwd_f32(const float* s0, const float* s1, const float* s2, const float* s3,
                                  int nsrc, float* y, int out_channels, int64_t npix, void* stream) {
  SHDR_REQUIRE(y && s0, SHDR_E_NULL, "pack3: null pointer");
  SHDR_REQUIRE(nsrc >= 1 && nsrc <= 4 && out_channels >= 3 * nsrc && npix > 0, SHDR_E_SHAPE,
               "pack3: need 1<=nsrc<=4, out_channels>=3*nsrc");
  const float* s[4] = {s0, s1, s2, s3};
  for (int i = 0; i < nsrc; ++i) SHDR_REQUIRE(s[i], SHDR_E_NULL, "pack3: source %d is null", i);
  Pack3Args pa{{s0, s1, s2, s3}};
  hipLaunchKernelGGL(pack3_kernel, dim3(shdr::stream_grid(npix * out_channels)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), pa, nsrc, y, out_channels, (long)npix);
  return shdr::check_launch("pack3");
}
