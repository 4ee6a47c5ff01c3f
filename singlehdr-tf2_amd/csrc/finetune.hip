// Extra kernels of the chained fine-tuning step (finetune_real_dataset.py:144-183), where every
// net is fed the previous net's prediction, so gradients also flow through the Linearization
// front end, the alpha mask, the Refinement-Net input concat and the mean normalisation.
#include "shdr_internal.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float block_sum(float v, float* sred) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// d soft_bin / d x  (linearization_net.py:340-347): -B*sign(x - centre) inside the support, else 0
__device__ __forceinline__ float soft_bin_grad(float x, int i, int B) {
  const float centre = (float)(2 * i - 1) / (float)(2 * B);
  const float t = x - centre;
  const float d = fabsf(t);
  if (!(d < 1.0f / (float)B)) return 0.f;
  return t > 0.f ? -(float)B : (t < 0.f ? (float)B : 0.f);
}

// Backward of the fused front end: dimg (zeroed by the caller) += J^T dF.
// one thread per pixel; identity + histogram terms are local, the REFLECT-padded sobel stencil is
// scattered with fp32 atomics (36 per pixel on a 3-channel image).  They, not the reads, set the kernel's time: 1.8 ms at
// 4 x 1024^2; a variant that stages the dF rows through LDS with coalesced float4 loads measured 2.05 ms.  A gather form
// (transposed REFLECT stencil) is the way to the 0.4 ms of HBM time.
__global__ __launch_bounds__(256) void lin_frontend_bwd_kernel(const float* __restrict__ img, const float* __restrict__ dF,
                                                               float* __restrict__ dimg, int N, int H, int W, int YC) {
  const long npix = (long)N * H * W;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const int w = (int)(p % W);
    const long t = p / W;
    const int h = (int)(t % H);
    const long ibase = (t / H) * (long)H * W;
    const float* g = dF + p * YC;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = img[p * 3 + c];
      float acc = g[c];
      for (int b = 0; b < 4; ++b) acc += g[9 + 3 * b + c] * soft_bin_grad(x, b + 1, 4);
      for (int b = 0; b < 8; ++b) acc += g[21 + 3 * b + c] * soft_bin_grad(x, b + 1, 8);
      for (int b = 0; b < 16; ++b) acc += g[45 + 3 * b + c] * soft_bin_grad(x, b + 1, 16);
      atomicAdd(dimg + p * 3 + c, acc);
      const float gdy = g[3 + 2 * c], gdx = g[4 + 2 * c];
      const int hm = reflect(h - 1, H), hp = reflect(h + 1, H), wm = reflect(w - 1, W), wp = reflect(w + 1, W);
      const int ws[3] = {wm, w, wp}, hs[3] = {hm, h, hp};
      const float k[3] = {1.f, 2.f, 1.f};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        atomicAdd(dimg + (ibase + (long)hp * W + ws[j]) * 3 + c, k[j] * gdy);
        atomicAdd(dimg + (ibase + (long)hm * W + ws[j]) * 3 + c, -k[j] * gdy);
        atomicAdd(dimg + (ibase + (long)hs[j] * W + wp) * 3 + c, k[j] * gdx);
        atomicAdd(dimg + (ibase + (long)hs[j] * W + wm) * 3 + c, -k[j] * gdx);
      }
    }
  }
}

// A = B + alpha(B) * rev(hal), alpha = clamp((max_c B - 1 + thr)/thr, 0, 1)   (finetune_real_dataset.py:156-163)
//   d hal = rev(alpha * dA);   d B = dA + onehot(argmax_c B) * [0 < alpha < 1] * <dA, rev(hal)> / thr
__global__ __launch_bounds__(256) void alpha_blend_full_bwd_kernel(const float* __restrict__ bp, const float* __restrict__ hal,
                                                                   const float* __restrict__ dA, float* __restrict__ dB,
                                                                   float* __restrict__ dhal, long npix, float thr) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const float b[3] = {bp[3 * p], bp[3 * p + 1], bp[3 * p + 2]};
    const float g[3] = {dA[3 * p], dA[3 * p + 1], dA[3 * p + 2]};
    const float hr[3] = {hal[3 * p + 2], hal[3 * p + 1], hal[3 * p]};
    int arg = 0;
    float mx = b[0];
    if (b[1] > mx) { mx = b[1]; arg = 1; }
    if (b[2] > mx) { mx = b[2]; arg = 2; }
    const float u = mx - 1.0f + thr;
    const float al = fminf(1.0f, fmaxf(0.0f, u) / thr);
    const bool live = (u > 0.f) && (u / thr < 1.0f);
    const float dal = live ? (g[0] * hr[0] + g[1] * hr[1] + g[2] * hr[2]) / thr : 0.f;
    dB[3 * p] = g[0] + (arg == 0 ? dal : 0.f);
    dB[3 * p + 1] = g[1] + (arg == 1 ? dal : 0.f);
    dB[3 * p + 2] = g[2] + (arg == 2 ? dal : 0.f);
    dhal[3 * p] = al * g[2];
    dhal[3 * p + 1] = al * g[1];
    dhal[3 * p + 2] = al * g[0];
  }
}

struct Unpack3Args { float* o[4]; };
// o[s][p][0..2] = y[p][3s .. 3s+2]   (backward of pack3 / slice of the Refinement-Net input)
__global__ __launch_bounds__(256) void unpack3_kernel(const float* __restrict__ y, Unpack3Args out, int nout, int C, long npix) {
  const long total = npix * 3 * nout;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long p = e / (3 * nout);
    const int r = (int)(e - p * 3 * nout);
    const int s = r / 3, c = r - 3 * s;
    float* o = s == 0 ? out.o[0] : (s == 1 ? out.o[1] : (s == 2 ? out.o[2] : out.o[3]));
    o[3 * p + c] = y[p * C + r];
  }
}

// per-sample sums: out[b] += sum_i a[b][i] * (b2 ? b2[b][i] : 1)
__global__ __launch_bounds__(256) void sample_dot_kernel(const float* __restrict__ a, const float* __restrict__ b2,
                                                         float* __restrict__ out, long n_per) {
  __shared__ float sred[4];
  const int b = blockIdx.y;
  const float* pa = a + (long)b * n_per;
  const float* pb = b2 ? b2 + (long)b * n_per : nullptr;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_per; i += (long)gridDim.x * 256) s += pb ? pa[i] * pb[i] : pa[i];
  const float t = block_sum(s, sred);
  if (threadIdx.x == 0) atomicAdd(out + b, t);
}

// out = r / (eps + mean_b(r)) * target          (finetune_real_dataset.py:170)
__global__ __launch_bounds__(256) void mean_norm_apply_kernel(const float* __restrict__ r, const float* __restrict__ sum,
                                                              float* __restrict__ out, long n_per, int B, float eps, float target) {
  const long total = n_per * B;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float m = sum[i / n_per] / (float)n_per;
    out[i] = r[i] / (eps + m) * target;
  }
}
// dr = target * ( g/(eps+m) - <g, r> / (n (eps+m)^2) )
__global__ __launch_bounds__(256) void mean_norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ sum,
                                                            const float* __restrict__ gdot, float* __restrict__ dr, long n_per,
                                                            int B, float eps, float target) {
  const long total = n_per * B;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / n_per;
    const float d = eps + sum[b] / (float)n_per;
    dr[i] = target * (g[i] / d - gdot[b] / ((float)n_per * d * d));
  }
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" int shdr_lin_frontend_bwd_f32(const float* img, const float* dF, float* dimg, int N, int H, int W, int y_channels,
                                         void* stream) {
  SHDR_REQUIRE(img && dF && dimg, SHDR_E_NULL, "lin_frontend_bwd: null pointer");
  SHDR_REQUIRE(N > 0 && H >= 2 && W >= 2 && y_channels >= 93, SHDR_E_SHAPE, "lin_frontend_bwd: bad shape");
  hipStream_t st = S(stream);
  const long npix = (long)N * H * W;
  if (hipMemsetAsync(dimg, 0, sizeof(float) * 3 * npix, st) != hipSuccess) return shdr::fail(SHDR_E_LAUNCH, "lin_frontend_bwd: memset");
  hipLaunchKernelGGL(lin_frontend_bwd_kernel, dim3(shdr::stream_grid(npix)), dim3(256), 0, st, img, dF, dimg, N, H, W, y_channels);
  return shdr::check_launch("lin_frontend_bwd");
}

extern "C" int shdr_alpha_blend_full_bwd_f32(const float* b, const float* hal, const float* dA, float* dB, float* dhal,
                                             int64_t npix, float thr, void* stream) {
  SHDR_REQUIRE(b && hal && dA && dB && dhal, SHDR_E_NULL, "alpha_blend_full_bwd: null pointer");
  SHDR_REQUIRE(npix > 0 && thr > 0.f, SHDR_E_SHAPE, "alpha_blend_full_bwd: bad arguments");
  hipLaunchKernelGGL(alpha_blend_full_bwd_kernel, dim3(shdr::stream_grid(npix)), dim3(256), 0, S(stream), b, hal, dA, dB, dhal,
                     (long)npix, thr);
  return shdr::check_launch("alpha_blend_full_bwd");
}

extern "C" int shdr_unpack3_f32(const float* y, float* o0, float* o1, float* o2, float* o3, int nout, int channels,
                                int64_t npix, void* stream) {
  SHDR_REQUIRE(y && o0, SHDR_E_NULL, "unpack3: null pointer");
  SHDR_REQUIRE(nout >= 1 && nout <= 4 && channels >= 3 * nout && npix > 0, SHDR_E_SHAPE, "unpack3: bad arguments");
  float* o[4] = {o0, o1, o2, o3};
  for (int i = 0; i < nout; ++i) SHDR_REQUIRE(o[i], SHDR_E_NULL, "unpack3: output %d is null", i);
  Unpack3Args ua{{o0, o1, o2, o3}};
  hipLaunchKernelGGL(unpack3_kernel, dim3(shdr::stream_grid(npix * 3 * nout)), dim3(256), 0, S(stream), y, ua, nout, channels,
                     (long)npix);
  return shdr::check_launch("unpack3");
}

extern "C" int shdr_sample_dot_f32(const float* a, const float* b, float* out, int B, int64_t n_per_sample, void* stream) {
  SHDR_REQUIRE(a && out, SHDR_E_NULL, "sample_dot: null pointer");
  SHDR_REQUIRE(B > 0 && B <= 65535 && n_per_sample > 0, SHDR_E_SHAPE, "sample_dot: bad shape");
  hipStream_t st = S(stream);
  if (hipMemsetAsync(out, 0, sizeof(float) * B, st) != hipSuccess) return shdr::fail(SHDR_E_LAUNCH, "sample_dot: memset");
  int gx = shdr::stream_grid(n_per_sample);
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(sample_dot_kernel, dim3(gx, B), dim3(256), 0, st, a, b, out, (long)n_per_sample);
  return shdr::check_launch("sample_dot");
}

extern "C" int shdr_mean_norm_fwd_f32(const float* r, const float* sum, float* out, int B, int64_t n_per_sample, float eps,
                                      float target, void* stream) {
  SHDR_REQUIRE(r && sum && out, SHDR_E_NULL, "mean_norm_fwd: null pointer");
  SHDR_REQUIRE(B > 0 && n_per_sample > 0, SHDR_E_SHAPE, "mean_norm_fwd: bad shape");
  hipLaunchKernelGGL(mean_norm_apply_kernel, dim3(shdr::stream_grid(n_per_sample * B)), dim3(256), 0, S(stream), r, sum, out,
                     (long)n_per_sample, B, eps, target);
  return shdr::check_launch("mean_norm_fwd");
}

extern "C" int shdr_mean_norm_bwd_f32(const float* g, const float* sum, const float* gdot, float* dr, int B,
                                      int64_t n_per_sample, float eps, float target, void* stream) {
  SHDR_REQUIRE(g && sum && gdot && dr, SHDR_E_NULL, "mean_norm_bwd: null pointer");
  SHDR_REQUIRE(B > 0 && n_per_sample > 0, SHDR_E_SHAPE, "mean_norm_bwd: bad shape");
  hipLaunchKernelGGL(mean_norm_bwd_kernel, dim3(shdr::stream_grid(n_per_sample * B)), dim3(256), 0, S(stream), g, sum, gdot, dr,
                     (long)n_per_sample, B, eps, target);
  return shdr::check_launch("mean_norm_bwd");
}
