// Device side of the inference tool's image plumbing (test_real_refinement.py:119-155 of the reference; SURVEY.md
// section 8f rank 2): everything between the decoded 8-bit JPEG and the Radiance RGBE bytes stays in HBM --
//   uint8 BGR/RGB -> float [0,1]  ->  bicubic resize to a multiple of 64  ->  symmetric pad by 32  ->  (networks)
//   ->  crop  ->  bicubic resize back  ->  float RGB -> 4-byte RGBE.
// All kernels are HBM-bound elementwise / small-stencil passes (one thread per output element or pixel).
#include "shdr_internal.h"

namespace {

// y[p][c] = x[p][reverse ? 2-c : c] / 255   (np.flip(img, -1).astype(float32) / 255.0, :125)
__global__ __launch_bounds__(256) void u8_to_unit_kernel(const uint8_t* __restrict__ x, float* __restrict__ y, long npix,
                                                         int reverse) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const float a = (float)x[3 * p] / 255.0f, b = (float)x[3 * p + 1] / 255.0f, c = (float)x[3 * p + 2] / 255.0f;
    y[3 * p] = reverse ? c : a;
    y[3 * p + 1] = b;
    y[3 * p + 2] = reverse ? a : c;
  }
}

// OpenCV INTER_CUBIC (resize.cpp interpolateCubic, A = -0.75): source coordinate (d + 0.5) * scale - 0.5, taps
// floor-1 .. floor+2 clamped to the image (replicated border), no antialiasing in either direction.
__device__ __forceinline__ void cubic_coeffs(float t, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (t + 1.0f) - 5.0f * A) * (t + 1.0f) + 8.0f * A) * (t + 1.0f) - 4.0f * A;
  c[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
  c[2] = ((A + 2.0f) * (1.0f - t) - (A + 3.0f)) * (1.0f - t) * (1.0f - t) + 1.0f;
  c[3] = 1.0f - c[0] - c[1] - c[2];
}

__global__ __launch_bounds__(256) void resize_cubic_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H,
                                                           int W, int C, int Ho, int Wo, float sy, float sx) {
  const long total = (long)N * Ho * Wo * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    long t = e / C;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const long n = t / Ho;
    const float fy = ((float)oh + 0.5f) * sy - 0.5f, fx = ((float)ow + 0.5f) * sx - 0.5f;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float cy[4], cx[4];
    cubic_coeffs(fy - (float)iy, cy);
    cubic_coeffs(fx - (float)ix, cx);
    const float* xn = x + n * (long)H * W * C + c;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int yy = min(max(iy - 1 + i, 0), H - 1);
      float row = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xx = min(max(ix - 1 + j, 0), W - 1);
        row += cx[j] * xn[((long)yy * W + xx) * C];
      }
      acc += cy[i] * row;
    }
    y[e] = acc;
  }
}

// np.pad(..., 'symmetric'): index -1 -> 0, -2 -> 1, n -> n-1, n+1 -> n-2 (the edge sample is repeated)
__device__ __forceinline__ int sym_index(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i); }

__global__ __launch_bounds__(256) void pad_symmetric_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H,
                                                            int W, int C, int pad) {
  const int Ho = H + 2 * pad, Wo = W + 2 * pad;
  const long total = (long)N * Ho * Wo * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    long t = e / C;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const long n = t / Ho;
    y[e] = x[((n * H + sym_index(oh - pad, H)) * (long)W + sym_index(ow - pad, W)) * C + c];
  }
}

// Radiance RGBE (Ward, "Real pixels"): v = max(r,g,b); v < 1e-32 -> 0,0,0,0; else m * 2^e = v with m in [0.5,1),
// byte = (uint8)(channel * m * 256 / v), exponent byte = e + 128 -- the conversion cv2.imwrite('.hdr') applies (:150).
__global__ __launch_bounds__(256) void rgbe_encode_kernel(const float* __restrict__ x, uint8_t* __restrict__ y, long npix,
                                                          int reverse) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    float r = x[3 * p + (reverse ? 2 : 0)], g = x[3 * p + 1], b = x[3 * p + (reverse ? 0 : 2)];
    r = fmaxf(r, 0.0f); g = fmaxf(g, 0.0f); b = fmaxf(b, 0.0f);         // RGBE holds non-negative radiance
    const float v = fmaxf(fmaxf(r, g), b);
    uchar4 o = make_uchar4(0, 0, 0, 0);
    if (v >= 1e-32f) {
      int e;
      const float s = frexpf(v, &e) * 256.0f / v;
      o = make_uchar4((uint8_t)(r * s), (uint8_t)(g * s), (uint8_t)(b * s), (uint8_t)(e + 128));
    }
    *reinterpret_cast<uchar4*>(y + 4 * p) = o;
  }
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" int shdr_u8_to_unit_f32(const uint8_t* x, float* y, int64_t npix, int reverse_channels, void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "u8_to_unit: null pointer");
  SHDR_REQUIRE(npix > 0, SHDR_E_SHAPE, "u8_to_unit: npix must be positive");
  hipLaunchKernelGGL(u8_to_unit_kernel, dim3(shdr::stream_grid(npix)), dim3(256), 0, S(stream), x, y, (long)npix, reverse_channels);
  return shdr::check_launch("u8_to_unit");
}

extern "C" int shdr_resize_cubic_f32(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "resize_cubic: null pointer");
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, SHDR_E_SHAPE, "resize_cubic: non-positive dimension");
  const long total = (long)N * Ho * Wo * C;
  hipLaunchKernelGGL(resize_cubic_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0, S(stream), x, y, N, H, W, C, Ho, Wo,
                     (float)((double)H / Ho), (float)((double)W / Wo));
  return shdr::check_launch("resize_cubic");
}

extern "C" int shdr_pad_symmetric_f32(const float* x, float* y, int N, int H, int W, int C, int pad, void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "pad_symmetric: null pointer");
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, SHDR_E_SHAPE, "pad_symmetric: non-positive dimension");
  SHDR_REQUIRE(pad >= 0 && pad <= H && pad <= W, SHDR_E_SHAPE, "pad_symmetric: pad %d exceeds the image %dx%d", pad, H, W);
  const long total = (long)N * (H + 2 * pad) * (W + 2 * pad) * C;
  hipLaunchKernelGGL(pad_symmetric_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0, S(stream), x, y, N, H, W, C, pad);
  return shdr::check_launch("pad_symmetric");
}

extern "C" int shdr_rgbe_encode_f32(const float* x, uint8_t* y, int64_t npix, int reverse_channels, void* stream) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "rgbe_encode: null pointer");
  SHDR_REQUIRE(npix > 0, SHDR_E_SHAPE, "rgbe_encode: npix must be positive");
  SHDR_REQUIRE((reinterpret_cast<uintptr_t>(y) & 3u) == 0, SHDR_E_ALIGN, "rgbe_encode: y must be 4-byte aligned");
  hipLaunchKernelGGL(rgbe_encode_kernel, dim3(shdr::stream_grid(npix)), dim3(256), 0, S(stream), x, y, (long)npix, reverse_channels);
  return shdr::check_launch("rgbe_encode");
}

// ---- HDR-Real record augmentation (finetune_real_dataset.py:51-61): per-sample horizontal flip, then a counter-clockwise
//      rotation by k * 90 degrees (tf.image.rot90), and a scalar divide (ref_LDR / 255.0, :49) in the same pass ----------
namespace {
__global__ __launch_bounds__(256) void flip_rot90_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         const int* __restrict__ flip, const int* __restrict__ rot, int N,
                                                         int S, int C, float divisor) {
  const long per = (long)S * S * C;
  const long total = (long)N * per;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int n = (int)(e / per);
    long r = e - (long)n * per;
    const int c = (int)(r % C);
    r /= C;
    const int j = (int)(r % S), i = (int)(r / S);          // output pixel (i, j)
    int si, sj;                                            // pixel of the flipped image that lands there
    switch (rot[n] & 3) {
      case 1: si = j; sj = S - 1 - i; break;               // np.rot90(m, 1)[i][j] = m[j][S-1-i]
      case 2: si = S - 1 - i; sj = S - 1 - j; break;
      case 3: si = S - 1 - j; sj = i; break;
      default: si = i; sj = j; break;
    }
    if (flip[n]) sj = S - 1 - sj;                          // flip_left_right was applied first
    y[e] = x[((long)n * S * S + (long)si * S + sj) * C + c] / divisor;
  }
}
}  // namespace

extern "C" int shdr_flip_rot90_f32(const float* x, float* y, const int32_t* flip, const int32_t* rot, int N, int side, int C,
                                   float divisor, void* stream) {
  SHDR_REQUIRE(x && y && flip && rot, SHDR_E_NULL, "flip_rot90: null pointer");
  SHDR_REQUIRE(N > 0 && side > 0 && C > 0 && divisor != 0.0f, SHDR_E_SHAPE, "flip_rot90: bad shape (square images only)");
  hipLaunchKernelGGL(flip_rot90_kernel, dim3(shdr::stream_grid((long)N * side * side * C)), dim3(256), 0, S(stream), x, y, flip,
                     rot, N, side, C, divisor);
  return shdr::check_launch("flip_rot90");
}
