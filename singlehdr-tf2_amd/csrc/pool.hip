// Pooling, bilinear 2x resize and global average pooling (NHWC fp32, gfx950).
// All HBM-bound: one thread per 16-byte channel quad, grid-stride, coalesced
// along the channel axis (a wavefront touches 1 KiB contiguous).
#include "shdr_internal.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 max4(float4 a, float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }

// decode e -> (n, oh, ow, q) for an output of [N, Ho, Wo, 4*Q]
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

// AveragePooling2D(2,2) VALID (dequantization_net.py:10)
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, Q = C >> 2;
  const long total = (long)N * Ho * Wo * Q;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    SHDR_DECODE_QUAD(e, Q, Wo, Ho, q, ow, oh, n)
    const float* p = x + (((n * H + 2 * oh) * W + 2 * ow) * (long)C + 4 * q);
    float4 s = add4(add4(ld4(p), ld4(p + C)), add4(ld4(p + (long)W * C), ld4(p + (long)W * C + C)));
    st4(y + e * 4, make_float4(s.x * 0.25f, s.y * 0.25f, s.z * 0.25f, s.w * 0.25f));
  }
}

// MaxPool2D(2,2,SAME), even dims -> no padding (hallucination_net.py:49, vgg16.py:54)
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, Q = C >> 2;
  const long total = (long)N * Ho * Wo * Q;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    SHDR_DECODE_QUAD(e, Q, Wo, Ho, q, ow, oh, n)
    const float* p = x + (((n * H + 2 * oh) * W + 2 * ow) * (long)C + 4 * q);
    st4(y + e * 4, max4(max4(ld4(p), ld4(p + C)), max4(ld4(p + (long)W * C), ld4(p + (long)W * C + C))));
  }
}

// MaxPool2D(3,3,stride 2,SAME): Ho = ceil(H/2); pad_total = max((Ho-1)*2+3-H, 0),
// pad_before = pad_total/2; padded cells never win (linearization_net.py:94).
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int N, int H, int W, int C, int Ho, int Wo,
                                                         int pt, int pl) {
  const int Q = C >> 2;
  const long total = (long)N * Ho * Wo * Q;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    SHDR_DECODE_QUAD(e, Q, Wo, Ho, q, ow, oh, n)
    const float ninf = -__builtin_huge_valf();
    float4 m = make_float4(ninf, ninf, ninf, ninf);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ih = 2 * oh - pt + i;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int iw = 2 * ow - pl + j;
        if ((unsigned)iw >= (unsigned)W) continue;
        m = max4(m, ld4(x + (((n * H + ih) * W + iw) * (long)C + 4 * q)));
      }
    }
    st4(y + e * 4, m);
  }
}

// tf.image.resize(2x, BILINEAR), half-pixel centres (dequantization_net.py:25):
// src = (dst+0.5)/2 - 0.5 -> even dst: taps (m-1, m) lerp .75; odd dst: (m, m+1) lerp .25,
// indices clamped; value = top + (bot - top)*ly with top = l + (r - l)*lx.
// One thread owns one INPUT pixel quad and writes its 2x2 output quads from the 3x3 input
// neighbourhood (9 loads per 4 stores instead of 16).
__global__ __launch_bounds__(256) void resize2x_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       int N, int H, int W, int C) {
  const int Q = C >> 2;
  const long total = (long)N * H * W * Q;
  const long orow = (long)2 * W * C;
  auto vl = [](float4 t, float4 u, float ly) {
    return make_float4(t.x + (u.x - t.x) * ly, t.y + (u.y - t.y) * ly, t.z + (u.z - t.z) * ly, t.w + (u.w - t.w) * ly);
  };
  // the 9 loads of one input quad
  auto load9 = [&](long e, float4 (&v)[3][3], float*& o) {
    SHDR_DECODE_QUAD(e, Q, W, H, q, w, h, n)
    const int hm = max(h - 1, 0), hp = min(h + 1, H - 1), wm = max(w - 1, 0), wp = min(w + 1, W - 1);
    const float* b = x + (n * H * (long)W) * C + 4 * q;
    const int rows[3] = {hm, h, hp};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float* rp = b + (long)rows[r] * W * C;
      v[r][0] = ld4(rp + (long)wm * C); v[r][1] = ld4(rp + (long)w * C); v[r][2] = ld4(rp + (long)wp * C);
    }
    o = y + ((n * 2 * H + 2 * h) * (long)(2 * W) + 2 * w) * C + 4 * q;
  };
  auto emit = [&](const float4 (&v)[3][3], float* o) {
    float4 lo[3], hi[3];   // horizontally interpolated: lo = output column 2w, hi = output column 2w+1
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float4 l = v[r][0], c = v[r][1], rr = v[r][2];
      lo[r] = make_float4(l.x + (c.x - l.x) * 0.75f, l.y + (c.y - l.y) * 0.75f, l.z + (c.z - l.z) * 0.75f, l.w + (c.w - l.w) * 0.75f);
      hi[r] = make_float4(c.x + (rr.x - c.x) * 0.25f, c.y + (rr.y - c.y) * 0.25f, c.z + (rr.z - c.z) * 0.25f, c.w + (rr.w - c.w) * 0.25f);
    }
    st4(o, vl(lo[0], lo[1], 0.75f));
    st4(o + C, vl(hi[0], hi[1], 0.75f));
    st4(o + orow, vl(lo[1], lo[2], 0.25f));
    st4(o + orow + C, vl(hi[1], hi[2], 0.25f));
  };
  const long step = (long)gridDim.x * 256;
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  for (; e + step < total; e += 2 * step) {       // two input quads (18 loads) in flight
    float4 va[3][3], vb[3][3];
    float *oa, *ob;
    load9(e, va, oa);
    load9(e + step, vb, ob);
    emit(va, oa);
    emit(vb, ob);
  }
  if (e < total) {
    float4 va[3][3];
    float* oa;
    load9(e, va, oa);
    emit(va, oa);
  }
}

// tf.reduce_mean(x,[1,2]) (linearization_net.py:118): x [N,HW,C] -> y [N,C].
// grid (C/64, N); thread = (channel quad 0..15, pixel lane 0..15).
__global__ __launch_bounds__(256) void gap_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                  int HW, int C) {
  __shared__ float4 part[16][16];
  const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 64 + 4 * q;
  const long n = blockIdx.y;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c0 < C) {
    // four independent loads in flight per thread: the sequential form ran at the latency of 16 dependent loads (0.12 ms for
    // the 33 MB of the Linearization-Net's 16 x 256 x 2048 tensor)
    const float* xb = x + n * HW * (long)C + c0;
    float4 s1 = s, s2 = s, s3 = s;
    int p = g;
    for (; p + 48 < HW; p += 64) {
      s = add4(s, ld4(xb + (long)p * C));
      s1 = add4(s1, ld4(xb + (long)(p + 16) * C));
      s2 = add4(s2, ld4(xb + (long)(p + 32) * C));
      s3 = add4(s3, ld4(xb + (long)(p + 48) * C));
    }
    for (; p < HW; p += 16) s = add4(s, ld4(xb + (long)p * C));
    s = add4(add4(s, s1), add4(s2, s3));
  }
  part[g][q] = s;
  __syncthreads();
  if (g == 0 && c0 < C) {
    float4 t = part[0][q];
#pragma unroll
    for (int i = 1; i < 16; ++i) t = add4(t, part[i][q]);
    const float inv = 1.0f / (float)HW;
    st4(y + n * C + c0, make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv));
  }
}

int check_nhwc4(const char* op, const void* x, const void* y, int N, int H, int W, int C) {
  SHDR_REQUIRE(x && y, SHDR_E_NULL, "%s: null pointer", op);
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, SHDR_E_SHAPE, "%s: non-positive dimension", op);
  SHDR_REQUIRE((C & 3) == 0, SHDR_E_ALIGN, "%s: C=%d must be a multiple of 4", op, C);
  SHDR_REQUIRE(shdr::aligned16(x) && shdr::aligned16(y), SHDR_E_ALIGN, "%s: tensors must be 16-byte aligned", op);
  return SHDR_OK;
}

}  // namespace

extern "C" int shdr_avgpool2_fwd_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  if (int rc = check_nhwc4("avgpool2", x, y, N, H, W, C)) return rc;
  SHDR_REQUIRE(H >= 2 && W >= 2, SHDR_E_SHAPE, "avgpool2: H, W must be >= 2");
  const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(avgpool2_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, N, H, W, C);
  return shdr::check_launch("avgpool2");
}

extern "C" int shdr_maxpool2_fwd_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  if (int rc = check_nhwc4("maxpool2", x, y, N, H, W, C)) return rc;
  SHDR_REQUIRE((H & 1) == 0 && (W & 1) == 0, SHDR_E_SHAPE, "maxpool2: H=%d, W=%d must be even", H, W);
  const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, N, H, W, C);
  return shdr::check_launch("maxpool2");
}

extern "C" int shdr_maxpool3s2_fwd_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  if (int rc = check_nhwc4("maxpool3s2", x, y, N, H, W, C)) return rc;
  int Ho, Wo, pt, pl;
  shdr_same_pad(H, 3, 2, &Ho, &pt);
  shdr_same_pad(W, 3, 2, &Wo, &pl);
  const long total = (long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3s2_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, N, H, W, C, Ho, Wo, pt, pl);
  return shdr::check_launch("maxpool3s2");
}

extern "C" int shdr_resize2x_fwd_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  if (int rc = check_nhwc4("resize2x", x, y, N, H, W, C)) return rc;
  const long total = (long)N * H * W * (C / 4);
  hipLaunchKernelGGL(resize2x_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, N, H, W, C);
  return shdr::check_launch("resize2x");
}

extern "C" int shdr_gap_fwd_f32(const float* x, float* y, int N, int HW, int C, void* stream) {
  if (int rc = check_nhwc4("gap", x, y, N, HW, 1, C)) return rc;
  hipLaunchKernelGGL(gap_kernel, dim3((C + 63) / 64, N), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, HW, C);
  return shdr::check_launch("gap");
}
