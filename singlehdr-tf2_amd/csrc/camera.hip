// Camera-pipeline simulator of joint_training.py:26-69 on gfx950: exposure, shot / read noise,
// dynamic-range clipping, 8-bit quantisation, a baseline-JPEG round trip and the over/under-exposure loss mask -- the part
// of the reference's train step that runs as tf ops plus a per-image libjpeg encode/decode on the CPU
// (tf.image.adjust_jpeg_quality).  Everything stays in HBM; the CRF itself is the existing shdr_apply_rf_fwd_f32.
//
//  * camera_expose_kernel : hdr * t, Gaussian noise from a counter-based Philox4x32-10 stream (stateless, reproducible
//                           for a given seed on any grid), relu, clip.
//  * jpeg_mcu_kernel      : one wavefront per 16x16 MCU.  libjpeg's arithmetic is integer and is reproduced BIT FOR BIT:
//                           RGB->YCbCr in 16-bit fixed point (jccolor.c), h2v2 chroma box filter with the alternating
//                           1/2 bias (jcsample.c), the "islow" Loeffler-Ligtenberg-Moschytz DCT (jfdctint.c), the IJG
//                           quality scaling of the Annex-K tables and the round-half-up quantiser (jcparam.c,
//                           jcdctmgr.c), dequantisation and the islow IDCT (jidctint.c).  Entropy coding is lossless and
//                           is skipped.  Output: Y' at full and Cb'/Cr' at half resolution, 8 bits each.
//  * jpeg_finish_kernel   : fancy (triangle) chroma upsampling across MCU borders (jdsample.c h2v2_fancy_upsample),
//                           YCbCr->RGB (jdcolor.c), /255, and the grey-level census of the loss mask.
// Compiled with -ffp-contract=off: the float steps (noise chain, grey conversion) round exactly like the fp32 restatement
// in oracle/camera.py.
#include "shdr_internal.h"

namespace {

// ---------------------------------------------------------------- Philox4x32-10 (Salmon et al., SC'11)
struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
    c = U4{(uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

__device__ __forceinline__ float u01_open(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f; }  // (0,1)
__device__ __forceinline__ float u01_half_open(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }                        // [0,1)

__global__ __launch_bounds__(256) void camera_expose_kernel(const float* __restrict__ hdr, const float* __restrict__ t,
                                                            float* __restrict__ hdr_t, float* __restrict__ clipped, int N,
                                                            long per_image, uint32_t k0, uint32_t k1) {
  const long total = (long)N * per_image;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int n = (int)(e / per_image);
    const int c = (int)(e % 3);
    // per-(sample, channel) noise levels: sigma_s = 0.08/6 * U[0,1), sigma_c = 0.005 * U[0,1)   (:33-34)
    const U4 s = philox4x32_10(U4{(uint32_t)(n * 3 + c), 0u, 0u, 1u}, k0, k1);
    const float sigma_s = (float)(0.08 / 6.0) * u01_half_open(s.x);
    const float sigma_c = 0.005f * u01_half_open(s.y);
    const U4 r = philox4x32_10(U4{(uint32_t)e, (uint32_t)((uint64_t)e >> 32), 0u, 0u}, k0, k1);
    const float rad = sqrtf(-2.0f * logf(u01_open(r.x)));
    const float ang = 6.283185307179586f * u01_open(r.y);
    const float z0 = rad * cosf(ang), z1 = rad * sinf(ang);
    const float x = hdr[e] * t[n];                  // _hdr_t = hdr * t                      (:30)
    float v = x + z0 * (sigma_s * x);               // + normal * (sigma_s * _hdr_t)         (:35-37)
    v = v + sigma_c * z1;                           // + sigma_c * normal                    (:38-39)
    v = fmaxf(v, 0.0f);                             // relu                                  (:40)
    hdr_t[e] = v;
    clipped[e] = fminf(v, 1.0f);                    // clip_by_value(_hdr_t, 0, 1)           (:43)
  }
}

// ---------------------------------------------------------------- libjpeg integer arithmetic
constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270;
constexpr int F_0_899976223 = 7373, F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137;
constexpr int F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// 1-D forward pass over d[0..7*stride]
__device__ __forceinline__ void fdct_1d(int* d, int stride, bool first) {
  const int a0 = d[0], a1 = d[stride], a2 = d[2 * stride], a3 = d[3 * stride], a4 = d[4 * stride], a5 = d[5 * stride],
            a6 = d[6 * stride], a7 = d[7 * stride];
  int t0 = a0 + a7, t7 = a0 - a7, t1 = a1 + a6, t6 = a1 - a6, t2 = a2 + a5, t5 = a2 - a5, t3 = a3 + a4, t4 = a3 - a4;
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  const int sh = first ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS;
  d[0] = first ? (t10 + t11) << PASS1_BITS : descale(t10 + t11, PASS1_BITS);
  d[4 * stride] = first ? (t10 - t11) << PASS1_BITS : descale(t10 - t11, PASS1_BITS);
  int z1 = (t12 + t13) * F_0_541196100;
  d[2 * stride] = descale(z1 + t13 * F_0_765366865, sh);
  d[6 * stride] = descale(z1 - t12 * F_1_847759065, sh);
  z1 = t4 + t7;
  int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int z5 = (z3 + z4) * F_1_175875602;
  t4 *= F_0_298631336; t5 *= F_2_053119869; t6 *= F_3_072711026; t7 *= F_1_501321110;
  z1 *= -F_0_899976223; z2 *= -F_2_562915447;
  z3 = z3 * -F_1_961570560 + z5;
  z4 = z4 * -F_0_390180644 + z5;
  d[7 * stride] = descale(t4 + z1 + z3, sh);
  d[5 * stride] = descale(t5 + z2 + z4, sh);
  d[3 * stride] = descale(t6 + z2 + z3, sh);
  d[stride] = descale(t7 + z1 + z4, sh);
}

__device__ __forceinline__ void idct_1d(int* d, int stride, bool first) {
  int z2 = d[2 * stride], z3 = d[6 * stride];
  int z1 = (z2 + z3) * F_0_541196100;
  int t2 = z1 - z3 * F_1_847759065, t3 = z1 + z2 * F_0_765366865;
  int t0 = (d[0] + d[4 * stride]) << CONST_BITS, t1 = (d[0] - d[4 * stride]) << CONST_BITS;
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  t0 = d[7 * stride]; t1 = d[5 * stride]; t2 = d[3 * stride]; t3 = d[stride];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
  int z4 = t1 + t3;
  const int z5 = (z3 + z4) * F_1_175875602;
  t0 *= F_0_298631336; t1 *= F_2_053119869; t2 *= F_3_072711026; t3 *= F_1_501321110;
  z1 *= -F_0_899976223; z2 *= -F_2_562915447;
  z3 = z3 * -F_1_961570560 + z5;
  z4 = z4 * -F_0_390180644 + z5;
  t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
  const int sh = first ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS + 3;
  d[0] = descale(t10 + t3, sh); d[7 * stride] = descale(t10 - t3, sh);
  d[stride] = descale(t11 + t2, sh); d[6 * stride] = descale(t11 - t2, sh);
  d[2 * stride] = descale(t12 + t1, sh); d[5 * stride] = descale(t12 - t1, sh);
  d[3 * stride] = descale(t13 + t0, sh); d[4 * stride] = descale(t13 - t0, sh);
}

// ITU T.81 Annex K tables (natural order)
__constant__ uint8_t kLumaQ[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                   14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
__constant__ uint8_t kChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                                     47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

__device__ __forceinline__ int fix16(double x) { return (int)(x * 65536.0 + 0.5); }

// grid = (W/16, H/16, N); block = 64 threads (one wavefront) = one MCU
__global__ __launch_bounds__(64) void jpeg_mcu_kernel(const float* __restrict__ ldr, const int* __restrict__ quality,
                                                      uint8_t* __restrict__ yp, uint8_t* __restrict__ cbp,
                                                      uint8_t* __restrict__ crp, int H, int W) {
  __shared__ int blk[6][64];      // Y00 Y01 Y10 Y11 Cb Cr, [row][col] of each 8x8 block
  __shared__ int cfull[2][256];   // full-resolution Cb, Cr of the MCU
  const int lane = threadIdx.x;
  const int n = blockIdx.z, my = blockIdx.y, mx = blockIdx.x;
  const float* img = ldr + (size_t)n * H * W * 3;

  // 8-bit quantisation (tf.round = half to even, :46-47) and colour conversion, 4 pixels per lane
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = lane + 64 * i, py = p >> 4, px = p & 15;
    const float* s = img + ((size_t)(my * 16 + py) * W + mx * 16 + px) * 3;
    const int r = (int)fminf(fmaxf(rintf(s[0] * 255.0f), 0.0f), 255.0f);
    const int g = (int)fminf(fmaxf(rintf(s[1] * 255.0f), 0.0f), 255.0f);
    const int b = (int)fminf(fmaxf(rintf(s[2] * 255.0f), 0.0f), 255.0f);
    const int half = 1 << 15, off = 128 << 16;
    const int y = (19595 * r + 38470 * g + 7471 * b + half) >> 16;                   // FIX(0.299), FIX(0.587), FIX(0.114)
    const int cb = (-11059 * r - 21709 * g + 32768 * b + off + half - 1) >> 16;      // FIX(0.16874), FIX(0.33126), FIX(0.5)
    const int cr = (32768 * r - 27439 * g - 5329 * b + off + half - 1) >> 16;        // FIX(0.5), FIX(0.41869), FIX(0.08131)
    blk[(py >> 3) * 2 + (px >> 3)][(py & 7) * 8 + (px & 7)] = y - 128;
    cfull[0][p] = cb;
    cfull[1][p] = cr;
  }
  __syncthreads();
  {  // h2v2 box filter: lane = chroma sample (cy, cx); bias 1, 2, 1, 2 ... along the row of the WHOLE image
    const int cy = lane >> 3, cx = lane & 7;
    const int bias = 1 + ((mx * 8 + cx) & 1);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int* c = cfull[k] + (2 * cy) * 16 + 2 * cx;
      blk[4 + k][lane] = ((c[0] + c[1] + c[16] + c[17] + bias) >> 2) - 128;
    }
  }
  __syncthreads();
  // forward DCT: 48 row passes, then 48 column passes
  if (lane < 48) fdct_1d(&blk[lane >> 3][(lane & 7) * 8], 1, true);
  __syncthreads();
  if (lane < 48) fdct_1d(&blk[lane >> 3][lane & 7], 8, false);
  __syncthreads();
  // quantise + dequantise: (|c| + 4q) / 8q with the sign restored, times q   (the DCT output is scaled by 8)
  {
    int q = quality[n];
    q = q < 1 ? 1 : (q > 100 ? 100 : q);
    const int scale = q < 50 ? 5000 / q : 200 - 2 * q;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      int tq = ((b < 4 ? kLumaQ[lane] : kChromaQ[lane]) * scale + 50) / 100;
      tq = tq < 1 ? 1 : (tq > 255 ? 255 : tq);
      const int c = blk[b][lane];
      const int a = ((c < 0 ? -c : c) + 4 * tq) / (8 * tq);
      blk[b][lane] = (c < 0 ? -a : a) * tq;
    }
  }
  __syncthreads();
  // inverse DCT: columns first (jidctint.c pass 1), then rows
  if (lane < 48) idct_1d(&blk[lane >> 3][lane & 7], 8, true);
  __syncthreads();
  if (lane < 48) idct_1d(&blk[lane >> 3][(lane & 7) * 8], 1, false);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = lane + 64 * i, py = p >> 4, px = p & 15;
    const int v = blk[(py >> 3) * 2 + (px >> 3)][(py & 7) * 8 + (px & 7)] + 128;
    yp[((size_t)n * H + my * 16 + py) * W + mx * 16 + px] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
  {
    const int cy = lane >> 3, cx = lane & 7;
    const size_t o = ((size_t)n * (H / 2) + my * 8 + cy) * (W / 2) + mx * 8 + cx;
    const int vb = blk[4][lane] + 128, vr = blk[5][lane] + 128;
    cbp[o] = (uint8_t)(vb < 0 ? 0 : (vb > 255 ? 255 : vb));
    crp[o] = (uint8_t)(vr < 0 ? 0 : (vr > 255 ? 255 : vr));
  }
}

__device__ __forceinline__ int fancy_up(const uint8_t* __restrict__ p, int ch, int cw, int y, int x) {
  // output sample (y, x) of the 2x upsampled plane: near row / far row, near column / far column, weights 9:3:3:1
  const int cy = y >> 1, cx = x >> 1;
  const int fy = min(max(cy + ((y & 1) ? 1 : -1), 0), ch - 1);
  const int fx = min(max(cx + ((x & 1) ? 1 : -1), 0), cw - 1);
  const int this_col = 3 * p[cy * cw + cx] + p[fy * cw + cx];
  const int far_col = 3 * p[cy * cw + fx] + p[fy * cw + fx];
  return (3 * this_col + far_col + ((x & 1) ? 7 : 8)) >> 4;
}

// one thread per pixel; counts[n][0] = #grey >= 249, counts[n][1] = #grey <= 6
__global__ __launch_bounds__(256) void jpeg_finish_kernel(const uint8_t* __restrict__ yp, const uint8_t* __restrict__ cbp,
                                                          const uint8_t* __restrict__ crp, float* __restrict__ out,
                                                          int* __restrict__ counts, int H, int W) {
  const int n = blockIdx.y;
  const int npix = H * W;
  const int ch = H / 2, cw = W / 2;
  const uint8_t* cb = cbp + (size_t)n * ch * cw;
  const uint8_t* cr = crp + (size_t)n * ch * cw;
  int over = 0, under = 0;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
    const int y = p / W, x = p - y * W;
    const int yy = yp[(size_t)n * npix + p];
    const int u = fancy_up(cb, ch, cw, y, x) - 128, v = fancy_up(cr, ch, cw, y, x) - 128;
    const int half = 1 << 15;
    int r = yy + ((91881 * v + half) >> 16);                       // FIX(1.40200)
    int g = yy + ((-22554 * u + half - 46802 * v) >> 16);          // FIX(0.34414), FIX(0.71414)
    int b = yy + ((116130 * u + half) >> 16);                      // FIX(1.77200)
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    b = b < 0 ? 0 : (b > 255 ? 255 : b);
    float* o = out + ((size_t)n * npix + p) * 3;
    o[0] = (float)r / 255.0f; o[1] = (float)g / 255.0f; o[2] = (float)b / 255.0f;     // tf.cast(jpeg, float32) / 255   (:52)
    // tf.image.rgb_to_grayscale on uint8: u8 * (1/255) -> dot (0.2989, 0.5870, 0.1140) -> saturate_cast(x * 255.5)
    const float k = 1.0f / 255.0f;
    const float gr = ((float)r * k) * 0.2989f + ((float)g * k) * 0.5870f + ((float)b * k) * 0.1140f;
    const int gray = (int)fminf(fmaxf(gr * 255.5f, 0.0f), 255.0f);
    over += gray >= 249;
    under += gray <= 6;
  }
  for (int off = 32; off > 0; off >>= 1) {
    over += __shfl_down(over, off);
    under += __shfl_down(under, off);
  }
  if ((threadIdx.x & 63) == 0) {
    if (over) atomicAdd(counts + 2 * n, over);
    if (under) atomicAdd(counts + 2 * n + 1, under);
  }
}

// loss_mask = !(over > 256*256*0.5 || under > 256*256*0.5)   (:56-63; the threshold is the reference's constant)
__global__ void loss_mask_kernel(const int* __restrict__ counts, float* __restrict__ mask, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) mask[n] = ((float)counts[2 * n] > 32768.0f || (float)counts[2 * n + 1] > 32768.0f) ? 0.0f : 1.0f;
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" int shdr_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
  if (!counter || !key || !out) return shdr::fail(SHDR_E_NULL, "philox: null pointer");
  const U4 r = philox4x32_10(U4{counter[0], counter[1], counter[2], counter[3]}, key[0], key[1]);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
  return SHDR_OK;
}

extern "C" int shdr_camera_expose_f32(const float* hdr, const float* t, float* hdr_t, float* clipped, int N, int H, int W,
                                      uint64_t seed, void* stream) {
  SHDR_REQUIRE(hdr && t && hdr_t && clipped, SHDR_E_NULL, "camera_expose: null pointer");
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0, SHDR_E_SHAPE, "camera_expose: non-positive dimension");
  const long per = (long)H * W * 3;
  hipLaunchKernelGGL(camera_expose_kernel, dim3(shdr::stream_grid((long)N * per)), dim3(256), 0, S(stream), hdr, t, hdr_t,
                     clipped, N, per, (uint32_t)seed, (uint32_t)(seed >> 32));
  return shdr::check_launch("camera_expose");
}

extern "C" int shdr_jpeg_round_trip_f32(const float* ldr, const int32_t* quality, float* jpeg, float* loss_mask,
                                        uint8_t* ws_planes, int32_t* ws_counts, int N, int H, int W, void* stream) {
  SHDR_REQUIRE(ldr && quality && jpeg && ws_planes && ws_counts, SHDR_E_NULL, "jpeg_round_trip: null pointer");
  SHDR_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0 && H / 16 <= 65535, SHDR_E_SHAPE,
               "jpeg_round_trip: need whole 16x16 MCUs (H %% 16 == W %% 16 == 0), got %dx%d", H, W);
  hipStream_t st = S(stream);
  const size_t npix = (size_t)N * H * W;
  uint8_t* yp = ws_planes;
  uint8_t* cbp = yp + npix;
  uint8_t* crp = cbp + npix / 4;
  hipError_t e = hipMemsetAsync(ws_counts, 0, sizeof(int32_t) * 2 * N, st);
  if (e != hipSuccess) return shdr::fail(SHDR_E_LAUNCH, "jpeg_round_trip: memset: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(jpeg_mcu_kernel, dim3(W / 16, H / 16, N), dim3(64), 0, st, ldr, quality, yp, cbp, crp, H, W);
  int gx = (H * W + 255) / 256;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(jpeg_finish_kernel, dim3(gx, N), dim3(256), 0, st, yp, cbp, crp, jpeg, ws_counts, H, W);
  if (loss_mask) hipLaunchKernelGGL(loss_mask_kernel, dim3((N + 63) / 64), dim3(64), 0, st, ws_counts, loss_mask, N);
  return shdr::check_launch("jpeg_round_trip");
}
