// Inverse-CRF head of the Linearization-Net and CRF application (gfx950).
//   shdr_invcrf_decode_fwd_f32  Dense(11) + EMoR PCA decode  (linearization_net.py:185-192,231-253)
//   shdr_increase_fwd_f32       monotone fix-up `_increase`   (linearization_net.py:368-392)
//   shdr_apply_rf_fwd_f32       1024-entry LUT lerp            (tf_utils.py:54-105)
#include "shdr_internal.h"

namespace {

constexpr int NPCA = 11;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// one block (256 threads) per batch row
__global__ __launch_bounds__(256) void invcrf_decode_kernel(const float* __restrict__ feat,
                                                            const float* __restrict__ wfc,
                                                            const float* __restrict__ bfc,
                                                            const float* __restrict__ table,
                                                            float* __restrict__ out, int F, int K) {
  __shared__ float red[4][NPCA];
  __shared__ float wv[NPCA];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float part[NPCA];
#pragma unroll
  for (int j = 0; j < NPCA; ++j) part[j] = 0.f;
  for (int f = tid; f < F; f += 256) {
    const float xv = feat[(long)b * F + f];
#pragma unroll
    for (int j = 0; j < NPCA; ++j) part[j] = fmaf(xv, wfc[f * NPCA + j], part[j]);
  }
#pragma unroll
  for (int j = 0; j < NPCA; ++j) {
    const float s = wave_sum(part[j]);
    if (lane == 0) red[wave][j] = s;
  }
  __syncthreads();
  if (tid < NPCA) wv[tid] = bfc[tid] + ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
  __syncthreads();
  for (int k = tid; k < K; k += 256) {
    const float* row = table + (long)k * (NPCA + 1);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NPCA; ++j) s = fmaf(row[1 + j], wv[j], s);
    out[(long)b * K + k] = row[0] + s;
  }
}

// one block (256 threads) per batch row, K <= 4096.  min / sum are block reductions; the
// prefix sum itself is run sequentially by one thread in the reference's (TF-CPU cumsum)
// order: fp32 addition is monotone, so with g >= 0 the CDF is non-decreasing bit for bit,
// which a tree scan does not guarantee.  1023 dependent adds ~ 2 us; one block per image.
__global__ __launch_bounds__(256) void increase_kernel(const float* __restrict__ rf,
                                                       float* __restrict__ out, int K) {
  extern __shared__ float g[];  // K - 1 gradients, reused for the running sum
  __shared__ float sred[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* r = rf + (long)b * K;
  const int G = K - 1;
  float mn = __builtin_huge_valf();
  for (int k = tid; k < G; k += 256) {
    const float d = r[k + 1] - r[k];
    g[k] = d;
    mn = fminf(mn, d);
  }
  mn = wave_min(mn);
  if (lane == 0) sred[wave] = mn;
  __syncthreads();
  mn = fminf(fminf(sred[0], sred[1]), fminf(sred[2], sred[3]));
  const float rr = fmaxf(-mn, 0.f);
  __syncthreads();
  float loc = 0.f;
  for (int k = tid; k < G; k += 256) {
    const float v = g[k] + rr;
    g[k] = v;
    loc += v;
  }
  loc = wave_sum(loc);
  if (lane == 0) sred[wave] = loc;
  __syncthreads();
  const float total = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  for (int k = tid; k < G; k += 256) g[k] = g[k] / total;
  __syncthreads();
  if (tid == 0) {
    float run = 0.f;
    for (int k = 0; k < G; ++k) {
      run += g[k];
      g[k] = run;
    }
  }
  __syncthreads();
  float* o = out + (long)b * K;
  if (tid == 0) o[0] = 0.f;
  for (int k = tid; k < G; k += 256) o[k + 1] = g[k];
}

// grid (blocks_x, B): the batch row's LUT lives in LDS (K*4 bytes).
__global__ __launch_bounds__(256) void apply_rf_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ rf,
                                                       float* __restrict__ y, long n_per_batch, int K) {
  extern __shared__ float lut[];
  const int b = blockIdx.y;
  for (int k = threadIdx.x; k < K; k += 256) lut[k] = rf[(long)b * K + k];
  __syncthreads();
  const float km1 = (float)(K - 1);
  auto one = [&](float xv) {
#pragma clang fp contract(off)
    const float yv = km1 * xv;
    const float y0 = floorf(yv);
    const float y1 = y0 + 1.0f;
    const int i0 = min(max((int)y0, 0), K - 1);
    const int i1 = min(max((int)y1, 0), K - 1);
    return (y1 - yv) * lut[i0] + (yv - y0) * lut[i1];
  };
  const float* xb = x + (long)b * n_per_batch;
  float* yb = y + (long)b * n_per_batch;
  const long nq = n_per_batch >> 2;
  const bool vec = ((n_per_batch & 3) == 0);
  if (vec) {
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
      const float4 v = *reinterpret_cast<const float4*>(xb + 4 * q);
      *reinterpret_cast<float4*>(yb + 4 * q) = make_float4(one(v.x), one(v.y), one(v.z), one(v.w));
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_per_batch; i += (long)gridDim.x * 256)
      yb[i] = one(xb[i]);
  }
}

}  // namespace

extern "C" int shdr_invcrf_decode_fwd_f32(const float* feat, const float* wfc, const float* bfc,
                                          const float* table, float* out, int B, int F, int K,
                                          void* stream) {
  SHDR_REQUIRE(feat && wfc && bfc && table && out, SHDR_E_NULL, "invcrf_decode: null pointer");
  SHDR_REQUIRE(B > 0 && F > 0 && K > 0, SHDR_E_SHAPE, "invcrf_decode: non-positive dimension");
  hipLaunchKernelGGL(invcrf_decode_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     feat, wfc, bfc, table, out, F, K);
  return shdr::check_launch("invcrf_decode");
}

extern "C" int shdr_increase_fwd_f32(const float* rf, float* out, int B, int K, void* stream) {
  SHDR_REQUIRE(rf && out, SHDR_E_NULL, "increase: null pointer");
  SHDR_REQUIRE(B > 0 && K >= 2 && K <= 4096, SHDR_E_SHAPE, "increase: need B>0 and 2 <= K <= 4096 (K=%d)", K);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(increase_kernel, dim3(B), dim3(256), (size_t)(K - 1) * sizeof(float), st, rf, out, K);
  return shdr::check_launch("increase");
}

extern "C" int shdr_apply_rf_fwd_f32(const float* x, const float* rf, float* y, int B,
                                     int64_t n_per_batch, int K, void* stream) {
  SHDR_REQUIRE(x && rf && y, SHDR_E_NULL, "apply_rf: null pointer");
  SHDR_REQUIRE(B > 0 && B <= 65535 && n_per_batch > 0 && K >= 2 && K <= 16384, SHDR_E_SHAPE,
               "apply_rf: bad shape (B=%d, K=%d)", B, K);
  SHDR_REQUIRE((n_per_batch & 3) != 0 || (shdr::aligned16(x) && shdr::aligned16(y)), SHDR_E_ALIGN,
               "apply_rf: x/y must be 16-byte aligned");
  int gx = shdr::stream_grid((n_per_batch + 3) / 4);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(apply_rf_kernel, dim3(gx, B), dim3(256), (size_t)K * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), x, rf, y, (long)n_per_batch, K);
  return shdr::check_launch("apply_rf");
}
