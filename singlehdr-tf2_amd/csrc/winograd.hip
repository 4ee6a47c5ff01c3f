// Winograd F(2x2, 3x3) for the wide 3x3 / stride-1 / SAME convolutions (Hallucination-Net encoder and
// decoder, VGG16): 2.25x fewer multiplies than the direct form.
//
//   U[xi] = (G g G^T)[xi]                 filter transform, [16][Cin][Cout]          (cached by the caller)
//   V[xi] = (B^T d B)[xi]                 input transform,  [16][T][Cin], T = tiles of 2x2 outputs
//   M[xi] = V[xi] @ U[xi]                 16 GEMMs = ONE launch of conv_mfma_dma_kernel on a
//                                         "16-image batch" with a per-image filter (w_batch_stride)
//   Y     = A^T M A (+ fused epilogue)    output transform
//
// The transforms are HBM-bound elementwise kernels (one thread = one tile x one channel quad, every
// plane access is a coalesced float4); the GEMM is the fp32-MFMA kernel.  Numerically F(2,3) only adds
// and halves: the result differs from the direct conv at the 1e-6 relative level.
#include "shdr_internal.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// u[xi = i*4+j][ci][co] = sum_{a,b} G[i][a] g[a][b][ci][co] G[j][b],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(256) void winograd_filter_kernel(const float* __restrict__ w, float* __restrict__ u, long cc) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < cc; e += (long)gridDim.x * 256) {
    float g[3][3], t[4][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = w[(a * 3 + b) * cc + e];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = g[0][b];
      t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      t[3][b] = g[2][b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[(i * 4 + 0) * cc + e] = t[i][0];
      u[(i * 4 + 1) * cc + e] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
      u[(i * 4 + 2) * cc + e] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
      u[(i * 4 + 3) * cc + e] = t[i][2];
    }
  }
}

// The same U = G g G^T, stored in the operand order of winograd_fused_kernel (winograd_fused.hip): for cout slice pn, wave w8
// (transform positions 2*w8, 2*w8+1), 8-channel chunk c, the 16 B-operand values of a lane are four float4
//   up[((((pn*8 + w8)*nch + c)*4 + j)*64 + lane)*4 + nt],   j = 2*(xi & 1) + (ci & 1),  lane = 16*((ci & 7) >> 1) + (co & 15),
//   nt = (co & 63) >> 4  -- one fully coalesced 1 KiB global_load_dwordx4 per (chunk, j) and wave, straight into VGPRs.
__global__ __launch_bounds__(256) void winograd_filter_packed_kernel(const float* __restrict__ w, float* __restrict__ up, int Cin,
                                                                     int Cout) {
  const long cc = (long)Cin * Cout;
  const int nch = Cin >> 3;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < cc; e += (long)gridDim.x * 256) {
    const int co = (int)(e % Cout), ci = (int)(e / Cout);
    float g[3][3], t[4][3], u[16];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = w[(a * 3 + b) * cc + e];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = g[0][b];
      t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      t[3][b] = g[2][b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[i * 4 + 0] = t[i][0];
      u[i * 4 + 1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
      u[i * 4 + 2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
      u[i * 4 + 3] = t[i][2];
    }
    const int pn = co >> 6, nt = (co & 63) >> 4, lane = 16 * ((ci & 7) >> 1) + (co & 15);
    const int c = ci >> 3, s = ci & 1;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      const int j = 2 * (xi & 1) + s;
      up[((((long)(pn * 8 + (xi >> 1)) * nch + c) * 4 + j) * 64 + lane) * 4 + nt] = u[xi];
    }
  }
}

// V[xi][t][c] = (B^T d B)[xi],  d = 4x4 input patch at (2*th-1, 2*tw-1), zero outside the image
// B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
__global__ __launch_bounds__(256) void winograd_input_kernel(const float* __restrict__ x, float* __restrict__ v, int N, int H,
                                                             int W, int C, int TH, int TW, long Tpad) {
  const int Q = C >> 2;
  const long T = (long)N * TH * TW;
  const long total = T * Q;
  const long plane = Tpad * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int q = (int)(e % Q);
    const long t = e / Q;
    const int tw = (int)(t % TW);
    const int th = (int)((t / TW) % TH);
    const long n = t / ((long)TW * TH);
    const int h0 = 2 * th - 1, w0 = 2 * tw - 1;
    float4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hh = h0 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ww = w0 + j;
        d[i][j] = ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
                      ? ld4(x + ((n * H + hh) * (long)W + ww) * C + 4 * q)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {           // rows: B^T d
      r[0][j] = d[0][j] - d[2][j];
      r[1][j] = d[1][j] + d[2][j];
      r[2][j] = d[2][j] - d[1][j];
      r[3][j] = d[1][j] - d[3][j];
    }
    float* vp = v + t * C + 4 * q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {           // columns: (.) B
      st4(vp + (i * 4 + 0) * plane, r[i][0] - r[i][2]);
      st4(vp + (i * 4 + 1) * plane, r[i][1] + r[i][2]);
      st4(vp + (i * 4 + 2) * plane, r[i][2] - r[i][1]);
      st4(vp + (i * 4 + 3) * plane, r[i][1] - r[i][3]);
    }
  }
}

struct WinoOutArgs {
  const float* m;
  float* y;
  const float* bias;
  const float* scale;
  const float* shift;
  int N, H, W, C, TH, TW;
  long Tpad;
  int act1, act2;
};

// Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]];  y = act2(affine(act1(Y + bias)))
__global__ __launch_bounds__(256) void winograd_output_kernel(const WinoOutArgs a) {
  const int Q = a.C >> 2;
  const long T = (long)a.N * a.TH * a.TW;
  const long total = T * Q;
  const long plane = a.Tpad * a.C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int q = (int)(e % Q);
    const long t = e / Q;
    const int tw = (int)(t % a.TW);
    const int th = (int)((t / a.TW) % a.TH);
    const long n = t / ((long)a.TW * a.TH);
    const float* mp = a.m + t * a.C + 4 * q;
    float4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 m0 = ld4(mp + (0 * 4 + j) * plane), m1 = ld4(mp + (1 * 4 + j) * plane);
      const float4 m2 = ld4(mp + (2 * 4 + j) * plane), m3 = ld4(mp + (3 * 4 + j) * plane);
      s[0][j] = m0 + m1 + m2;
      s[1][j] = m1 - m2 - m3;
    }
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = b4;
    if (a.bias) b4 = ld4(a.bias + 4 * q);
    if (a.scale) { sc = ld4(a.scale + 4 * q); sh = ld4(a.shift + 4 * q); }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oh = 2 * th + i;
      if (oh >= a.H) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ow = 2 * tw + j;
        if (ow >= a.W) continue;
        float4 o = (j == 0) ? s[i][0] + s[i][1] + s[i][2] : s[i][1] - s[i][2] - s[i][3];
        float vv[4] = {o.x + b4.x, o.y + b4.y, o.z + b4.z, o.w + b4.w};
        const float scs[4] = {sc.x, sc.y, sc.z, sc.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float z = shdr::act_apply(vv[k], a.act1);
          if (a.scale) z = z * scs[k] + shs[k];
          vv[k] = shdr::act_apply(z, a.act2);
        }
        st4(a.y + ((n * a.H + oh) * (long)a.W + ow) * a.C + 4 * q, make_float4(vv[0], vv[1], vv[2], vv[3]));
      }
    }
  }
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" int shdr_winograd_filter_f32(const float* w, float* u, int Cin, int Cout, void* stream) {
  SHDR_REQUIRE(w && u, SHDR_E_NULL, "winograd_filter: null pointer");
  SHDR_REQUIRE(Cin > 0 && Cout > 0, SHDR_E_SHAPE, "winograd_filter: bad shape");
  const long cc = (long)Cin * Cout;
  hipLaunchKernelGGL(winograd_filter_kernel, dim3(shdr::stream_grid(cc)), dim3(256), 0, S(stream), w, u, cc);
  return shdr::check_launch("winograd_filter");
}

extern "C" int shdr_winograd_filter_packed_f32(const float* w, float* up, int Cin, int Cout, void* stream) {
  SHDR_REQUIRE(w && up, SHDR_E_NULL, "winograd_filter_packed: null pointer");
  SHDR_REQUIRE(Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 64 == 0, SHDR_E_SHAPE,
               "winograd_filter_packed: need Cin %% 8 == 0 and Cout %% 64 == 0");
  hipLaunchKernelGGL(winograd_filter_packed_kernel, dim3(shdr::stream_grid((long)Cin * Cout)), dim3(256), 0, S(stream), w, up, Cin, Cout);
  return shdr::check_launch("winograd_filter_packed");
}

extern "C" int64_t shdr_winograd_tiles(int N, int H, int W) {
  const int64_t t = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2);
  return (t + 127) / 128 * 128;   // rows of every V / M plane, padded to the 128-row GEMM tile
}

extern "C" int shdr_winograd_input_f32(const float* x, float* v, int N, int H, int W, int C, void* stream) {
  SHDR_REQUIRE(x && v, SHDR_E_NULL, "winograd_input: null pointer");
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, SHDR_E_SHAPE, "winograd_input: need C %% 4 == 0");
  SHDR_REQUIRE(shdr::aligned16(x) && shdr::aligned16(v), SHDR_E_ALIGN, "winograd_input: tensors must be 16-byte aligned");
  const int TH = (H + 1) / 2, TW = (W + 1) / 2;
  const long total = (long)N * TH * TW * (C / 4);
  hipLaunchKernelGGL(winograd_input_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0, S(stream), x, v, N, H, W, C, TH, TW,
                     (long)shdr_winograd_tiles(N, H, W));
  return shdr::check_launch("winograd_input");
}

extern "C" int shdr_winograd_output_f32(const float* m, float* y, const float* bias, const float* scale, const float* shift,
                                        int N, int H, int W, int C, int act1, int act2, void* stream) {
  SHDR_REQUIRE(m && y, SHDR_E_NULL, "winograd_output: null pointer");
  SHDR_REQUIRE((scale == nullptr) == (shift == nullptr), SHDR_E_NULL, "winograd_output: scale and shift come together");
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, SHDR_E_SHAPE, "winograd_output: need C %% 4 == 0");
  SHDR_REQUIRE(shdr::aligned16(m) && shdr::aligned16(y), SHDR_E_ALIGN, "winograd_output: tensors must be 16-byte aligned");
  WinoOutArgs a{m, y, bias, scale, shift, N, H, W, C, (H + 1) / 2, (W + 1) / 2, (long)shdr_winograd_tiles(N, H, W), act1, act2};
  const long total = (long)N * a.TH * a.TW * (C / 4);
  hipLaunchKernelGGL(winograd_output_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0, S(stream), a);
  return shdr::check_launch("winograd_output");
}
