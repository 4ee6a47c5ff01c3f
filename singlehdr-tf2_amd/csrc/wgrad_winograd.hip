// Winograd-domain weight gradient of the 3x3 / stride-1 / SAME convolutions on gfx950 (the backward counterpart of
// winograd_fused.hip): with V = B^T d B (4x4 input patch of a 2x2 output tile) and Q = A dY A^T (the tile's 2x2 output
// gradient), dU[xi][ci][co] = sum over tiles of V[xi][tile][ci] * Q[xi][tile][co] and dW = G^T dU G -- 16 MACs per tile and
// (ci, co) instead of the 36 of the direct form (2.25x fewer MFMAs than wgrad_mfma_kernel).
//
//   block  = 32 input channels x 64 output channels x all 16 transform positions; 256 threads = 4 wavefronts, wave w owns
//            the row combination i = w with all four column combinations (xi = 4w .. 4w+3): 4 x (2 x 4 tiles of 16 x 16) fp32
//            accumulators = 128 registers; TWO independent blocks per CU, so one block's read / transform phase overlaps
//            the other's MFMAs (the 8-wave, one-block-per-CU mapping ran in lockstep: 135-147 TFLOP/s).
//   K loop = Winograd tiles, 8 per chunk (one row segment of 2 x 16 output pixels): the 4 x 18 input patch (32 ci) and the
//            2 x 16 gradient patch (64 co) go HBM/L2 -> LDS by global_load_lds, double buffered; every lane builds its MFMA
//            operands from them on the fly -- A: V[xi][tile = 4s + lane>>4][ci = lane & 15] from 2 patch rows x 3 patch
//            columns, B: Q[xi][tile][co] from the 2 x 2 gradient tile -- so neither V nor Q is ever stored.
//            All LDS reads of a chunk precede the next chunk's DMA issue (see winograd_fused.hip).
//   LDS images: 16 channel quads of 16 bytes per pixel, quad index XOR-swizzled on the DMA source side with the tile column
//            (q ^ 4*((column >> 1) & 3)): the four tile columns of a wave instruction land on four different 64-byte groups.
//   output = partial dU tiles by fp32 atomics (the caller zeroes dU); winograd_dw_kernel then folds G^T dU G into dW.
// Replaces GradientTape.gradient w.r.t. the kernels of the same Conv2D call sites as winograd_fused.hip.
#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) float g_ww_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

struct WinoWgradArgs {
  const float* x;    // [N,H,W,Cx]
  const float* dz;   // [N,H,W,Cout]
  float* du;         // [16][Cx][Cout], accumulated
  int N, H, W, Cx, Cout;
  int tiles_y, segs, units, slice;
  int tiles_m, tiles_n;
};

constexpr int XP_PIX = 4 * 18, ZP_PIX = 2 * 16;
constexpr int Z_INSTR = (ZP_PIX * 16 + 63) / 64;      // 8 wave DMA instructions (32 pixels x 16 quads)
constexpr int Z_FLOATS = Z_INSTR * 256;
template <int MTC>
struct WW {
  static constexpr int XQ = 4 * MTC;                                  // channel quads per patch pixel
  static constexpr int X_INSTR = (XP_PIX * XQ + 63) / 64;             // 18 (64 ci) / 9 (32 ci)
  static constexpr int X_FLOATS = X_INSTR * 256;
  static constexpr int LDS_BYTES = 2 * (X_FLOATS + Z_FLOATS) * 4;
};

template <int MTC>
__global__ __launch_bounds__(256, 2) void wgrad_winograd_kernel(const WinoWgradArgs a) {
  constexpr int XQ = WW<MTC>::XQ, X_INSTR = WW<MTC>::X_INSTR, X_FLOATS = WW<MTC>::X_FLOATS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                          // [2][72 pixels][XQ quads][4]
  float* Zs = smem + 2 * X_FLOATS;           // [2][32 pixels][16 quads][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fg = lane >> 4;
  int t = blockIdx.x;
  const int tn = t % a.tiles_n;
  const int tm = t / a.tiles_n;
  const int ci0 = tm * 16 * MTC, co0 = tn * 64;
  const int u_begin = blockIdx.y * a.slice, u_end = min(u_begin + a.slice, a.units);
  const float* zero = g_ww_zero_page;

  // ---- DMA: instruction j of this wave covers slots (wave + 8*j)*64 + lane; the geometry is recomputed per call (a few
  //      integer ops) instead of being kept in registers across the MFMA loop (it spilled to scratch) -----------------------
  constexpr int XJ = (X_INSTR + 3) / 4, ZJ = (Z_INSTR + 3) / 4;      // per wave (4 waves)
  auto dma_unit = [&](int u, int buf) {
    const int seg = u % a.segs;
    const int q = u / a.segs;
    const int ty = q % a.tiles_y, n = q / a.tiles_y;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      if ((wave + 4 * j) < X_INSTR) {
        const int slot = (wave + 4 * j) * 64 + lane;
        const int pix = slot / XQ, pq = slot % XQ;
        const int row = (pix * 3641) >> 16;              // pix / 18 for pix < 72
        const int col = pix - row * 18;
        const int ih = 2 * ty - 1 + row, iw = 16 * seg - 1 + col;
        const bool ok = pix < XP_PIX && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        const float* p = ok ? a.x + ((size_t)(n * a.H + ih) * a.W + iw) * a.Cx + ci0 + 4 * (pq ^ ((4 * ((col >> 1) & 3)) & (XQ - 1))) : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(Xs + buf * X_FLOATS + (wave + 4 * j) * 256), 16, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < ZJ; ++j) {
      const int slot = (wave + 4 * j) * 64 + lane;       // Z_INSTR == 8: two instructions per wave
      const int pix = slot >> 4, pq = slot & 15;
      const int oh = 2 * ty + (pix >> 4), ow = 16 * seg + (pix & 15);
      const bool ok = oh < a.H && ow < a.W;
      const float* p = ok ? a.dz + ((size_t)(n * a.H + oh) * a.W + ow) * a.Cout + co0 + 4 * (pq ^ (4 * (((pix & 15) >> 1) & 3))) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(Zs + buf * Z_FLOATS + (wave + 4 * j) * 256), 16, 0, 0);
    }
  };

  // ---- operand geometry: wave w owns the row combination i = w and all four column combinations j (xi = 4w .. 4w+3) ------
  const int wi = wave;
  // V row i = d[ra] + sr * d[rb] (B^T): i=0: d0-d2, 1: d1+d2, 2: d2-d1, 3: d1-d3
  const int ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1);
  const int rb = wi == 3 ? 3 : (wi == 2 ? 1 : 2);
  const float sr = wi == 1 ? 1.0f : -1.0f;
  // Q row i = za * dY[0] + zb * dY[1] (A): i=0: dY0, 1: dY0+dY1, 2: dY0-dY1, 3: -dY1
  const float za = wi == 3 ? 0.0f : 1.0f;
  const float zb = wi == 0 ? 0.0f : (wi == 1 ? 1.0f : -1.0f);

  f32x4 acc[4][MTC][4];
#pragma unroll
  for (int x2 = 0; x2 < 4; ++x2)
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[x2][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // (A software-pipelined variant -- operands of chunk c+1 built under the MFMAs of chunk c, three patch buffers, 238
  //  registers -- measured 5 % slower than this plain double-buffered loop.)
  // (Pairing channels so that one ds_read_b64 / ds_read_b128 feeds two / four operand tiles -- 24 LDS instructions per chunk
  //  instead of 64 -- measured 12-25 % SLOWER: 0.89 -> 1.01 ms on 32 x 64^2 x 256 -> 256; the dword reads stay.)
  if (u_begin < u_end) dma_unit(u_begin, 0);
  __syncthreads();
#pragma unroll 1
  for (int u = u_begin; u < u_end; ++u) {
    const int b = (u - u_begin) & 1;
    const float* xb = Xs + b * X_FLOATS;
    const float* zs = Zs + b * Z_FLOATS;
    // ---- operands of both k-steps, built straight from LDS (all LDS reads of the chunk precede the next DMA issue) ----
    float va[2][4][MTC], qb[2][4][4];        // [s][j][mt] / [s][j][nt]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int tx = 4 * s + fg;             // tile column of this lane in k-step s
      float zv[4][2][2];                     // [nt][gradient row][gradient column]
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int col = 2 * tx + c;
          const int swz = 4 * ((col >> 1) & 3);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) zv[nt][r][c] = zs[((r * 16 + col) * 16 + ((nt * 4 + (fi >> 2)) ^ swz)) * 4 + (fi & 3)];
        }
      float xv[MTC][2][4];                   // [mt][row a / b][patch column]
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int pc = 2 * tx + k;           // patch column
        const int swz = (4 * ((pc >> 1) & 3)) & (XQ - 1);
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt) {
          const int q = (mt * 4 + (fi >> 2)) ^ swz;
          xv[mt][0][k] = xb[((ra * 18 + pc) * XQ + q) * 4 + (fi & 3)];
          xv[mt][1][k] = xb[((rb * 18 + pc) * XQ + q) * 4 + (fi & 3)];
        }
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float r0 = za * zv[nt][0][0] + zb * zv[nt][1][0];
        const float r1 = za * zv[nt][0][1] + zb * zv[nt][1][1];
        qb[s][0][nt] = r0;                   // j = 0: dY col 0
        qb[s][1][nt] = r0 + r1;              // j = 1: col0 + col1
        qb[s][2][nt] = r0 - r1;              // j = 2: col0 - col1
        qb[s][3][nt] = -r1;                  // j = 3: -col1
      }
#pragma unroll
      for (int mt = 0; mt < MTC; ++mt) {
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = xv[mt][0][k] + sr * xv[mt][1][k];
        va[s][0][mt] = r[0] - r[2];          // j = 0: c0 - c2
        va[s][1][mt] = r[1] + r[2];          // j = 1: c1 + c2
        va[s][2][mt] = r[2] - r[1];          // j = 2: c2 - c1
        va[s][3][mt] = r[1] - r[3];          // j = 3: c1 - c3
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < u_end) dma_unit(u + 1, b ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int x2 = 0; x2 < 4; ++x2)
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[x2][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[s][x2][mt], qb[s][x2][nt], acc[x2][mt][nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  // ---- partial dU tiles: D[row = ci (4*fg + r)][col = co (fi)] --------------------------------------------------------------
#pragma unroll
  for (int x2 = 0; x2 < 4; ++x2) {
    float* dst = a.du + (size_t)(4 * wave + x2) * a.Cx * a.Cout;
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          atomicAdd(dst + (size_t)(ci0 + mt * 16 + 4 * fg + r) * a.Cout + co0 + nt * 16 + fi, acc[x2][mt][nt][r]);
  }
}

// dW[a][b][ci_off + ci][co] += scale * sum_{i,j} G[i][a] dU[4i + j][ci][co] G[j][b],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(256) void winograd_dw_kernel(const float* __restrict__ du, float* __restrict__ dw, int Cx, int Cout,
                                                          int Ct, int ci_off, float scale) {
  const long cc = (long)Cx * Cout;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < cc; e += (long)gridDim.x * 256) {
    const int co = (int)(e % Cout), ci = (int)(e / Cout);
    float u[4][4], t[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) u[i][j] = du[(i * 4 + j) * cc + e];
#pragma unroll
    for (int j = 0; j < 4; ++j) {            // G^T u
      t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
      t[1][j] = 0.5f * (u[1][j] - u[2][j]);
      t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
#pragma unroll
    for (int aa = 0; aa < 3; ++aa) {         // (.) G
      const float w0 = t[aa][0] + 0.5f * (t[aa][1] + t[aa][2]);
      const float w1 = 0.5f * (t[aa][1] - t[aa][2]);
      const float w2 = 0.5f * (t[aa][1] + t[aa][2]) + t[aa][3];
      float* o = dw + ((size_t)(aa * 3) * Ct + ci_off + ci) * Cout + co;
      o[0] += scale * w0;
      o[(size_t)Ct * Cout] += scale * w1;
      o[(size_t)2 * Ct * Cout] += scale * w2;
    }
  }
}

}  // namespace

extern "C" int shdr_conv2d_wgrad_winograd_f32(const float* x, const float* dz, float* du, float* dw, int N, int H, int W, int Cx,
                                              int Cout, int Ct, int ci_off, float x_scale, void* stream) {
  SHDR_REQUIRE(x && dz && du && dw, SHDR_E_NULL, "wgrad_winograd: null pointer");
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && Cx > 0 && Cout > 0 && Ct >= Cx && ci_off >= 0 && ci_off + Cx <= Ct, SHDR_E_SHAPE,
               "wgrad_winograd: bad shape");
  SHDR_REQUIRE(Cx % 32 == 0 && Cout % 64 == 0, SHDR_E_SHAPE, "wgrad_winograd: need Cin %% 32 == 0 and Cout %% 64 == 0");
  SHDR_REQUIRE(shdr::aligned16(x) && shdr::aligned16(dz), SHDR_E_ALIGN, "wgrad_winograd: x and dz must be 16-byte aligned");
  SHDR_REQUIRE((long)N * H * W * Cx < (1L << 32) && (long)N * H * W * Cout < (1L << 32), SHDR_E_SHAPE,
               "wgrad_winograd: tensor with more than 2^32 elements");
  // du is scratch of this call: zeroed here, in stream order (the caller hands over uninitialised memory)
  if (hipMemsetAsync(du, 0, sizeof(float) * 16 * (size_t)Cx * Cout, reinterpret_cast<hipStream_t>(stream)) != hipSuccess)
    return shdr::fail(SHDR_E_LAUNCH, "wgrad_winograd: memset of the dU scratch failed");
  WinoWgradArgs a{};
  a.x = x; a.dz = dz; a.du = du;
  a.N = N; a.H = H; a.W = W; a.Cx = Cx; a.Cout = Cout;
  a.tiles_y = (H + 1) / 2;
  a.segs = (W + 15) / 16;
  a.units = N * a.tiles_y * a.segs;
  // 32 input channels per block: a 64-channel tile (128 accumulator registers) leaves no room for the pipelined operands and
  // spills (measured 10 % slower even without the pipeline)
  a.tiles_m = Cx / 32;
  a.tiles_n = Cout / 64;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static long slots_of[shdr::kMaxDevices] = {};
  const int dev_slot = shdr::device_slot();
  if (slots_of[dev_slot] == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_winograd_kernel<2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WW<2>::LDS_BYTES);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    slots_of[dev_slot] = shdr::block_slots(wgrad_winograd_kernel<2>, 256, WW<2>::LDS_BYTES);
    if (slots_of[dev_slot] == 0) return shdr::fail(SHDR_E_ARCH, "wgrad_winograd: occupancy query failed");
  }
  // unit slices: one round of the chip's block slots, >= 32 units (256 tiles) per block to bound the atomics
  const long tiles = (long)a.tiles_m * a.tiles_n;
  long slice = shdr::slice_for_rounds(slots_of[dev_slot], tiles, a.units, 32);
  if (SHDR_ENV("SHDR_WGRAD_LEGACY_GRID")) {
    const long want = (768 + tiles - 1) / tiles;
    slice = (a.units + want - 1) / want;
    if (slice < 32) slice = 32;
  }
  a.slice = (int)slice;
  const long nslices = (a.units + slice - 1) / slice;
  SHDR_REQUIRE(nslices <= 65535, SHDR_E_SHAPE, "wgrad_winograd: too many unit slices");
  hipLaunchKernelGGL(wgrad_winograd_kernel<2>, dim3((unsigned)tiles, (unsigned)nslices), dim3(256), WW<2>::LDS_BYTES, st, a);
  if (int rc = shdr::check_launch("wgrad_winograd_kernel")) return rc;
  hipLaunchKernelGGL(winograd_dw_kernel, dim3(shdr::stream_grid((long)Cx * Cout)), dim3(256), 0, st, du, dw, Cx, Cout, Ct, ci_off,
                     x_scale);
  return shdr::check_launch("winograd_dw_kernel");
}
