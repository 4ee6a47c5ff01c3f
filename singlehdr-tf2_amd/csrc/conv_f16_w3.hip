// Wide 3x3 / stride-1 fp16 convolution (BASELINE configs[4]): the layers of the Hallucination-Net and of the U-Net / ResNet trunks with
// >= 32 input channels (a multiple of 32 per source) and a multiple of 64 output channels -- forward and input gradient.
//
// The implicit-GEMM kernel (conv_f16.hip) stages the im2col rows tap by tap: every input pixel goes through the LDS-DMA path nine
// times and the kernel is bound by that L2 -> LDS feed (410-940 TFLOP/s).  Here a block owns a 16 x 16 pixel tile x 64 couts and
// stages, per 32-channel chunk, the raw 18 x 18 PATCH once (20.7 KB instead of 9 x 16 KB); the MFMA operand of a lane -- 8 channels
// of one pixel -- is read from the patch at the tap's offset (ds_read_b128).  The filter streams in (chunk, filter row) units of
// 3 taps x 64 couts x 32 channels (12 KB), double-buffered like the patch: one barrier per unit of 48 MFMAs per wave.
// Packed filter and k order as conv_f16.hip (FAST order: kc = chunk * 9 + tap).  Two blocks per CU (72 KB of LDS each).
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) unsigned g_w3_zero_page[4] = {0u, 0u, 0u, 0u};

struct W3Args {
  const _Float16* x1;
  const _Float16* x2;
  const _Float16* wp;      // packed [Ct/32 * 9][Cout][32]
  const float* bias;
  _Float16* y;
  int N, H, W, C1, C2, Cout, tiles_x, tiles_y, nblk_m, nblk_n, act1;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__host__ __device__ inline int f4(int row) { return (-(row >> 2)) & 3; }      // filter image: rows = couts, 16-row aligned fragments
// patch image: column-keyed slot swizzle + lane -> tile-column permutation (conv_x3.hip: a ds_read_b128 lane group = 8 lanes of one
// channel group + 8 of its neighbour; with pcol() each set reads 8 consecutive columns -> conflict-free at every tap shift, and the
// operand address is a per-lane column term + a scalar row term)
__host__ __device__ inline int sx(int col) { return ((col >> 2) & 1) * 2; }
__device__ __forceinline__ int pcol(int fi) { return fi < 4 ? fi : (fi >= 12 ? fi - 8 : fi + 4); }

constexpr int PWID = 18, PPIX = PWID * PWID;                   // raw patch of a 16 x 16 tile
constexpr int PJ = 6;                                          // patch DMA instructions per wave (24 x 64 pieces >= 1296)
constexpr int PATCH_HALVES = 4 * PJ * 512;
constexpr int BN = 64, NT = 4, MT = 4;
constexpr int FJ = 3;                                          // filter DMA instructions per wave and unit (3 taps x 64 rows / 16 / 4)
constexpr int FILT_HALVES = 3 * BN * 32;
constexpr int W3_LDS_BYTES = (2 * PATCH_HALVES + 2 * FILT_HALVES) * 2;

__global__ __launch_bounds__(256, 2) void conv_f16_w3_kernel(const W3Args a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 wsm[];
  _Float16* patch = wsm;                                       // [2][PATCH_HALVES]
  _Float16* filt = wsm + 2 * PATCH_HALVES;                     // [2][FILT_HALVES]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = xcd_remap(blockIdx.x, a.nblk_m * a.nblk_n);
  const int pn = L % a.nblk_n;
  int pm = L / a.nblk_n;
  const int tx = pm % a.tiles_x;
  pm /= a.tiles_x;
  const int ty = pm % a.tiles_y;
  const int img = pm / a.tiles_y;
  const int n0 = pn * BN, oh0 = ty * 16, ow0 = tx * 16;
  const _Float16* zero = reinterpret_cast<const _Float16*>(g_w3_zero_page);

  // ---- patch DMA geometry (fixed per block): piece -> (patch pixel, physical slot) -------------------------------------------
  unsigned poff1[PJ], poff2[PJ];
  bool pok[PJ];
#pragma unroll
  for (int j = 0; j < PJ; ++j) {
    const int piece = (wave * PJ + j) * 64 + lane;
    const int pix = piece >> 2, slot = piece & 3;
    const int py = pix / PWID, px = pix - py * PWID;
    const int ih = oh0 - 1 + py, iw = ow0 - 1 + px;
    pok[j] = pix < PPIX && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    const unsigned p = pok[j] ? (unsigned)((img * a.H + ih) * a.W + iw) : 0u;
    const unsigned kg = (unsigned)(slot ^ sx(px));
    poff1[j] = p * (unsigned)a.C1 + 8u * kg;
    poff2[j] = p * (unsigned)a.C2 + 8u * kg;
  }
  // ---- filter DMA geometry: instruction (wave * 3 + j) covers 16 rows of the [3 kw][64 couts] unit image -----------------------
  unsigned foff[FJ];
#pragma unroll
  for (int j = 0; j < FJ; ++j) {
    const int R = (wave * FJ + j) * 16 + (lane >> 2);
    const int kw = R >> 6, co = R & 63;
    foff[j] = (unsigned)(kw * a.Cout + n0 + co) * 32u + 8u * (unsigned)((lane & 3) ^ f4(co));
  }
  const int nch = (a.C1 + a.C2) >> 5;
  const int nunits = nch * 3;
  auto dma_patch = [&](int c, int buf) {
    const int c0 = c << 5;
    const bool second = c0 >= a.C1;
    const _Float16* src = second ? a.x2 : a.x1;
    const unsigned cc = (unsigned)(second ? c0 - a.C1 : c0);
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      const _Float16* p = pok[j] ? src + (size_t)((second ? poff2[j] : poff1[j]) + cc) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(patch + buf * PATCH_HALVES + (wave * PJ + j) * 512), 16, 0, 0);
    }
  };
  auto dma_filt = [&](int unit, int buf) {                       // unit = chunk * 3 + kh  ->  packed chunks unit * 3 + {0, 1, 2}
    const _Float16* base = a.wp + (size_t)unit * 3 * a.Cout * 32;
#pragma unroll
    for (int j = 0; j < FJ; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + foff[j]), (lptr_t)(filt + buf * FILT_HALVES + (wave * FJ + j) * 512), 16, 0, 0);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fg = lane >> 4;
  int acol[3], b_rd[NT];
  const int pc = pcol(fi);
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) acol[kw] = (pc + kw) * 32 + 8 * (fg ^ sx(pc + kw));
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    const int row = ni * 16 + fi;
    b_rd[ni] = row * 32 + 8 * (fg ^ f4(row));
  }

  dma_patch(0, 0);
  dma_filt(0, 0);
#pragma unroll 1
  for (int u = 0; u < nunits; ++u) {
    const int c = u / 3, kh = u - 3 * c;
    __syncthreads();                                           // unit u's filter (and chunk c's patch) landed; buffers of u - 1 are free
    if (u + 1 < nunits) {
      dma_filt(u + 1, (u + 1) & 1);
      if (kh == 0 && c + 1 < nch) dma_patch(c + 1, (c + 1) & 1);
    }
    const _Float16* P = patch + (c & 1) * PATCH_HALVES;
    const _Float16* F = filt + (u & 1) * FILT_HALVES;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      f16x8 wb[NT], pa[MT];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) wb[ni] = *reinterpret_cast<const f16x8*>(F + kw * BN * 32 + b_rd[ni]);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) pa[mi] = *reinterpret_cast<const f16x8*>(P + ((wave * MT + mi + kh) * PWID) * 32 + acol[kw]);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], pa[mi], acc[mi][ni], 0, 0, 0);
    }
  }
  __syncthreads();

  // ---- epilogue: bias + activation in fp32 -> fp16 through LDS -> 16-byte row-contiguous stores (as conv_f16.hip) ----------------
  constexpr int RS = BN + 8;
  _Float16* stage = wsm;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int r = (wave * MT + mi) * 16 + pc;                    // the lane's tile column (pcol permutation)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int cl = ni * 16 + 4 * fg;
      f32x4 v = acc[mi][ni];
      if (a.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n0 + cl);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
      shdr::act_apply4<0>(v, a.act1);                      // (no tanh here: shdr_conv2d_w3_ok_f16; shdr_internal.h act_apply4)
      f16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];
      *reinterpret_cast<f16x4*>(stage + r * RS + cl) = h;
    }
  }
  __syncthreads();
  constexpr int QR = BN / 8;
#pragma unroll 2
  for (int e = tid; e < 256 * QR; e += 256) {
    const int r = e / QR, q = e - r * QR;
    const int oh = oh0 + (r >> 4), ow = ow0 + (r & 15);
    if (oh >= a.H || ow >= a.W) continue;
    const size_t pix = ((size_t)img * a.H + oh) * a.W + ow;
    *reinterpret_cast<f16x8*>(a.y + pix * a.Cout + n0 + 8 * q) = *reinterpret_cast<const f16x8*>(stage + r * RS + 8 * q);
  }
}

}  // namespace

extern "C" int shdr_conv2d_w3_ok_f16(const shdr_conv2d_desc* d) {
  if (!d || d->stride != 1 || d->KH != 3 || d->KW != 3 || d->pad_t != 1 || d->pad_l != 1 || d->Ho != d->H || d->Wo != d->W) return 0;
  if (d->C1 % 32 || d->C2 % 32 || d->Cout % 64 || d->act1 == SHDR_ACT_TANH) return 0;      // (tanhf is not compiled into this kernel)
  const int cv = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  if (cv != d->Cout) return 0;
  // enough blocks to fill the chip: the deepest, smallest maps stay on the 128 x 128 implicit-GEMM tiles
  const long blocks = (long)d->N * ((d->H + 15) / 16) * ((d->W + 15) / 16) * (d->Cout / 64);
  long min_blocks = 384;
  if (const char* e = SHDR_ENV("SHDR_W3_MIN_BLOCKS")) min_blocks = atol(e);
  return blocks >= min_blocks ? 1 : 0;
}

extern "C" int shdr_conv2d_fwd_w3_f16(const shdr_conv2d_desc* d, const void* x1, const void* x2, const void* wp, const float* bias, void* y,
                                      void* stream) {
  SHDR_REQUIRE(d && x1 && wp && y, SHDR_E_NULL, "conv2d_w3_f16: null desc/x1/wp/y");
  SHDR_REQUIRE(shdr_conv2d_w3_ok_f16(d), SHDR_E_SHAPE, "conv2d_w3_f16: layer shape not taken by this kernel");
  SHDR_REQUIRE((d->C2 == 0) == (x2 == nullptr), SHDR_E_NULL, "conv2d_w3_f16: x2 must be given iff C2 > 0");
  SHDR_REQUIRE((long)d->N * d->H * d->W * (d->C1 > d->C2 ? d->C1 : d->C2) < (1L << 32), SHDR_E_SHAPE, "conv2d_w3_f16: tensor too large");
  SHDR_REQUIRE(shdr::aligned16(x1) && (!x2 || shdr::aligned16(x2)) && shdr::aligned16(wp) && shdr::aligned16(y) && (!bias || shdr::aligned16(bias)),
               SHDR_E_ALIGN, "conv2d_w3_f16: tensors must be 16-byte aligned");
  W3Args a{};
  a.x1 = reinterpret_cast<const _Float16*>(x1);
  a.x2 = reinterpret_cast<const _Float16*>(x2 ? x2 : x1);
  a.wp = reinterpret_cast<const _Float16*>(wp);
  a.bias = bias;
  a.y = reinterpret_cast<_Float16*>(y);
  a.N = d->N; a.H = d->H; a.W = d->W; a.C1 = d->C1; a.C2 = d->C2; a.Cout = d->Cout;
  a.tiles_x = (d->W + 15) / 16;
  a.tiles_y = (d->H + 15) / 16;
  a.nblk_m = a.N * a.tiles_x * a.tiles_y;
  a.nblk_n = a.Cout / 64;
  a.act1 = d->act1;
  static bool attr_done[shdr::kMaxDevices] = {};
  const int dev_slot = shdr::device_slot();
  if (!attr_done[dev_slot]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16_w3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS_BYTES);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done[dev_slot] = true;
  }
  const long nblk = (long)a.nblk_m * a.nblk_n;
  if (nblk > 0x7fffffffL) return shdr::fail(SHDR_E_SHAPE, "conv2d_w3_f16: grid of %ld blocks", nblk);
  hipLaunchKernelGGL(conv_f16_w3_kernel, dim3((unsigned)nblk), dim3(256), W3_LDS_BYTES, reinterpret_cast<hipStream_t>(stream), a);
  return shdr::check_launch("conv_f16_w3_kernel");
}
