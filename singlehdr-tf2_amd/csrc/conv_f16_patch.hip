// Narrow-layer fp16 convolution (BASELINE configs[4]): the full-resolution layers of the Dequantization- / Refinement-Net U-Nets
// (7x7 8->16 and 16->16, 5x5 16->32 and 32->32, 3x3 32->16, 16+16->16, 16->16 ...) and their input gradients.
//
// With <= 32 input channels per tap the implicit-GEMM kernel (conv_f16.hip) stages the im2col rows tap by tap: every input pixel
// travels through the LDS-DMA path once per tap (49 times for 7x7) and the kernel is bound by that feed (60-160 TFLOP/s).  Here
// a block keeps the WHOLE filter in LDS for its lifetime (persistent blocks) and stages the raw input PATCH of a 16 x 16 pixel
// tile (+ halo) once; the MFMA operand of a lane -- 8 consecutive channels of one pixel at one tap -- is read straight from the
// patch with ds_read_b128 at the tap's offset.  One v_mfma_f32_16x16x32_f16 covers 32 / CT taps (CT = channels per pixel: 8, 16,
// 32, or 16 + 16 from two concatenated sources); the k order is the natural (tap, channel) order of conv_pack_filter_f16_kernel.
// Stride 1, SAME padding, odd square filters.  Epilogue as conv_f16.hip (bias + activation, fp16 through LDS, or an fp32 head).
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) unsigned g_p_zero_page[4] = {0u, 0u, 0u, 0u};

struct PatchArgs {
  const _Float16* x1;
  const _Float16* x2;
  const _Float16* wp;      // packed [nsteps][Cout][32], natural k order
  const float* bias;
  _Float16* y16;
  float* y32;
  int N, H, W, Cout, tiles_x, tiles_y, ntiles, act1, cout_valid;
};

__host__ __device__ inline int pswz(int row) { return (-(row >> 2)) & 3; }

template <int KK, int CT, int NT, bool TWO>
struct PG {
  static constexpr int TPP = 32 / CT;                          // taps per MFMA k-step
  static constexpr int NTAPS = KK * KK;
  static constexpr int NS = (NTAPS + TPP - 1) / TPP;           // k-steps
  static constexpr int PW = 16 + KK - 1, PH = 16 + KK - 1;
  static constexpr int PP = PW * PH;                           // patch pixels
  static constexpr int CS = TWO ? 16 : CT;                     // channels per pixel of ONE source tensor
  static constexpr int PIECES = PP * CT / 8;                   // 16-byte pieces of the patch (both sources)
  static constexpr int PINSTR = (PIECES + 63) / 64;            // wave DMA instructions per patch
  static constexpr int PJ = (PINSTR + 3) / 4;                  // per wave
  static constexpr int PATCH_HALVES = PJ * 4 * 512;            // rounded up to whole instructions
  static constexpr int COUT = NT * 16;
  static constexpr int FILT_HALVES = NS * COUT * 32;
  static constexpr int FINSTR = NS * COUT / 16;                // filter DMA instructions (16 rows of 64 bytes each)
  static constexpr int STAGE_HALVES = 256 * (COUT + 8);
  static constexpr int LDS_HALVES = FILT_HALVES + 2 * PATCH_HALVES > STAGE_HALVES ? FILT_HALVES + 2 * PATCH_HALVES : STAGE_HALVES;
};

template <int KK, int CT, int NT, bool TWO>
__global__ __launch_bounds__(256) void conv_f16_patch_kernel(const PatchArgs a) {
  using G = PG<KK, CT, NT, TWO>;
  constexpr int MT = 4;                                        // wave w owns tile rows 4w .. 4w+3
  constexpr int PAD = (KK - 1) / 2;
  extern __shared__ __attribute__((aligned(16))) _Float16 psm[];
  _Float16* filt = psm;                                        // [NS][COUT][32], swizzled rows
  _Float16* patch = psm + G::FILT_HALVES;                      // [2][PATCH_HALVES]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const _Float16* zero = reinterpret_cast<const _Float16*>(g_p_zero_page);

  // ---- filter -> LDS, once per block (rows of 64 bytes, physical slot = k-group ^ swz(row)) ----------------------------------
  for (int j = wave; j < G::FINSTR; j += 4) {
    const int r = j * 16 + (lane >> 2);                        // row = step * COUT + cout
    const int co = r % G::COUT;
    const _Float16* p = a.wp + (size_t)r * 32 + 8 * ((lane & 3) ^ pswz(co));
    __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(filt + j * 512), 16, 0, 0);
  }

  // ---- patch DMA geometry: piece -> (source, patch pixel, channel group), fixed over the tiles ---------------------------------
  int ppy[G::PJ], ppx[G::PJ], pch[G::PJ], psrc[G::PJ];
  bool pok[G::PJ];
#pragma unroll
  for (int j = 0; j < G::PJ; ++j) {
    const int piece = (wave * G::PJ + j) * 64 + lane;
    pok[j] = piece < G::PIECES;
    int pix, cg, src = 0;
    if (TWO) {                                                 // image [source][pixel][16 channels]
      src = piece / (G::PP * 2);
      const int rem = piece - src * G::PP * 2;
      pix = rem >> 1;
      cg = rem & 1;
    } else {
      pix = piece / (CT / 8);
      cg = piece - pix * (CT / 8);
      if (CT == 32) cg ^= pswz(pix);                           // 64-byte pixels: the b128 bank swizzle, applied on the source side
    }
    psrc[j] = src;
    ppy[j] = pix / G::PW;
    ppx[j] = pix - ppy[j] * G::PW;
    pch[j] = 8 * cg;
  }
  auto dma_patch = [&](int tile, int buf) {
    int pm = tile;
    const int tx = pm % a.tiles_x;
    pm /= a.tiles_x;
    const int ty = pm % a.tiles_y;
    const int img = pm / a.tiles_y;
    const int ih0 = ty * 16 - PAD, iw0 = tx * 16 - PAD;
#pragma unroll
    for (int j = 0; j < G::PJ; ++j) {
      const int ih = ih0 + ppy[j], iw = iw0 + ppx[j];
      const bool ok = pok[j] && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
      const _Float16* src = (TWO && psrc[j]) ? a.x2 : a.x1;
      const _Float16* p = ok ? src + ((size_t)(img * a.H + ih) * a.W + iw) * G::CS + pch[j] : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(patch + buf * G::PATCH_HALVES + (wave * G::PJ + j) * 512), 16, 0, 0);
    }
  };

  // ---- operand geometry -----------------------------------------------------------------------------------------------------
  const int fi = lane & 15, fg = lane >> 4;
  constexpr int GPT = TWO ? 2 : 4 / G::TPP;                    // k-groups (8 channels) per tap (TWO: per source)
  const int jl = fg / GPT;                                     // which tap of the k-step this lane's k-group belongs to (TWO: which source)
  const int cgl = fg % GPT;                                    // which 8-channel group of the pixel
  int a_base[MT];                                              // half offset of (tile row, pixel fi, tap 0) in the patch image
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int prow = wave * MT + mi;
    if (TWO) a_base[mi] = (jl ? G::PP * 16 : 0) + (prow * G::PW + fi) * 16 + 8 * cgl;      // here jl = source, one tap per step
    else a_base[mi] = (prow * G::PW + fi) * CT + 8 * cgl;
  }
  int b_rd[NT];
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    const int row = ni * 16 + fi;
    b_rd[ni] = row * 32 + 8 * (fg ^ pswz(row));
  }

  // the bias is the same for every tile of the persistent block: loaded once (inside the epilogue each load is followed by
  // "s_waitcnt vmcnt(0)", which sat out the round trip of the output store before it and of the next tile's patch DMA)
  f32x4 bias_r[NT];
#pragma unroll
  for (int ni = 0; ni < NT; ++ni)
    bias_r[ni] = (a.bias && !a.y32) ? *reinterpret_cast<const f32x4*>(a.bias + ni * 16 + 4 * fg) : (f32x4){0.f, 0.f, 0.f, 0.f};
  int tile = blockIdx.x;
  if (tile < a.ntiles) dma_patch(tile, 0);
  int buf = 0;
  // The barrier of the tile loop waits for this wave's patch DMA only: the DMA of patch(tile) is OLDER than the output stores of the
  // previous tile (vector-memory operations retire in issue order), so "vmcnt(stores issued since)" lets those stores stay in flight;
  // __syncthreads() is a full fence (vmcnt(0)) and sat out their round trip once per tile.
  int stores_since = -1;                                        // -1: unknown -> drain
  for (; tile < a.ntiles; tile += gridDim.x, buf ^= 1) {
    if (stores_since == MT * NT) __builtin_amdgcn_s_waitcnt(0x0F70 | ((MT * NT) & 15) | (((MT * NT) >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // patch(tile) (and the filter) landed; everyone is done with buf ^ 1
    const int next = tile + gridDim.x;
    if (next < a.ntiles) dma_patch(next, buf ^ 1);
    const _Float16* P = patch + buf * G::PATCH_HALVES;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < G::NS; ++s) {
      // tap offset of this lane's k-group: tap = s * TPP + jl (clamped into the filter: the packed filter is zero beyond it)
      int toff = 0;
      if (TWO) {
        toff = ((s / KK) * G::PW + (s % KK)) * 16;
      } else {
#pragma unroll
        for (int j = 0; j < G::TPP; ++j) {
          const int t = s * G::TPP + j < G::NTAPS ? s * G::TPP + j : 0;
          const int o = ((t / KK) * G::PW + (t % KK)) * CT;
          toff = (jl == j) ? o : toff;
        }
      }
      f16x8 wb[NT], pa[MT];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) wb[ni] = *reinterpret_cast<const f16x8*>(filt + s * G::COUT * 32 + b_rd[ni]);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        int ad = a_base[mi] + toff;
        if (!TWO && CT == 32) {                                // undo the source-side swizzle of the 64-byte pixel
          const int pix = ad >> 5;
          ad = (pix << 5) + 8 * (cgl ^ pswz(pix));
        }
        pa[mi] = *reinterpret_cast<const f16x8*>(P + ad);
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], pa[mi], acc[mi][ni], 0, 0, 0);
    }

    // ---- epilogue: lane (fi, fg) holds couts 4fg..4fg+3 of pixel (row wave*4 + mi, column fi) per 16-cout tile ----------------
    int pm = tile;
    const int tx = pm % a.tiles_x;
    pm /= a.tiles_x;
    const int ty = pm % a.tiles_y;
    const int img = pm / a.tiles_y;
    const int oh0 = ty * 16, ow0 = tx * 16;
    stores_since = (!a.y32 && oh0 + wave * MT + MT <= a.H) ? MT * NT : -1;      // one 8-byte store per (row, cout tile) of this wave
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int oh = oh0 + wave * MT + mi, ow = ow0 + fi;
      if (oh >= a.H || ow >= a.W) continue;
      const size_t pix = ((size_t)img * a.H + oh) * a.W + ow;
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        const int co = ni * 16 + 4 * fg;
        f32x4 v = acc[mi][ni];
        if (a.y32) {                                             // the fp32 heads (tanh lives here only: shdr_internal.h act_apply4)
          if (co >= a.cout_valid) continue;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (a.bias && co + e < a.cout_valid) v[e] += a.bias[co + e];
          shdr::act_apply4<1>(v, a.act1);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < a.cout_valid) a.y32[pix * a.cout_valid + co + e] = v[e];
        } else {
          v += bias_r[ni];
          shdr::act_apply4<0>(v, a.act1);                    // (fp16 outputs with tanh go to the general kernel: shdr_conv2d_fwd_f16)
          f16x4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];
          // 16 lanes x 8 bytes at a stride of Cout * 2 bytes: with 16 / 32 couts the four lane groups of a pixel complete its
          // 32- / 64-byte row in one instruction
          *reinterpret_cast<f16x4*>(a.y16 + pix * a.Cout + co) = h;
        }
      }
    }
  }
}

template <int KK, int CT, int NT, bool TWO>
int launch_patch(PatchArgs& a, hipStream_t st) {
  using G = PG<KK, CT, NT, TWO>;
  constexpr int lds = G::LDS_HALVES * 2;
  if constexpr (lds > 160 * 1024) {
    return shdr::fail(SHDR_E_SHAPE, "conv2d_patch_f16: filter + patches (%d bytes) do not fit the LDS", lds);
  } else {
  const int dev_slot = shdr::device_slot();
  static bool attr_done[shdr::kMaxDevices] = {};
  static int occ[shdr::kMaxDevices] = {};
  if (!attr_done[dev_slot]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16_patch_kernel<KK, CT, NT, TWO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_f16_patch_kernel<KK, CT, NT, TWO>), 256, lds);
    occ[dev_slot] = (e != hipSuccess || nb < 1) ? 1 : (nb > 4 ? 4 : nb);
    attr_done[dev_slot] = true;
  }
  long grid = 256L * occ[dev_slot];
  if (grid > a.ntiles) grid = a.ntiles;
  hipLaunchKernelGGL((conv_f16_patch_kernel<KK, CT, NT, TWO>), dim3((unsigned)grid), dim3(256), lds, st, a);
  return shdr::check_launch("conv_f16_patch_kernel");
  }
}

template <int KK, int NT>
int dispatch_ct(PatchArgs& a, int C1, int C2, hipStream_t st) {
  if (C2 == 16 && C1 == 16) return launch_patch<KK, 32, NT, true>(a, st);
  if (C1 == 8) return launch_patch<KK, 8, NT, false>(a, st);
  if (C1 == 16) return launch_patch<KK, 16, NT, false>(a, st);
  return launch_patch<KK, 32, NT, false>(a, st);
}

}  // namespace

// 1 if the patch kernel takes the layer (the dispatcher of shdr_conv2d_fwd_f16 asks first)
extern "C" int shdr_conv2d_patch_ok_f16(const shdr_conv2d_desc* d) {
  if (!d || d->stride != 1 || d->KH != d->KW || !(d->KH == 3 || d->KH == 5 || d->KH == 7)) return 0;
  if (d->pad_t != (d->KH - 1) / 2 || d->pad_l != (d->KW - 1) / 2 || d->Ho != d->H || d->Wo != d->W) return 0;
  if (!(d->Cout == 16 || d->Cout == 32)) return 0;
  const bool one = d->C2 == 0 && (d->C1 == 8 || d->C1 == 16 || d->C1 == 32), two = d->C1 == 16 && d->C2 == 16;
  if (!(one || two)) return 0;
  // filter + two patches within the LDS of a CU (7x7 with 32 channels per pixel does not fit: 100 + 2 x 31 KB)
  const long filt = (long)((d->KH * d->KW * (d->C1 + d->C2) + 31) / 32) * d->Cout * 64;
  const long patch = (long)(16 + d->KH - 1) * (16 + d->KW - 1) * (d->C1 + d->C2) * 2 + 4096;
  return filt + 2 * patch <= 150 * 1024 ? 1 : 0;
}

extern "C" int shdr_conv2d_fwd_patch_f16(const shdr_conv2d_desc* d, const void* x1, const void* x2, const void* wp, const float* bias,
                                         void* y, int y_is_f32, void* stream) {
  SHDR_REQUIRE(d && x1 && wp && y, SHDR_E_NULL, "conv2d_patch_f16: null desc/x1/wp/y");
  SHDR_REQUIRE(shdr_conv2d_patch_ok_f16(d), SHDR_E_SHAPE, "conv2d_patch_f16: layer shape not taken by the patch kernel");
  SHDR_REQUIRE(y_is_f32 || d->act1 != SHDR_ACT_TANH, SHDR_E_SHAPE, "conv2d_patch_f16: tanh is compiled into the fp32-output (head) path only");
  SHDR_REQUIRE((d->C2 == 0) == (x2 == nullptr), SHDR_E_NULL, "conv2d_patch_f16: x2 must be given iff C2 > 0");
  const int cout_valid = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  SHDR_REQUIRE(cout_valid <= d->Cout && (y_is_f32 || cout_valid == d->Cout), SHDR_E_SHAPE, "conv2d_patch_f16: bad cout_valid");
  SHDR_REQUIRE((long)d->N * d->H * d->W * 32 < (1L << 32), SHDR_E_SHAPE, "conv2d_patch_f16: tensor too large");
  SHDR_REQUIRE(shdr::aligned16(x1) && (!x2 || shdr::aligned16(x2)) && shdr::aligned16(wp) && shdr::aligned16(y) && (!bias || shdr::aligned16(bias)),
               SHDR_E_ALIGN, "conv2d_patch_f16: tensors must be 16-byte aligned");
  PatchArgs a{};
  a.x1 = reinterpret_cast<const _Float16*>(x1);
  a.x2 = reinterpret_cast<const _Float16*>(x2);
  a.wp = reinterpret_cast<const _Float16*>(wp);
  a.bias = bias;
  a.y16 = y_is_f32 ? nullptr : reinterpret_cast<_Float16*>(y);
  a.y32 = y_is_f32 ? reinterpret_cast<float*>(y) : nullptr;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.tiles_x = (d->W + 15) / 16;
  a.tiles_y = (d->H + 15) / 16;
  a.ntiles = a.N * a.tiles_x * a.tiles_y;
  a.act1 = d->act1;
  a.cout_valid = cout_valid;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->Cout == 16) {
    if (d->KH == 3) return dispatch_ct<3, 1>(a, d->C1, d->C2, st);
    if (d->KH == 5) return dispatch_ct<5, 1>(a, d->C1, d->C2, st);
    return dispatch_ct<7, 1>(a, d->C1, d->C2, st);
  }
  if (d->KH == 3) return dispatch_ct<3, 2>(a, d->C1, d->C2, st);
  if (d->KH == 5) return dispatch_ct<5, 2>(a, d->C1, d->C2, st);
  return dispatch_ct<7, 2>(a, d->C1, d->C2, st);
}
