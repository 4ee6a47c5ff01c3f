// Narrow fp32 convolutions on the fp16 matrix pipe (the split-operand arithmetic of conv_x3.hip: x = xh + xl 2^-11, w 2^S = wh + wl,
// x w = xh wh + xl (wh 2^-11) + xh wl, fp32 accumulation) for the full-resolution layers of the Dequantization- / Refinement-Net
// U-Nets (dequantization_net.py:8-9,21-22,27,35-36,46,62-63, refinement_net.py): 7x7 3(4) -> 16 and 16 -> 16, 5x5 16 -> 32 and
// 32 -> 32, 3x3 32 -> 16, 16 + 16 -> 16 (tf.concat), 16 -> 3 (tanh + residual).  With 16 couts every activation value feeds ONE
// MFMA column block: the exact-fp32 kernels are bound by the fp32 matrix pipe or by their operand feed (54-105 TFLOP/s).
//
// Layout of conv_f16_patch.hip with fp32 tensors in HBM: persistent blocks keep the WHOLE filter (two fp16 images, k in the natural
// (tap, channel) order) in LDS; per 16 x 16 pixel tile the raw fp32 patch (+ halo) is loaded into registers under the MFMAs of the
// previous tile, split, and written as two fp16 patch images; the MFMA operand of a lane -- 8 consecutive channels of one pixel at
// one tap -- is read straight from the patch at the tap's offset (one v_mfma_f32_16x16x32_f16 covers 32 / CT taps).  Epilogue:
// y = act1(acc 2^-S + bias) + residual, fp32, only the first cout_valid channels stored.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int XN_HEADER_FLOATS = 16;                            // [0] max |w| (bits), [1] 2^-S

__device__ __attribute__((aligned(16))) float g_xn_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

struct X3nArgs {
  const float* x1;
  const float* x2;
  const _Float16* wp;      // packed [2 images: wh, wl][nsteps][COUT][32], natural k order
  const float* hdr;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* res;        // residual [N,H,W,res_cs] (or null)
  float* y;                // [N,H,W,cout_valid]
  float* yp;               // [N,H/2,W/2,COUT] = 2x2 pooled y (or null): needs even H, W and cout_valid == COUT
  int pool_avg;            // AveragePooling2D(2) instead of MaxPool2D(2)
  int N, H, W, C1, tiles_x, tiles_y, ntiles, act1, act2, cout_valid, res_cs;
  const unsigned* xr1;     // range slots of the sources (bits of an upper bound of max |x|, conv_x3.hip "Range"; null: no scaling)
  const unsigned* xr2;
  unsigned* yr;            // range slot of the output: atomicMax of max |y| (null: not wanted)
};

// conv_x3.hip: x3_range_scale / x3_split4 / x3_range_out (the same arithmetic; kept per translation unit)
__device__ __forceinline__ void xn_range_scale(const unsigned* r1, const unsigned* r2, float& xs, float& ixs) {
  xs = 1.0f;
  ixs = 1.0f;
  unsigned b = r1 ? *r1 : 0u;
  if (r2) {
    const unsigned b2 = *r2;
    b = b2 > b ? b2 : b;
  }
  if (b != 0u && b < 0x7f800000u) {
    int ex;
    frexpf(__uint_as_float(b), &ex);
    int T = 11 - ex;
    T = T < -126 ? -126 : (T > 126 ? 126 : T);
    xs = ldexpf(1.0f, T);
    ixs = ldexpf(1.0f, -T);
  }
}
__device__ __forceinline__ void xn_split4(const f32x4 v, float xs, unsigned (&h)[2], unsigned (&l)[2]) {
  const float k2048 = 2048.0f;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float t0, t1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[p]) : "v"(v[2 * p]), "s"(xs));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[p]) : "v"(v[2 * p + 1]), "s"(xs));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(t0) : "v"(v[2 * p]), "s"(xs), "v"(h[p]));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(t1) : "v"(v[2 * p + 1]), "s"(xs), "v"(h[p]));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(l[p]) : "v"(t0), "s"(k2048));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(l[p]) : "v"(t1), "s"(k2048));
  }
}
// max |y| of a persistent block -> the range slot: one atomicMax per BLOCK (conv_x3.hip x3_range_out: the blocks all finish together,
// same-address atomics execute one after the other at the memory side); the waves' maxima meet in four LDS words behind the images
__device__ __forceinline__ void xn_range_out(unsigned* slot, float m, int lane, int wave, unsigned* lds) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) lds[wave] = __float_as_uint(m);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (threadIdx.x == 0) {
    const unsigned b01 = lds[0] > lds[1] ? lds[0] : lds[1], b23 = lds[2] > lds[3] ? lds[2] : lds[3];
    const unsigned b = b01 > b23 ? b01 : b23;
#ifndef SHDR_ABL_NO_RANGE_ATOMIC
    if (b > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, b);
#endif
  }
}

struct __attribute__((packed, aligned(4))) f32x3u { float x, y, z; };      // 12 bytes at 4-byte alignment: global_load / store_dwordx3

__host__ __device__ inline int pswz(int row) { return (-(row >> 2)) & 3; }

template <int KK, int CT, int NT, bool TWO>
struct NG {
  static constexpr int TPP = 32 / CT;                          // taps per MFMA k-step
  static constexpr int NTAPS = KK * KK;
  static constexpr int NS = (NTAPS + TPP - 1) / TPP;           // k-steps
  static constexpr int PW = 16 + KK - 1, PH = 16 + KK - 1;
  static constexpr int PP = PW * PH;                           // patch pixels
  static constexpr int PIECES = PP * CT / 4;                   // float4 pieces of the patch (both sources)
  static constexpr int PJ = (PIECES + 255) / 256;              // per thread
  static constexpr int PATCH_HALVES = PP * CT;                 // one fp16 image
  static constexpr int COUT = NT * 16;
  // HALF: with 16 channels per tap and an odd tap count the last k-step carries ONE real tap; its rows are stored as 16 halves (the zero
  // tap's half is neither stored nor read: its lanes are fed a zero patch operand) -- 512 bytes per image less, which is what puts two
  // blocks of the 7x7 16 -> 16 kernel on a CU (82 192 -> 81 168 bytes; the limit is 81 920)
#ifdef SHDR_ABL_X3N_NO_HALF
  static constexpr bool HALF = false;
#else
  static constexpr bool HALF = !TWO && CT == 16 && (NTAPS % 2 == 1) && KK >= 5;      // (3x3: five k-steps, the special last one costs 8 %)
#endif
  static constexpr int FULL_STEPS = HALF ? NS - 1 : NS;
  static constexpr int FILT_HALVES = FULL_STEPS * COUT * 32 + (HALF ? COUT * 16 : 0);      // one image
  static constexpr int FINSTR = FULL_STEPS * COUT / 16;        // filter DMA instructions per image for the full steps (16 rows of 64 bytes each)
  static constexpr int IMAGE_BYTES = (2 * FILT_HALVES + 2 * PATCH_HALVES) * 2;
  static constexpr int LDS_BYTES = IMAGE_BYTES + 16;           // + the waves' output maxima (xn_range_out)
};

// TANH: the epilogue knows SHDR_ACT_TANH (instantiated for the 16-cout single-source layers only: shdr_conv2d_x3n_ok_f32)
// Waves per SIMD the register allocator is held to (rocm 7.2 spends registers freely when nothing bounds it: the 4 -> 64 image layer
// <3,8,4> sat at 168 + 96 accumulation registers = one wave per SIMD for a kernel that fits 215, 0.386 -> 0.277 ms; the 3x3 tanh heads took
// 9 registers more than their tanh-free twins and lost their third wave, 0.252 -> 0.203 ms): 3 for the 3x3 tanh heads, 2 wherever the
// instantiation fits 256 registers without spilling (all but the 5x5 32 -> 32 and the unused 7x7 32-channel forms, now that the tap
// offsets are no longer hoisted out of the tile loop, see the k-step loop)
template <int KK, int CT, int NT>
constexpr int x3n_min_waves(bool tanh_head) {
#ifdef SHDR_ABL_X3N_NO_BOUNDS
  return 1;
#endif
#ifdef SHDR_ABL_X3N_ALL_TWO
  return (tanh_head && KK == 3 && CT <= 16) ? 3 : 2;
#endif
  return (tanh_head && KK == 3 && CT <= 16) ? 3 : (KK == 5 && CT == 32 && NT == 2) ? 1 : (KK == 7 && CT == 32) ? 1 : 2;
}
template <int KK, int CT, int NT, bool TWO, bool TANH>
__global__ __launch_bounds__(256, (x3n_min_waves<KK, CT, NT>(TANH))) void conv_x3n_kernel(const X3nArgs a) {
  using G = NG<KK, CT, NT, TWO>;
  constexpr int MT = 4;                                        // wave w owns tile rows 4w .. 4w+3
  constexpr int PAD = (KK - 1) / 2;
  extern __shared__ __attribute__((aligned(16))) _Float16 nsm[];
#define filt_h (nsm)                                           /* [NS][COUT][32], swizzled rows */
#define filt_l (nsm + G::FILT_HALVES)
#define patch_h (nsm + 2 * G::FILT_HALVES)                     /* [PP][CT] (TWO: [source][PP][16]) */
#define patch_l (nsm + 2 * G::FILT_HALVES + G::PATCH_HALVES)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- both filter images -> LDS, once per block (rows of 64 bytes, physical slot = k-group ^ swz(cout)) ---------------------------
  for (int j = wave; j < 2 * G::FINSTR; j += 4) {
    const int img = j >= G::FINSTR, jj = j - img * G::FINSTR;
    const int r = jj * 16 + (lane >> 2);                       // row = step * COUT + cout of image img
    const int co = r % G::COUT;
    const _Float16* p = a.wp + img * G::FILT_HALVES + (size_t)r * 32 + 8 * ((lane & 3) ^ pswz(co));
    __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(nsm + img * G::FILT_HALVES + jj * 512), 16, 0, 0);
  }
  if (G::HALF && wave < 2) {                                   // the 32-byte rows of the last k-step: two lanes per row, one (partial) instruction per image
    static_assert(!G::HALF || G::COUT <= 32, "one DMA instruction covers 32 half rows");
    const int co = lane >> 1;
    if (co < G::COUT) {
      const int base = wave * G::FILT_HALVES + G::FULL_STEPS * G::COUT * 32;
      const _Float16* p = a.wp + base + co * 16 + 8 * ((lane & 1) ^ (pswz(co) & 1));
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(nsm + base), 16, 0, 0);
    }
  }

  // ---- patch geometry: piece = tid + 256 j -> (source, patch pixel, float4 of the pixel), fixed over the tiles ---------------------
  int ppy[G::PJ], ppx[G::PJ], pdst[G::PJ], pch[G::PJ];        // pch: channel offset in the source tensor, -1: zero padding channels / no piece
#pragma unroll
  for (int j = 0; j < G::PJ; ++j) {
    const int piece = tid + 256 * j;
    int pix, c4, src = 0;
    if (TWO) {                                                 // image [source][pixel][16 channels]
      src = piece / (G::PP * 4);
      const int rem = piece - src * G::PP * 4;
      pix = rem >> 2;
      c4 = rem & 3;
    } else {
      pix = piece / (CT / 4);
      c4 = piece - pix * (CT / 4);
    }
    ppy[j] = pix / G::PW;
    ppx[j] = pix - ppy[j] * G::PW;
    const bool ok = piece < G::PIECES;
    if (TWO) {
      pdst[j] = ok ? src * G::PP * 16 + pix * 16 + 4 * c4 : -1;
      pch[j] = ok ? (src << 16) | (4 * c4) : -1;
    } else {
      const int cg = (CT == 32) ? ((c4 >> 1) ^ pswz(pix)) : (c4 >> 1);       // 64-byte pixels: the b128 bank swizzle
      pdst[j] = ok ? pix * CT + 8 * cg + 4 * (c4 & 1) : -1;
      pch[j] = (ok && 4 * c4 < a.C1) ? 4 * c4 : -1;            // channels beyond the source's (3 -> 4, 9 -> 12 padded inputs) are zero
    }
  }
  float xs = 1.0f, ixs = 1.0f;                                 // input scale 2^T and its inverse (conv_x3.hip "Range"): set behind the first patch loads
  f32x4 pr[G::PJ];
  // the 7x7 image layers (4 -> 16: four stores per tile and a long tap loop) measured 0.157 ms with compiler-scheduled loads against
  // 0.205 with the asm loads + counted wait that help every other shape: they keep the plain loads
  constexpr bool ASM_LOADS = !(CT == 8 && NT == 1);
  auto load_patch = [&](int tile) __attribute__((always_inline)) {
    int pm = tile;
    const int tx = pm % a.tiles_x;
    pm /= a.tiles_x;
    const int ty = pm % a.tiles_y;
    const int img = pm / a.tiles_y;
    const int ih0 = ty * 16 - PAD, iw0 = tx * 16 - PAD;
#pragma unroll
    for (int j = 0; j < G::PJ; ++j) {
      const int ih = ih0 + ppy[j], iw = iw0 + ppx[j];
      const bool ok = pch[j] >= 0 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
      const float* src = (TWO && (pch[j] >> 16)) ? a.x2 : a.x1;
      const int cs = TWO ? 16 : a.C1;
      if (!ASM_LOADS) {                                          // compiler-scheduled loads (and its own wait in front of store_patch)
        pr[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ok) pr[j] = *reinterpret_cast<const f32x4*>(src + ((size_t)(img * a.H + ih) * a.W + iw) * cs + (pch[j] & 0xFFFF));
        continue;
      }
      const float* p = ok ? src + ((size_t)(img * a.H + ih) * a.W + iw) * cs + (pch[j] & 0xFFFF) : g_xn_zero_page;
      // inline asm, one load per piece from every lane (padding and out-of-image pieces read the zero page): outside the compiler's
      // scoreboard -- patch_wait() below is the loads' only wait.  (Exec-masked loads that leave zeros in the padding lanes measured
      // slower on the 4 -> 64 layer, 0.475 vs 0.427 ms, and equal elsewhere.)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pr[j]) : "v"(p) : "memory");
    }
  };
  // the next tile's patch loads are OLDER than this tile's output stores (vector-memory operations retire in issue order on gfx9:
  // one counter for loads and stores), so "vmcnt(number of stores issued since)" = the patch has landed while the stores stay in
  // flight; the compiler's own wait for a register loaded in the previous loop iteration is vmcnt(0), i.e. the stores' round trip
  auto patch_wait = [&](int stores_since) __attribute__((always_inline)) {
    if (!ASM_LOADS) return;
    constexpr int FULL = MT * NT, FULL_POOL = MT * NT * 3 / 2;       // gfx9 encoding: vmcnt = bits 3:0 and 15:14
    if (stores_since == FULL) __builtin_amdgcn_s_waitcnt(0x0F70 | (FULL & 15) | ((FULL >> 4) << 14));
    else if (stores_since == FULL_POOL) __builtin_amdgcn_s_waitcnt(0x0F70 | (FULL_POOL & 15) | ((FULL_POOL >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int j = 0; j < G::PJ; ++j) asm volatile("" : "+v"(pr[j]));
  };
  auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < G::PJ; ++j) {
      if (pdst[j] < 0) continue;
      if (ASM_LOADS) {
        unsigned h[2], l[2];
        xn_split4(pr[j], xs, h, l);
        *reinterpret_cast<uint2*>(patch_h + pdst[j]) = make_uint2(h[0], h[1]);
        *reinterpret_cast<uint2*>(patch_l + pdst[j]) = make_uint2(l[0], l[1]);
      } else {                                                 // the 7x7 image layers: compiler-scheduled split as well (0.16 vs 0.20 ms with the asm form)
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = pr[j][e] * xs;
          h[e] = (_Float16)v;
          l[e] = (_Float16)((v - (float)h[e]) * 2048.0f);
        }
        *reinterpret_cast<f16x4*>(patch_h + pdst[j]) = h;
        *reinterpret_cast<f16x4*>(patch_l + pdst[j]) = l;
      }
    }
  };

  // ---- operand geometry -----------------------------------------------------------------------------------------------------
  const int fi = lane & 15, fg = lane >> 4;
  constexpr int GPT = TWO ? 2 : 4 / G::TPP;                    // k-groups (8 channels) per tap (TWO: per source)
  const int jl = fg / GPT;                                     // which tap of the k-step this lane's k-group belongs to (TWO: which source)
  const int cgl = fg % GPT;                                    // which 8-channel group of the pixel
  int a_base[MT];                                              // half offset of (tile row, pixel fi, tap 0) in a patch image
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

  // Workgroup barriers of the tile loop order LDS traffic only: __syncthreads() also fences global memory (s_waitcnt vmcnt(0)), i.e.
  // every wave would sit out the write latency of the previous tile's output stores at the top of each tile (PMC on the 3x3 4 -> 64
  // image layer: 63 % of the wave cycles in s_waitcnt with two blocks per CU).  The stores stay in flight across the raw barrier;
  // the patch registers are waited for by the compiler's own counted vmcnt (loads and stores retire in issue order).
  auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // The bias is the same for every tile of the persistent block: loaded once.  Inside the epilogue each of its loads was followed by
  // "s_waitcnt vmcnt(0)", which also drained the output store issued just before it -- sixteen serialised store round trips per tile.
  f32x4 bias_r[NT];
#pragma unroll
  for (int ni = 0; ni < NT; ++ni)
    bias_r[ni] = (a.bias && a.cout_valid == G::COUT) ? *reinterpret_cast<const f32x4*>(a.bias + ni * 16 + 4 * fg) : (f32x4){0.f, 0.f, 0.f, 0.f};
  float ym = 0.0f;                                             // max |y| over the values this lane stores (a.yr)
  int tile = blockIdx.x;
  if (tile < a.ntiles) load_patch(tile);
#ifndef SHDR_ABL_NO_SCALE
  xn_range_scale(a.xr1, a.xr2, xs, ixs);                     // behind the first patch loads: the slot's round trip hides under theirs
#endif
  const float inv_s = a.hdr[1] * ixs;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of the filter (LDS-DMA) have landed
  int stores_since = -1;                                        // output store instructions this wave issued after its last patch loads (-1: unknown)
  for (; tile < a.ntiles; tile += gridDim.x) {
    lds_barrier();                                             // every wave is done with the previous tile's patch (first tile: the filter is complete)
    patch_wait(stores_since);
    store_patch();
    lds_barrier();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) load_patch(next);                     // lands under this tile's MFMAs
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
      // opaque to the optimiser: the tap offsets are tile-invariant, and hoisted out of the persistent tile loop the MT x NS operand
      // addresses of the unrolled k-steps were 100 - 250 live registers (7x7 16 -> 16: 256 + 116 registers = one wave per SIMD; 195 + 20 now)
      asm volatile("" : "+v"(toff));
      f16x8 wh[NT], wl[NT], ws[NT], ph[MT], pl[MT];
      const bool last_half = G::HALF && s == G::NS - 1;       // (compile-time: the loop is unrolled)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        int bo = s * G::COUT * 32 + b_rd[ni];
        if (last_half) {                                         // 16-half rows; the lanes of the absent tap alias the real one's (finite) weights
          const int row = ni * 16 + fi;
          bo = s * G::COUT * 32 + row * 16 + 8 * ((fg & 1) ^ (pswz(row) & 1));
        }
        wh[ni] = *reinterpret_cast<const f16x8*>(filt_h + bo);
        wl[ni] = *reinterpret_cast<const f16x8*>(filt_l + bo);
        ws[ni] = wh[ni] * (_Float16)(1.0f / 2048.0f);
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        int ad = a_base[mi] + toff;
        if (!TWO && CT == 32) {                                // undo the swizzle of the 64-byte pixel
          const int pix = ad >> 5;
          ad = (pix << 5) + 8 * (cgl ^ pswz(pix));
        }
        ph[mi] = *reinterpret_cast<const f16x8*>(patch_h + ad);
        pl[mi] = *reinterpret_cast<const f16x8*>(patch_l + ad);
        if (last_half && jl > 0) {                               // the absent tap of the last k-step: a zero operand
          ph[mi] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
          pl[mi] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], ph[mi], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ws[ni], pl[mi], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ni], ph[mi], acc[mi][ni], 0, 0, 0);
        }
    }

    // ---- epilogue: lane (fi, fg) holds couts 4fg..4fg+3 of pixel (row wave*4 + mi, column fi) per 16-cout tile ----------------
    int pm = tile;
    const int tx = pm % a.tiles_x;
    pm /= a.tiles_x;
    const int ty = pm % a.tiles_y;
    const int img = pm / a.tiles_y;
    const int oh0 = ty * 16, ow0 = tx * 16;
    // store instructions this wave issues below: one per (row, cout tile) and one per (row pair, cout tile) of the pooled output when
    // all its rows are inside the image and every filter column is stored -- anything else makes patch_wait() drain
    stores_since = (a.cout_valid == G::COUT && oh0 + wave * MT + MT <= a.H && !a.res && !a.scale) ? (a.yp ? MT * NT * 3 / 2 : MT * NT) : -1;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int oh = oh0 + wave * MT + mi, ow = ow0 + fi;
      if (oh >= a.H || ow >= a.W) continue;
      const size_t pix = ((size_t)img * a.H + oh) * a.W + ow;
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        const int co = ni * 16 + 4 * fg;
        f32x4 v = acc[mi][ni] * inv_s;
        // y = act2(affine(act1(acc + bias)) + residual)
        if (a.cout_valid == G::COUT) {
          v += bias_r[ni];
          shdr::act_apply4<(TANH ? 2 : 0)>(v, a.act1);
          if (a.scale) v = v * *reinterpret_cast<const f32x4*>(a.scale + co) + *reinterpret_cast<const f32x4*>(a.shift + co);
          if (a.res) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += a.res[pix * a.res_cs + co + e];
          }
          shdr::act_apply4<(TANH ? 2 : 0)>(v, a.act2);
          // 16 lanes x 16 bytes at a stride of COUT * 4 bytes: the four lane groups of a pixel complete its 64- / 128-byte row
          *reinterpret_cast<f32x4*>(a.y + pix * G::COUT + co) = v;
#ifndef SHDR_ABL_NO_YM
          if (a.yr) ym = fmaxf(fmaxf(fmaxf(fmaxf(ym, fabsf(v[0])), fabsf(v[1])), fabsf(v[2])), fabsf(v[3]));
#endif
          acc[mi][ni] = v;                                            // kept for the pooled output below
        } else {                                                      // a zero-padded filter (e.g. 3 of 16 couts stored): guarded scalar accesses
          if (co >= a.cout_valid) continue;
          f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (a.bias && co + e < a.cout_valid) t[e] = a.bias[co + e];
          v += t;
          shdr::act_apply4<(TANH ? 2 : 0)>(v, a.act1);
          const bool three = a.cout_valid == 3;                     // the 3-channel heads: ONE 12-byte access per pixel instead of three scalar ones
          if (three && a.res) {                                     // (16 lanes x 12 bytes are contiguous: a pixel row of the image)
            const f32x3u r3 = *reinterpret_cast<const f32x3u*>(a.res + pix * a.res_cs);
            t = (f32x4){r3.x, r3.y, r3.z, 0.f};
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < a.cout_valid) {
              if (a.scale) v[e] = v[e] * a.scale[co + e] + a.shift[co + e];
              if (a.res) v[e] += three ? t[e] : a.res[pix * a.res_cs + co + e];
            }
          shdr::act_apply4<(TANH ? 2 : 0)>(v, a.act2);
          if (three) {
            f32x3u o3;
            o3.x = v[0]; o3.y = v[1]; o3.z = v[2];
            *reinterpret_cast<f32x3u*>(a.y + pix * 3) = o3;
            if (a.yr) ym = fmaxf(fmaxf(fmaxf(ym, fabsf(v[0])), fabsf(v[1])), fabsf(v[2]));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (co + e < a.cout_valid) {
                a.y[pix * a.cout_valid + co + e] = v[e];
                if (a.yr) ym = fmaxf(ym, fabsf(v[e]));
              }
          }
        }
      }
    }
    // ---- optional second output: the 2 x 2 window is rows (2 mp, 2 mp + 1) of this lane and of lane fi ^ 1 (H, W even: a window is
    //      inside the image or outside as a whole); sums in pool.hip's order, (top-left + top-right) + (bottom-left + bottom-right)
    if (a.yp) {
#pragma unroll
      for (int mp = 0; mp < MT / 2; ++mp) {
        const int oh = oh0 + wave * MT + 2 * mp, ow = ow0 + fi;
        if (oh >= a.H) continue;                                      // wave-uniform
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          f32x4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t0 = acc[2 * mp][ni][e], b0 = acc[2 * mp + 1][ni][e];
            if (a.pool_avg) {
              const float t = t0 + __shfl_xor(t0, 1, 64), b = b0 + __shfl_xor(b0, 1, 64);
              m[e] = 0.25f * (t + b);
            } else {
              const float mx = fmaxf(t0, b0);
              m[e] = fmaxf(mx, __shfl_xor(mx, 1, 64));
            }
          }
          if (!(fi & 1) && ow < a.W)
            *reinterpret_cast<f32x4*>(a.yp + (((size_t)img * (a.H >> 1) + (oh >> 1)) * (a.W >> 1) + (ow >> 1)) * G::COUT + ni * 16 + 4 * fg) = m;
        }
      }
    }
  }
#ifndef SHDR_ABL_NO_TAIL
  if (a.yr) xn_range_out(a.yr, ym, lane, wave, reinterpret_cast<unsigned*>(reinterpret_cast<char*>(nsm) + G::IMAGE_BYTES));                        // once per persistent block and wave (a pooled output has the same bound)
#endif
#undef filt_h
#undef filt_l
#undef patch_h
#undef patch_l
}

// max |w| -> hdr[0] (bits)
__global__ __launch_bounds__(256) void x3n_absmax_kernel(const float* __restrict__ w, long n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
// packed[image][step][co][k]: k = (tap, channel) in natural order, CT channels per tap (the source's C1 [+ C2] real ones, zero beyond),
// zero beyond the last tap and beyond the real couts; w * x2-scale * 2^S split into wh, wl
__global__ __launch_bounds__(256) void x3n_pack_kernel(const float* __restrict__ w, float* __restrict__ hdr, _Float16* __restrict__ out, int ntaps,
                                                       int CT, int C1, int Creal, int cout_real, int Cout_w, int COUT, int NS, float x2_scale,
                                                       int half_last) {
  const float mx = fmaxf(__uint_as_float(reinterpret_cast<const unsigned*>(hdr)[0]) * fmaxf(1.0f, fabsf(x2_scale)), 1e-30f);
  int ex;
  frexpf(mx, &ex);
  int S = 14 - ex;
  S = S < -100 ? -100 : (S > 100 ? 100 : S);
  const float s = ldexpf(1.0f, S);
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[1] = ldexpf(1.0f, -S);
  // half_last (NG::HALF): the last k-step holds one 16-channel tap: rows of 16 halves instead of 32
  const long full = (long)(NS - (half_last ? 1 : 0)) * COUT * 32;
  const long total = full + (half_last ? COUT * 16 : 0);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    int kk, co, st;
    if (e < full) {
      kk = (int)(e & 31); co = (int)((e >> 5) % COUT); st = (int)(e / (32L * COUT));
    } else {
      const int e2 = (int)(e - full);
      kk = e2 & 15; co = e2 >> 4; st = NS - 1;
    }
    const int k = st * 32 + kk;
    const int tap = k / CT, ch = k - tap * CT;
    float v = 0.0f;
    if (tap < ntaps && ch < Creal && co < cout_real) {
      v = w[((size_t)tap * Creal + ch) * Cout_w + co] * s;
      if (ch >= C1) v *= x2_scale;
    }
    const _Float16 h = (_Float16)v;
    out[e] = h;
    out[total + e] = (_Float16)(v - (float)h);
  }
}

template <int KK, int CT, int NT, bool TWO, bool TANH = false>
int launch_x3n(X3nArgs& a, hipStream_t st) {
  using G = NG<KK, CT, NT, TWO>;
  constexpr int lds = G::LDS_BYTES;
  if constexpr (lds > 160 * 1024) {
    return shdr::fail(SHDR_E_SHAPE, "conv2d_x3n: filter + patch (%d bytes) do not fit the LDS", lds);
  } else {
    const int dev_slot = shdr::device_slot();
    static bool attr_done[shdr::kMaxDevices] = {};
    static int occ[shdr::kMaxDevices] = {};
    if (!attr_done[dev_slot]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3n_kernel<KK, CT, NT, TWO, TANH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
      int nb = 0;
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_x3n_kernel<KK, CT, NT, TWO, TANH>), 256, lds);
      occ[dev_slot] = (e != hipSuccess || nb < 1) ? 1 : (nb > 4 ? 4 : nb);
      attr_done[dev_slot] = true;
    }
    long grid = 256L * occ[dev_slot];
    if (grid > a.ntiles) grid = a.ntiles;
    hipLaunchKernelGGL((conv_x3n_kernel<KK, CT, NT, TWO, TANH>), dim3((unsigned)grid), dim3(256), lds, st, a);
    return shdr::check_launch("conv_x3n_kernel");
  }
}

inline int ct_of(const shdr_conv2d_desc* d) { return d->C2 > 0 ? 32 : (d->C1 <= 8 ? 8 : (d->C1 <= 16 ? 16 : 32)); }

template <int KK, int NT>
int dispatch_ct(X3nArgs& a, const shdr_conv2d_desc* d, hipStream_t st) {
  if (d->C2 > 0) return launch_x3n<KK, 32, NT, true>(a, st);
  const int ct = ct_of(d);
  if constexpr (NT == 1) {                                   // the tanh heads of the U-Nets (16 couts, 3 stored): the only TANH instantiations
    if (d->act1 == SHDR_ACT_TANH || d->act2 == SHDR_ACT_TANH) {
      if (ct == 8) return launch_x3n<KK, 8, 1, false, true>(a, st);
      if (ct == 16) return launch_x3n<KK, 16, 1, false, true>(a, st);
      return launch_x3n<KK, 32, 1, false, true>(a, st);
    }
  }
  if (ct == 8) return launch_x3n<KK, 8, NT, false>(a, st);
  if (ct == 16) return launch_x3n<KK, 16, NT, false>(a, st);
  return launch_x3n<KK, 32, NT, false>(a, st);
}

}  // namespace

// 1 if the narrow split-operand kernel takes the layer
extern "C" int shdr_conv2d_x3n_ok_f32(const shdr_conv2d_desc* d) {
  if (!d || d->stride != 1 || d->KH != d->KW || !(d->KH == 3 || d->KH == 5 || d->KH == 7)) return 0;
  if (d->pad_t != (d->KH - 1) / 2 || d->pad_l != (d->KW - 1) / 2 || d->Ho != d->H || d->Wo != d->W) return 0;
  // (desc.prologue plays no part: the plan of a layer must not depend on it -- shdr_conv2d_fwd_prepared_f32 materialises a prologue this
  //  kernel does not fuse and re-enters with the same prepared filter)
  // Cout 64: the 3x3 image layers (3 -> 4 channels in, hal conv1_1 / VGG conv1_1: K = 36, an HBM-write-bound layer)
  const bool image64 = d->Cout == 64 && d->KH == 3 && d->C2 == 0 && d->C1 <= 8 && SHDR_ENV("SHDR_NO_X3N_IMAGE64") == nullptr;
  if (!(d->Cout == 16 || d->Cout == 32 || image64) || d->w_batch_stride != 0 || d->y_pix_stride > 1) return 0;
  const int cv = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  if (cv > d->Cout || (d->y_cstride != 0 && d->y_cstride != cv)) return 0;
  // tanh is compiled into the 16-cout single-source kernels only (act_apply4: its code would evict the tile loop from the instruction cache)
  if ((d->act1 == SHDR_ACT_TANH || d->act2 == SHDR_ACT_TANH) && !(d->Cout == 16 && d->C2 == 0)) return 0;
  const bool one = d->C2 == 0 && d->C1 % 4 == 0 && d->C1 >= 4 && d->C1 <= 32, two = d->C1 == 16 && d->C2 == 16;
  if (!(one || two) || SHDR_ENV("SHDR_NO_X3") || SHDR_ENV("SHDR_NO_X3N")) return 0;
  if ((long)d->N * d->H * d->W * 32 >= (1L << 31)) return 0;
  const int ct = ct_of(d);
  const long filt = 2L * ((d->KH * d->KW * ct + 31) / 32) * d->Cout * 64;
  const long patch = 2L * (16 + d->KH - 1) * (16 + d->KW - 1) * ct * 2;
  if (filt + patch > 150 * 1024) return 0;
  long min_tiles = 256;                                     // persistent blocks: at least one tile per CU
  if (const char* e = SHDR_ENV("SHDR_X3_MIN_BLOCKS")) min_tiles = atol(e);
  return (long)d->N * ((d->H + 15) / 16) * ((d->W + 15) / 16) >= min_tiles ? 1 : 0;
}

extern "C" int64_t shdr_conv2d_x3n_filter_elems_f32(const shdr_conv2d_desc* d) {
  if (!d || d->KH <= 0 || !(d->Cout == 16 || d->Cout == 32 || d->Cout == 64)) return -1;
  const int ct = ct_of(d);
  const int64_t ns = (d->KH * d->KW * ct + 31) / 32;
  return XN_HEADER_FLOATS + ns * d->Cout * 32;               // header + two fp16 images, in floats
}

// w: HWIO [KH][KW][C1 + C2][cout_w] with cout_w = the filter tensor's channel count (>= cout_valid; the desc's Cout may be its padding)
extern "C" int shdr_conv2d_x3n_prepare_filter_premax_f32(const shdr_conv2d_desc* d, const float* w, float* prepared, int premax, void* stream);
extern "C" int shdr_conv2d_x3n_prepare_filter_f32(const shdr_conv2d_desc* d, const float* w, float* prepared, void* stream) {
  return shdr_conv2d_x3n_prepare_filter_premax_f32(d, w, prepared, 0, stream);
}
// premax: header slot 0 already holds max |w| (conv_x3.hip: shdr_conv2d_x3_prepare_filter_premax_f32)
extern "C" int shdr_conv2d_x3n_prepare_filter_premax_f32(const shdr_conv2d_desc* d, const float* w, float* prepared, int premax, void* stream) {
  SHDR_REQUIRE(d && w && prepared, SHDR_E_NULL, "conv2d_x3n_prepare_filter: null pointer");
  SHDR_REQUIRE(shdr::aligned16(prepared), SHDR_E_ALIGN, "conv2d_x3n_prepare_filter: prepared must be 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int ct = ct_of(d), Creal = d->C1 + d->C2, ntaps = d->KH * d->KW, ns = (ntaps * ct + 31) / 32;
  const int cv = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  const long nw = (long)ntaps * Creal * d->Cout;             // the filter tensor handed over has the desc's (padded) Cout columns
  if (!premax) {
    if (hipMemsetAsync(prepared, 0, XN_HEADER_FLOATS * sizeof(float), st) != hipSuccess) return shdr::fail(SHDR_E_LAUNCH, "conv2d_x3n_prepare_filter: memset");
    hipLaunchKernelGGL(x3n_absmax_kernel, dim3(shdr::stream_grid(nw) < 64 ? shdr::stream_grid(nw) : 64), dim3(256), 0, st, w, nw,
                       reinterpret_cast<unsigned*>(prepared));
  }
  hipLaunchKernelGGL(x3n_pack_kernel, dim3(shdr::stream_grid((long)ns * d->Cout * 32)), dim3(256), 0, st, w, prepared,
                     reinterpret_cast<_Float16*>(prepared + XN_HEADER_FLOATS), ntaps, ct, d->C1, Creal, cv, d->Cout, d->Cout, ns,
#ifdef SHDR_ABL_X3N_NO_HALF
                     d->C2 > 0 ? d->x2_scale : 1.0f, 0);
#else
                     d->C2 > 0 ? d->x2_scale : 1.0f, (d->C2 == 0 && ct == 16 && ntaps % 2 == 1 && ntaps >= 25) ? 1 : 0);       // NG::HALF
#endif
  return shdr::check_launch("conv2d_x3n_prepare_filter");
}

extern "C" int shdr_conv2d_fwd_x3n_ranged_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                                              const float* scale, const float* shift, const float* residual, float* y, float* y_pool,
                                              const float* x1_range, const float* x2_range, float* y_range, void* stream) {
  SHDR_REQUIRE(d && x1 && prepared && y, SHDR_E_NULL, "conv2d_x3n: null desc/x1/filter/y");
  SHDR_REQUIRE(!y_pool || (d->H % 2 == 0 && d->W % 2 == 0 && shdr::aligned16(y_pool) && (d->cout_valid == 0 || d->cout_valid == d->Cout)),
               SHDR_E_SHAPE, "conv2d_x3n: the pooled output needs even H, W and every filter column stored");
  SHDR_REQUIRE(shdr_conv2d_x3n_ok_f32(d), SHDR_E_SHAPE, "conv2d_x3n: layer shape not taken by this kernel");
  SHDR_REQUIRE((d->C2 == 0) == (x2 == nullptr), SHDR_E_NULL, "conv2d_x3n: x2 must be given iff C2 > 0");
  SHDR_REQUIRE((scale == nullptr) == (shift == nullptr), SHDR_E_NULL, "conv2d_x3n: scale and shift come together");
  SHDR_REQUIRE(!residual || d->res_cstride >= (d->cout_valid > 0 ? d->cout_valid : d->Cout), SHDR_E_SHAPE, "conv2d_x3n: res_cstride");
  SHDR_REQUIRE(shdr::aligned16(x1) && (!x2 || shdr::aligned16(x2)) && shdr::aligned16(prepared) && shdr::aligned16(y) && (!bias || shdr::aligned16(bias)),
               SHDR_E_ALIGN, "conv2d_x3n: tensors must be 16-byte aligned");
  X3nArgs a{};
  a.x1 = x1; a.x2 = x2;
  a.hdr = prepared;
  a.wp = reinterpret_cast<const _Float16*>(prepared + XN_HEADER_FLOATS);
  a.bias = bias; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
  a.yp = y_pool; a.pool_avg = d->pool == SHDR_POOL_AVG;
  a.N = d->N; a.H = d->H; a.W = d->W; a.C1 = d->C1;
  a.tiles_x = (d->W + 15) / 16;
  a.tiles_y = (d->H + 15) / 16;
  a.ntiles = a.N * a.tiles_x * a.tiles_y;
  a.act1 = d->act1; a.act2 = d->act2;
  SHDR_REQUIRE(!x2 || ((x1_range == nullptr) == (x2_range == nullptr)), SHDR_E_NULL, "conv2d_x3n: give the range of both sources or of neither");
  a.xr1 = reinterpret_cast<const unsigned*>(x1_range);
  a.xr2 = reinterpret_cast<const unsigned*>(x2 ? x2_range : nullptr);
  if (!x1_range && d->prologue == SHDR_PROLOGUE_RANGE_SCALE) a.xr1 = reinterpret_cast<const unsigned*>(prepared) + 2;   // header-slot protocol
  a.yr = reinterpret_cast<unsigned*>(y_range);
  a.cout_valid = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  a.res_cs = d->res_cstride;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->Cout == 64) return launch_x3n<3, 8, 4, false>(a, st);
  if (d->Cout == 16) {
    if (d->KH == 3) return dispatch_ct<3, 1>(a, d, st);
    if (d->KH == 5) return dispatch_ct<5, 1>(a, d, st);
    return dispatch_ct<7, 1>(a, d, st);
  }
  if (d->KH == 3) return dispatch_ct<3, 2>(a, d, st);
  if (d->KH == 5) return dispatch_ct<5, 2>(a, d, st);
  return dispatch_ct<7, 2>(a, d, st);
}

// the low-level entry point without range slots (conv_x3.hip: shdr_conv2d_fwd_x3_f32)
extern "C" int shdr_conv2d_fwd_x3n_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                                       const float* scale, const float* shift, const float* residual, float* y, float* y_pool, void* stream) {
  return shdr_conv2d_fwd_x3n_ranged_f32(d, x1, x2, prepared, bias, scale, shift, residual, y, y_pool, nullptr, nullptr, nullptr, stream);
}
