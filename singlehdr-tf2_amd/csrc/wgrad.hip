// Convolution backward on gfx950: weight gradient and the filter transform used by dgrad.
//
//   dW[kh][kw][ci][co] += x_scale * sum_{n,oh,ow} X[n, oh*s+kh-pt, ow*s+kw-pl, ci] * dZ[n,oh,ow,co]
//
//  * wgrad_mfma_kernel: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) with the contraction over
//    PIXELS: D[ci][co] = sum_p Xs[p][ci] * dZ[p][co].  Both operands are pixel-major /
//    channel-contiguous in HBM, so a chunk of 32 pixels x (BMc | BNc) channels goes to LDS by
//    global_load_lds_dwordx4 (lane-linear image, column swizzle c ^ 16*(p&1) applied on the source
//    side, zero page for taps that fall into the padding).  One block = one filter tap x one
//    (ci, co) tile x one slice of pixels; partial tiles are accumulated into dW with fp32 atomics
//    (each wave instruction adds 4 rows x 64 contiguous bytes).
//  * wgrad_direct_kernel: VALU fallback for channel counts the MFMA tile cannot take.
//  * dgrad itself is the forward kernel run on dZ with the transformed filter
//    Wt[kh][kw][co][ci] = W[KH-1-kh][KW-1-kw][ci][co]  (filter_transform_kernel).
//
// Replaces GradientTape.gradient through tf.keras.layers.Conv2D (joint_training.py:185,
// train.py:175,195,242, finetune_real_dataset.py:177).
#include <stdlib.h>
#include <type_traits>

#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) float g_wg_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

struct WgradArgs {
  const float* x;    // [N,H,W,Cx]
  const float* dz;   // [N,Ho,Wo,Cout]
  float* dw;         // [KH*KW][Ct][Cout]
  int N, H, W, Cx, Ct, ci_off, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo;
  int npix;          // N*Ho*Wo
  int slice;         // pixels per block (multiple of 32)
  int nslices;
  int tiles_m, tiles_n;
  float x_scale;
  int prec;          // 0: exact fp32 MFMA, 1: fp16 operands, 2: bf16 operands
};

constexpr int PK = 32;  // pixels per chunk

// PREC = 1 / 2: fp16 / bf16 MFMA operands (BASELINE configs[4]).  Same fp32 LDS image and the same fragment reads; the
// 8 pixels {4s + g, s = 0..7} a lane group owns in a chunk are rounded and packed into one operand of
// v_mfma_f32_16x16x32_{f16,bf16}: 1 MFMA per 16x16 tile and chunk instead of 8; fp32 accumulation and fp32 atomics.
template <int BMc, int BNc, int WM, int WN, int PREC = 0>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(const WgradArgs a) {
  static_assert(WM * WN <= 4, "at most 4 computing waves (the others only feed the DMA)");
  constexpr int MT = BMc / WM / 16;   // 16-ci groups per computing wave
  constexpr int NT = BNc / WN / 16;
  static_assert(MT >= 1 && NT >= 1 && MT * WM * 16 == BMc && NT * WN * 16 == BNc, "wave tiles must cover the block tile");
  constexpr int XQ = BMc / 4, ZQ = BNc / 4;          // quads per pixel row
  constexpr int XI_TOTAL = PK * XQ / 64, ZI_TOTAL = PK * ZQ / 64;   // wave DMA instructions per chunk
  constexpr int XI = (XI_TOTAL + 3) / 4, ZI = (ZI_TOTAL + 3) / 4;   // per wave
  constexpr int X_WAVES = XI_TOTAL >= 4 ? 4 : XI_TOTAL, Z_WAVES = ZI_TOTAL >= 4 ? 4 : ZI_TOTAL;
  constexpr bool XSWZ = BMc >= 32, ZSWZ = BNc >= 32;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                       // [2][PK][BMc]
  float* Zs = smem + 2 * PK * BMc;        // [2][PK][BNc]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool computing = wave < WM * WN;
  const int wm = (wave / WN) % WM, wn = wave % WN;
  int t = blockIdx.x;
  const int tn = t % a.tiles_n; t /= a.tiles_n;
  const int tm = t % a.tiles_m; t /= a.tiles_m;
  const int tap = t;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int ci0 = tm * BMc, co0 = tn * BNc;
  const int p_begin = blockIdx.y * a.slice;
  const int p_end = min(p_begin + a.slice, a.npix);
  const int nchunks = (p_end - p_begin + PK - 1) / PK;
  const float* zero = g_wg_zero_page;

  // lane -> (pixel in chunk, channel quad) of each DMA instruction
  int xp[XI], xc[XI], zp[ZI], zc[ZI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int Q = (wave * XI + i) * 64 + lane;
    xp[i] = Q / XQ;
    const int pq = Q % XQ;
    xc[i] = 4 * (XSWZ ? (pq ^ (4 * (xp[i] & 1))) : pq);
  }
#pragma unroll
  for (int i = 0; i < ZI; ++i) {
    const int Q = (wave * ZI + i) * 64 + lane;
    zp[i] = Q / ZQ;
    const int pq = Q % ZQ;
    zc[i] = 4 * (ZSWZ ? (pq ^ (4 * (zp[i] & 1))) : pq);
  }

  // (n, oh, ow) of this lane's pixel for each X instruction, advanced by PK pixels per chunk
  int s_ow[XI], s_oh[XI], s_n[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int p = p_begin + xp[i];
    s_ow[i] = p % a.Wo;
    const int q = p / a.Wo;
    s_oh[i] = q % a.Ho;
    s_n[i] = q / a.Ho;
  }

  auto dma_chunk = [&](int c, int buf) {
    const int p0 = p_begin + c * PK;
    float* Xb = Xs + buf * PK * BMc + (wave * XI) * 256;
    float* Zb = Zs + buf * PK * BNc + (wave * ZI) * 256;
    if (wave < X_WAVES) {
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int p = p0 + xp[i];
        const int ih = s_oh[i] * a.stride - a.pad_t + kh, iw = s_ow[i] * a.stride - a.pad_l + kw;
        const bool ok = p < p_end && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        const float* src = ok ? a.x + ((size_t)((s_n[i] * a.H + ih) * a.W + iw) * a.Cx + ci0 + xc[i]) : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Xb + i * 256), 16, 0, 0);
        s_ow[i] += PK;
        while (s_ow[i] >= a.Wo) {
          s_ow[i] -= a.Wo;
          if (++s_oh[i] == a.Ho) { s_oh[i] = 0; ++s_n[i]; }
        }
      }
    }
    if (wave < Z_WAVES) {
#pragma unroll
      for (int i = 0; i < ZI; ++i) {
        const int p = p0 + zp[i];
        const float* src = p < p_end ? a.dz + ((size_t)p * a.Cout + co0 + zc[i]) : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Zb + i * 256), 16, 0, 0);
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fg = lane >> 4;
  int xcol[MT], zcol[NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int c = wm * MT * 16 + mi * 16 + fi;
    xcol[mi] = XSWZ ? (c ^ (16 * (fg & 1))) : c;
  }
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    const int c = wn * NT * 16 + ni * 16 + fi;
    zcol[ni] = ZSWZ ? (c ^ (16 * (fg & 1))) : c;
  }

  auto compute_chunk = [&](int buf) {
    const float* Xb = Xs + buf * PK * BMc + fg * BMc;
    const float* Zb = Zs + buf * PK * BNc + fg * BNc;
    if constexpr (PREC != 0) {
      using frag_t = std::conditional_t<PREC == 1, f16x8, bf16x8>;
      using elem_t = std::conditional_t<PREC == 1, _Float16, __bf16>;
      frag_t xf[MT], zf[NT];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) xf[mi][s] = (elem_t)Xb[4 * s * BMc + xcol[mi]];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) zf[ni][s] = (elem_t)Zb[4 * s * BNc + zcol[ni]];
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          if constexpr (PREC == 1)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[mi], zf[ni], acc[mi][ni], 0, 0, 0);
          else
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[mi], zf[ni], acc[mi][ni], 0, 0, 0);
        }
      return;
    }
#pragma unroll
    for (int s = 0; s < PK / 4; ++s) {   // pixel p = 4*s + fg of the chunk
      float xa[MT], zb[NT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) xa[mi] = Xb[4 * s * BMc + xcol[mi]];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) zb[ni] = Zb[4 * s * BNc + zcol[ni]];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[mi], zb[ni], acc[mi][ni], 0, 0, 0);
    }
  };

  if (nchunks > 0) {
    dma_chunk(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) dma_chunk(c + 1, (c + 1) & 1);
      if (computing) compute_chunk(c & 1);
      __syncthreads();
    }
  }

  // D[row = ci][col = co]: lane holds rows 4*fg + r, column fi of each 16x16 tile
  if (!computing) return;
  float* dwt = a.dw + (size_t)tap * a.Ct * a.Cout;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int co = co0 + wn * NT * 16 + ni * 16 + fi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + wm * MT * 16 + mi * 16 + 4 * fg + r;
        if (ci < a.Cx && co < a.Cout)
          atomicAdd(dwt + (size_t)(a.ci_off + ci) * a.Cout + co, acc[mi][ni][r] * a.x_scale);
      }
    }
}

// All-taps weight gradient for the narrow full-resolution layers (Cin = 16 / 32 per source, Cout = 16 / 32, stride 1, square
// 3x3 / 5x5 / 7x7 filters: the U-Nets of the Dequantization- and Refinement-Net).  wgrad_mfma_kernel<16,16> gives every filter
// tap its own block, so the 32-pixel X and dZ tiles (4 KB) are DMA-staged again for each of the up to 49 taps and feed 8
// MFMAs: 250 B of LDS-DMA per kFLOP, which is the LDS-DMA ceiling of a CU (33 TFLOP/s).  Here ONE block covers all taps of a
// 32-pixel row segment: the KK x (32 + KK - 1) input patch and the 32 gradient pixels are staged once per chunk (24 B per
// kFLOP at 7x7) and the four waves split the taps -- wave w keeps the tiles of taps w, w+4, ... (<= 13) in registers, reads
// the dZ operand once per k-step and the X operand of a tap at its shifted pixel offset (consecutive lanes = consecutive
// channels and pixels: conflict-free b32 reads).  Partial tiles go to dW by fp32 atomics once per block.
template <int KK, int MT, int NT>
__global__ __launch_bounds__(256) void wgrad_alltaps_kernel(const WgradArgs a) {
  constexpr int NTAPS = KK * KK, TPW = (NTAPS + 3) / 4;
  constexpr int CX = 16 * MT, CO = 16 * NT;
  constexpr int PWD = 32 + KK - 1;                       // patch width in pixels
  constexpr int XQ = CX / 4, ZQ = CO / 4;                // 16-byte quads per pixel
  constexpr int X_SLOTS = KK * PWD * XQ, Z_SLOTS = 32 * ZQ;
  constexpr int X_INSTR = (X_SLOTS + 63) / 64, Z_INSTR = (Z_SLOTS + 63) / 64;
  constexpr int XJ = (X_INSTR + 3) / 4, ZJ = (Z_INSTR + 3) / 4;
  constexpr int X_FLOATS = X_INSTR * 256, Z_FLOATS = Z_INSTR * 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * (X_FLOATS + Z_FLOATS)];
  float* Xs = smem;
  float* Zs = smem + 2 * X_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, fi = lane & 15, fg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* zero = g_wg_zero_page;

  f32x4 acc[TPW][MT][NT];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) acc[j][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int segs = (a.Wo + 31) >> 5;
  const int units = a.N * a.Ho * segs;
  const int u_begin = blockIdx.x * a.slice, u_end = min(u_begin + a.slice, units);

  auto dma_unit = [&](int u, int buf) {
    const int seg = u % segs;
    const int q = u / segs;
    const int oh = q % a.Ho, n = q / a.Ho;
    const int ow0 = seg * 32;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      if (wave + 4 * j < X_INSTR) {
        const int slot = (wave + 4 * j) * 64 + lane;
        const int pix = slot / XQ, pq = slot - pix * XQ;
        const int row = pix / PWD, col = pix - row * PWD;
        const int ih = oh + row - a.pad_t, iw = ow0 + col - a.pad_l;
        const bool ok = slot < X_SLOTS && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        const float* p = ok ? a.x + ((size_t)(n * a.H + ih) * a.W + iw) * a.Cx + 4 * pq : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(Xs + buf * X_FLOATS + (wave + 4 * j) * 256), 16, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < ZJ; ++j) {
      if (wave + 4 * j < Z_INSTR) {
        const int slot = (wave + 4 * j) * 64 + lane;
        const int pix = slot / ZQ, pq = slot - pix * ZQ;
        const int ow = ow0 + pix;
        const bool ok = slot < Z_SLOTS && ow < a.Wo;
        const float* p = ok ? a.dz + ((size_t)(n * a.Ho + oh) * a.Wo + ow) * a.Cout + 4 * pq : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(Zs + buf * Z_FLOATS + (wave + 4 * j) * 256), 16, 0, 0);
      }
    }
  };

  if (u_begin < u_end) dma_unit(u_begin, 0);
  __syncthreads();
#pragma unroll 1
  for (int u = u_begin; u < u_end; ++u) {
    const int b = (u - u_begin) & 1;
    const float* xb = Xs + b * X_FLOATS + fi;
    const float* zb = Zs + b * Z_FLOATS + fi;
    // dZ operand of the 8 k-steps (pixel 4s + fg), shared by all taps of this wave
    float bz[8][NT];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) bz[s][ni] = zb[(4 * s + fg) * CO + ni * 16];
    // X operands: every LDS read of the chunk is issued before the next chunk's DMA (see winograd_fused.hip); the taps are
    // processed in two halves to bound the live registers
    constexpr int H0 = (TPW + 1) / 2;
    float xa[H0][8][MT];
    auto load_taps = [&](int j0, int cnt) {
#pragma unroll
      for (int jj = 0; jj < H0; ++jj) {
        const int j = j0 + jj;
        const int t = wave + 4 * j;
        if (jj < cnt && t < NTAPS) {                     // wave-uniform
          const int kh = t / KK, kw = t - kh * KK;
#pragma unroll
          for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) xa[jj][s][mi] = xb[((kh * PWD) + 4 * s + fg + kw) * CX + mi * 16];
        }
      }
    };
    auto mfma_taps = [&](int j0, int cnt) {
#pragma unroll
      for (int jj = 0; jj < H0; ++jj) {
        const int j = j0 + jj;
        if (jj < cnt && wave + 4 * j < NTAPS) {
#pragma unroll
          for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
              for (int ni = 0; ni < NT; ++ni)
                acc[j][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[jj][s][mi], bz[s][ni], acc[j][mi][ni], 0, 0, 0);
        }
      }
    };
    load_taps(0, H0);
    mfma_taps(0, H0);
    load_taps(H0, TPW - H0);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < u_end) dma_unit(u + 1, b ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_taps(H0, TPW - H0);
    __syncthreads();
  }
  // D[row = ci][col = co]: lane holds rows 4*fg + r, column fi
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = wave + 4 * j;
    if (t < NTAPS) {
      float* dwt = a.dw + (size_t)t * a.Ct * a.Cout;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            atomicAdd(dwt + (size_t)(a.ci_off + mi * 16 + 4 * fg + r) * a.Cout + ni * 16 + fi, acc[j][mi][ni][r] * a.x_scale);
    }
  }
}

template <int KK, int MT, int NT>
int launch_wgrad_alltaps(WgradArgs& a, hipStream_t st) {
  const long units = (long)a.N * a.Ho * ((a.Wo + 31) / 32);
  long blocks = 1024;                                     // 4 per CU; bounds the atomics (taps x 256 x MT x NT per block)
  if (blocks > units) blocks = units;
  a.slice = (int)((units + blocks - 1) / blocks);
  blocks = (units + a.slice - 1) / a.slice;
  hipLaunchKernelGGL((wgrad_alltaps_kernel<KK, MT, NT>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  return shdr::check_launch("wgrad_alltaps_kernel");
}

template <int MT, int NT>
int dispatch_alltaps(WgradArgs& a, hipStream_t st) {
  if (a.KH == 3) return launch_wgrad_alltaps<3, MT, NT>(a, st);
  if constexpr (MT * NT <= 2) {
    if (a.KH == 5) return launch_wgrad_alltaps<5, MT, NT>(a, st);
  }
  if constexpr (MT * NT == 1) {
    if (a.KH == 7) return launch_wgrad_alltaps<7, MT, NT>(a, st);
  }
  return shdr::fail(SHDR_E_SHAPE, "wgrad_alltaps: no variant for %dx%d, %d -> %d", a.KH, a.KW, a.Cx, a.Cout);
}

// VALU fallback: one block per (pixel slice); threads stride over (tap, ci, co).
__global__ __launch_bounds__(256) void wgrad_direct_kernel(const WgradArgs a) {
  const int p_begin = blockIdx.x * a.slice;
  const int p_end = min(p_begin + a.slice, a.npix);
  const int per_tap = a.Cx * a.Cout;
  const int total = a.KH * a.KW * per_tap;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int tap = e / per_tap;
    const int r = e - tap * per_tap;
    const int ci = r / a.Cout, co = r - ci * a.Cout;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    float s = 0.f;
    for (int p = p_begin; p < p_end; ++p) {
      const int ow = p % a.Wo;
      const int q = p / a.Wo;
      const int oh = q % a.Ho;
      const int n = q / a.Ho;
      const int ih = oh * a.stride - a.pad_t + kh, iw = ow * a.stride - a.pad_l + kw;
      if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
        s = fmaf(a.x[(size_t)((n * a.H + ih) * a.W + iw) * a.Cx + ci], a.dz[(size_t)p * a.Cout + co], s);
    }
    atomicAdd(a.dw + ((size_t)tap * a.Ct + a.ci_off + ci) * a.Cout + co, s * a.x_scale);
  }
}

// Wt[kh][kw][co][ci - c_begin] = W[KH-1-kh][KW-1-kw][ci][co] * scale,  ci in [c_begin, c_begin+c_count)
__global__ __launch_bounds__(256) void filter_transform_kernel(const float* __restrict__ w,
                                                               float* __restrict__ wt, int KH, int KW,
                                                               int Cin, int Cout, int c_begin,
                                                               int c_count, float scale) {
  const long total = (long)KH * KW * Cout * c_count;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int ci = (int)(e % c_count);
    long t = e / c_count;
    const int co = (int)(t % Cout);
    t /= Cout;
    const int kw = (int)(t % KW), kh = (int)(t / KW);
    wt[e] = scale * w[(((long)(KH - 1 - kh) * KW + (KW - 1 - kw)) * Cin + c_begin + ci) * Cout + co];
  }
}

// db[c] += sum over pixels of dz[p][c]; grid-stride over pixel blocks, one atomic per block and channel
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dz, float* __restrict__ db,
                                                        long npix, int C) {
  __shared__ float part[256];
  // thread = (channel lane, pixel lane): CL channels x PL pixel lanes, CL = min(C rounded to pow2, 256)
  int CL = 1;
  while (CL < C && CL < 256) CL <<= 1;
  const int PL = 256 / CL;
  const int cl = threadIdx.x % CL, pl = threadIdx.x / CL;
  for (int c0 = 0; c0 < C; c0 += CL) {
    const int c = c0 + cl;
    float s = 0.f;
    if (c < C)
      for (long p = (long)blockIdx.x * PL + pl; p < npix; p += (long)gridDim.x * PL) s += dz[p * C + c];
    part[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && c < C) {
      float tsum = s;
      for (int j = 1; j < PL; ++j) tsum += part[j * CL + cl];
      atomicAdd(db + c, tsum);
    }
    __syncthreads();
  }
}

template <int BMc, int BNc, int WM, int WN, int PREC>
int launch_wgrad_prec(WgradArgs& a, hipStream_t st) {
  a.tiles_m = (a.Cx + BMc - 1) / BMc;
  a.tiles_n = (a.Cout + BNc - 1) / BNc;
  constexpr int lds = 2 * PK * (BMc + BNc) * 4;
  static long slots_of[shdr::kMaxDevices] = {};
  const int dev_slot = shdr::device_slot();
  if (slots_of[dev_slot] == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_mfma_kernel<BMc, BNc, WM, WN, PREC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    slots_of[dev_slot] = shdr::block_slots(wgrad_mfma_kernel<BMc, BNc, WM, WN, PREC>, 256, lds);
    if (slots_of[dev_slot] == 0) return shdr::fail(SHDR_E_ARCH, "wgrad: occupancy query failed");
  }
  // pixel slices: one round of the chip's block slots, at least 1024 pixels per slice to bound the atomics
  const long tiles = (long)a.KH * a.KW * a.tiles_m * a.tiles_n;
  long slice = shdr::slice_for_rounds(slots_of[dev_slot], tiles, a.npix, 1024);
  if (SHDR_ENV("SHDR_WGRAD_LEGACY_GRID")) {
    const long want = (256L * 8 + tiles - 1) / tiles;
    slice = (a.npix + want - 1) / want;
    if (slice < 2048) slice = 2048;
  }
  slice = (slice + PK - 1) / PK * PK;
  a.slice = (int)slice;
  a.nslices = (int)((a.npix + slice - 1) / slice);
  if (a.nslices > 65535) return shdr::fail(SHDR_E_SHAPE, "wgrad: too many pixel slices");
  hipLaunchKernelGGL((wgrad_mfma_kernel<BMc, BNc, WM, WN, PREC>), dim3((unsigned)tiles, (unsigned)a.nslices), dim3(256),
                     lds, st, a);
  return shdr::check_launch("wgrad_mfma_kernel");
}

template <int BMc, int BNc, int WM, int WN>
int launch_wgrad(WgradArgs& a, hipStream_t st) {
  if (a.prec == 1) return launch_wgrad_prec<BMc, BNc, WM, WN, 1>(a, st);
  if (a.prec == 2) return launch_wgrad_prec<BMc, BNc, WM, WN, 2>(a, st);
  return launch_wgrad_prec<BMc, BNc, WM, WN, 0>(a, st);
}

template <int BMc>
int dispatch_n(WgradArgs& a, hipStream_t st) {
  // (ci tile, co tile) -> computing-wave layout WM x WN; every tile divides its channel count
  if (a.Cout % 128 == 0) return launch_wgrad<BMc, 128, (BMc >= 32 ? 2 : 1), (BMc >= 32 ? 2 : 4)>(a, st);
  if (a.Cout % 64 == 0) return launch_wgrad<BMc, 64, (BMc >= 32 ? 2 : 1), (BMc >= 32 ? 2 : 4)>(a, st);
  if (a.Cout % 32 == 0) return launch_wgrad<BMc, 32, (BMc >= 32 ? 2 : 1), 2>(a, st);
  return launch_wgrad<BMc, 16, (BMc >= 64 ? 4 : BMc / 16), 1>(a, st);
}

// 96 input channels (the 7x7 / 2 stem of the Linearization-Net on the 96-channel frontend tensor): one 96-ci tile instead
// of three 32-ci tiles -- 52 instead of 94 bytes of LDS-DMA per kFLOP (the 32 x 64 tile sits at the LDS-DMA ceiling)
int dispatch_n96(WgradArgs& a, hipStream_t st) {
  if (a.Cout % 128 == 0) return launch_wgrad<96, 128, 2, 2>(a, st);
  if (a.Cout % 64 == 0) return launch_wgrad<96, 64, 2, 2>(a, st);
  return launch_wgrad<96, 32, 2, 2>(a, st);
}

}  // namespace

extern "C" int shdr_conv2d_wgrad_f32(const shdr_conv2d_desc* d, const float* x, int which, const float* dz,
                                     float* dw, void* stream) {
  SHDR_REQUIRE(d && x && dz && dw, SHDR_E_NULL, "wgrad: null pointer");
  SHDR_REQUIRE(which == 0 || which == 1, SHDR_E_SHAPE, "wgrad: which must be 0 (x1) or 1 (x2)");
  SHDR_REQUIRE(which == 0 || d->C2 > 0, SHDR_E_SHAPE, "wgrad: no second source");
  SHDR_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C1 > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 &&
                   d->stride > 0 && d->Ho > 0 && d->Wo > 0, SHDR_E_SHAPE, "wgrad: non-positive dimension");
  SHDR_REQUIRE((long)d->N * d->Ho * d->Wo < (1L << 31) && (long)d->N * d->H * d->W < (1L << 31), SHDR_E_SHAPE,
               "wgrad: more than 2^31 pixels");
  WgradArgs a{};
  a.x = x; a.dz = dz; a.dw = dw;
  a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cx = which ? d->C2 : d->C1;
  a.Ct = d->C1 + d->C2;
  a.ci_off = which ? d->C1 : 0;
  a.Cout = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.Ho = d->Ho; a.Wo = d->Wo;
  a.npix = d->N * d->Ho * d->Wo;
  a.x_scale = which ? d->x2_scale : 1.0f;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool mfma_ok = (a.Cx % 16 == 0) && (a.Cout % 16 == 0) && shdr::aligned16(x) && shdr::aligned16(dz);
  a.prec = (d->algo == SHDR_ALGO_MFMA_F16 || d->algo == SHDR_ALGO_AUTO_F16) ? 1
           : (d->algo == SHDR_ALGO_MFMA_BF16 || d->algo == SHDR_ALGO_AUTO_BF16) ? 2 : 0;
  // (also in the reduced-precision modes: the exact-fp32 all-taps kernel is 2x faster than the fp16-operand narrow kernel,
  //  which sits at the same LDS-DMA ceiling, and errs on the accurate side)
  const bool auto_algo = d->algo == SHDR_ALGO_AUTO || d->algo == SHDR_ALGO_AUTO_F16 || d->algo == SHDR_ALGO_AUTO_BF16;
  if (mfma_ok && auto_algo && a.stride == 1 && (a.Cx == 16 || a.Cx == 32) &&
      (a.Cout == 16 || a.Cout == 32) && a.KH == a.KW && (a.KH == 3 || a.KH == 5 || a.KH == 7) && a.Ho == a.H && a.Wo == a.W &&
      (a.KH == 3 || (a.KH == 5 && a.Cx * a.Cout <= 512) || (a.KH == 7 && a.Cx == 16 && a.Cout == 16)) &&   // <= 256 VGPRs
      SHDR_ENV("SHDR_NO_ALLTAPS") == nullptr) {
    if (a.Cx == 16) return a.Cout == 16 ? dispatch_alltaps<1, 1>(a, st) : dispatch_alltaps<1, 2>(a, st);
    return a.Cout == 16 ? dispatch_alltaps<2, 1>(a, st) : dispatch_alltaps<2, 2>(a, st);
  }
  if (mfma_ok && d->algo != SHDR_ALGO_DIRECT) {
    if (a.Cx % 128 == 0) return dispatch_n<128>(a, st);
    if (a.Cx % 64 == 0) return dispatch_n<64>(a, st);
    if (a.Cx % 96 == 0 && a.Cout % 32 == 0 && SHDR_ENV("SHDR_NO_WGRAD96") == nullptr) return dispatch_n96(a, st);
    if (a.Cx % 32 == 0) return dispatch_n<32>(a, st);
    return dispatch_n<16>(a, st);
  }
  SHDR_REQUIRE(d->algo != SHDR_ALGO_MFMA && d->algo != SHDR_ALGO_MFMA_F16 && d->algo != SHDR_ALGO_MFMA_BF16, SHDR_E_ALIGN,
               "wgrad: MFMA path needs Cin%%16==0 and Cout%%16==0");
  a.slice = 1024;
  a.nslices = (a.npix + a.slice - 1) / a.slice;
  hipLaunchKernelGGL(wgrad_direct_kernel, dim3((unsigned)a.nslices), dim3(256), 0, st, a);
  return shdr::check_launch("wgrad_direct_kernel");
}

extern "C" int shdr_filter_transform_f32(const float* w, float* wt, int KH, int KW, int Cin, int Cout,
                                         int c_begin, int c_count, float scale, void* stream) {
  SHDR_REQUIRE(w && wt, SHDR_E_NULL, "filter_transform: null pointer");
  SHDR_REQUIRE(KH > 0 && KW > 0 && Cin > 0 && Cout > 0 && c_begin >= 0 && c_count > 0 && c_begin + c_count <= Cin,
               SHDR_E_SHAPE, "filter_transform: bad shape");
  const long total = (long)KH * KW * Cout * c_count;
  hipLaunchKernelGGL(filter_transform_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), w, wt, KH, KW, Cin, Cout, c_begin, c_count, scale);
  return shdr::check_launch("filter_transform");
}

extern "C" int shdr_bias_grad_f32(const float* dz, float* db, int64_t npix, int C, void* stream) {
  SHDR_REQUIRE(dz && db, SHDR_E_NULL, "bias_grad: null pointer");
  SHDR_REQUIRE(npix > 0 && C > 0, SHDR_E_SHAPE, "bias_grad: bad shape");
  int CL = 1;
  while (CL < C && CL < 256) CL <<= 1;
  const int PL = 256 / CL;
  long g = (npix + (long)PL * 64 - 1) / ((long)PL * 64);
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dz,
                     db, (long)npix, C);
  return shdr::check_launch("bias_grad");
}
