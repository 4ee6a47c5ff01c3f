// Convolution forward for the SingleHDR hot path on gfx950 (MI355X).
//
//  * conv_mfma_dma_kernel -- exact-fp32 implicit GEMM on v_mfma_f32_16x16x4_f32, k-chunks staged HBM/L2 -> LDS by
//      global_load_lds (the general kernel: any filter shape / stride, two concatenated sources, fused epilogue);
//      PREC = 1 / 2 packs fp16 / bf16 operands for v_mfma_f32_16x16x32_* (BASELINE configs[4]).
//  * conv_mfma_kernel     -- its register-staged twin (x2_scale != 1, SHDR_ALGO_MFMA_REG).
//      GEMM view  D[cout][pixel] = sum_k W[k][cout] * X[pixel][k],
//      k = (tap, cin) with cin contiguous (HWIO filters need no re-layout).
//      A 2-D pixel tile (BM/16 rows x 16 columns) keeps the 3x3/5x5/7x7 halo of
//      one tile inside the CU's L1; the im2col row of a pixel is never
//      materialised: each k-chunk of 32 channels of one tap is one 128-byte
//      line per pixel.  Channel concat (two sources), the skip-scale of
//      hallucination_net.skipLayer, bias, activation, folded inference BN,
//      residual add and a second activation are fused.
//  * conv_rega_kernel     -- narrow layers (Cout <= 32, <= 16 channels per source): activations global -> VGPR, DPP
//      row shifts along the filter row, filter resident in LDS, persistent blocks.
//  * conv_direct_kernel   -- VALU direct convolution for the shapes the MFMA tile cannot fill (Cin = 3/6/9, Cout = 3).
//  (The 3x3 / stride-1 layers with Cin % 8 == 0 and Cout % 64 == 0 take winograd_fused.hip instead.)
//
// Replaces the TF op call sites listed at shdr_conv2d_fwd_f32 in include/shdr.h.
#include <stdlib.h>
#include <type_traits>

#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct ConvArgs {
  const float* x1;
  const float* x2;
  const float* w;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* res;
  float* y;
  int N, H, W, C1, C2, Ct, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo;
  int K;         // KH*KW*Ct
  int ntaps;     // KH*KW
  int nchunks;   // number of 32-wide k chunks
  int chunked;   // Ct % 32 == 0: a chunk never straddles a tap
  int tiles_x, tiles_y;
  int nblk_m, nblk_n;
  float x2_scale;
  int act1, act2, res_cs, y_cs;
  long w_bstride;  // filter elements between consecutive images (0: one filter for the batch)
  int no_dma;      // force the register-staged kernel (SHDR_ALGO_MFMA_REG)
  int cout_valid;  // channels actually stored (<= Cout; the filter may be zero-padded to Cout)
  int YH, YW, ys, yoh, yow;   // geometry of the OUTPUT tensor: pixel (oh, ow) of the conv lands at (oh * ys + yoh, ow * ys + yow) of a
                              // [N, YH, YW, y_cs] tensor (ys = 1, offsets 0, YH = Ho, YW = Wo: dense; ys = 2: the phases of a strided
                              // input gradient / of an up-sampling conv are written in place, interleaved)
  int prec;        // 0: exact fp32 MFMA; 1: fp16 / 2: bf16 MFMA operands (fp32 in HBM and LDS, fp32 accumulate)
  int legacy_epilogue;   // SHDR_CONV_LEGACY_EPILOGUE: store straight from the accumulator layout (comparison)
  unsigned* yr;          // range slot of the output (conv_x3.hip "Range"): atomicMax of max |y| over the stored values, or null
};

// max |y| -> the range slot, only if it exceeds what the slot holds (agent-scope load).  BLOCK: the four waves' maxima meet in LDS and
// one thread issues the atomicMax (conv_x3.hip x3_range_out: same-address atomics execute one after the other at the memory side); the
// call must then be reached by every wave of the block exactly once per kernel.  Otherwise one atomicMax per wave and call.
// `seen` = the slot's value loaded BEFORE the epilogue's stores (conv_range_seen): a load at the end of the block kept its LDS and
// registers allocated for a round trip to the memory side, ~2 us on blocks that live 10 - 20 us.
__device__ __forceinline__ unsigned conv_range_seen(const unsigned* slot) {
  return slot ? __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}
template <bool BLOCK>
__device__ __forceinline__ void conv_range_out(unsigned* slot, float m, unsigned seen) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  unsigned b = __float_as_uint(m);
  if constexpr (BLOCK) {
    __shared__ unsigned wave_max[4];
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = b;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (threadIdx.x != 0) return;
    const unsigned b01 = wave_max[0] > wave_max[1] ? wave_max[0] : wave_max[1], b23 = wave_max[2] > wave_max[3] ? wave_max[2] : wave_max[3];
    b = b01 > b23 ? b01 : b23;
  } else if ((threadIdx.x & 63) != 0) {
    return;
  }
  if (b > seen) atomicMax(slot, b);
}

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int BK = 32;
constexpr int SA = 34;  // A-tile row stride in dwords: bank(2*i+g) conflict-free, 8-byte aligned rows

__host__ __device__ constexpr int sb_stride(int bn) { return (bn < 32 ? 32 : bn) + 16; }

template <int BM, int BN>
__host__ __device__ constexpr int conv_lds_bytes() {
  return 2 * (BM * SA + BK * sb_stride(BN)) * 4;
}

// Bijective XCD-aware remap: consecutive logical ids share an XCD (and its L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Epilogue shared by the MFMA kernels: lane (fi, fg) holds, per 16x16 tile, 4 consecutive
// couts (4*fg..4*fg+3) of pixel fi.
template <int MT, int NT, bool RANGE_BLOCK = true>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[MT][NT], int img, int oh0,
                                              int ow0, int n0, int wm, int wn, int fi, int fg) {
  float ym = 0.0f;
  const unsigned yr_seen = conv_range_seen(a.yr);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int r = wm * MT * 16 + mi * 16 + fi;
    const int oh = oh0 + (r >> 4), ow = ow0 + (r & 15);
    if (oh >= a.Ho || ow >= a.Wo) continue;
    const long pix = ((long)img * a.YH + oh * a.ys + a.yoh) * a.YW + ow * a.ys + a.yow;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int co = n0 + wn * NT * 16 + ni * 16 + 4 * fg;
      if (co >= a.cout_valid) continue;
      f32x4 v = acc[mi][ni];
      if (co + 4 <= a.cout_valid) {
        if (a.bias) {
          const float4 b4 = *reinterpret_cast<const float4*>(a.bias + co);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
        if (a.act1 != SHDR_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = shdr::act_apply(v[e], a.act1);
        }
        if (a.scale) {
          const float4 s4 = *reinterpret_cast<const float4*>(a.scale + co);
          const float4 t4 = *reinterpret_cast<const float4*>(a.shift + co);
          v[0] = v[0] * s4.x + t4.x; v[1] = v[1] * s4.y + t4.y;
          v[2] = v[2] * s4.z + t4.z; v[3] = v[3] * s4.w + t4.w;
        }
        if (a.res) {
          const float4 r4 = *reinterpret_cast<const float4*>(a.res + (size_t)pix * a.res_cs + co);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        if (a.act2 != SHDR_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = shdr::act_apply(v[e], a.act2);
        }
        *reinterpret_cast<float4*>(a.y + (size_t)pix * a.y_cs + co) = make_float4(v[0], v[1], v[2], v[3]);
        if (a.yr) ym = fmaxf(fmaxf(fmaxf(fmaxf(ym, fabsf(v[0])), fabsf(v[1])), fabsf(v[2])), fabsf(v[3]));
      } else {  // ragged tail of a zero-padded filter (e.g. Cout = 3): scalar, unaligned-safe
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (co + e >= a.cout_valid) break;
          float t = v[e];
          if (a.bias) t += a.bias[co + e];
          t = shdr::act_apply(t, a.act1);
          if (a.scale) t = t * a.scale[co + e] + a.shift[co + e];
          if (a.res) t += a.res[(size_t)pix * a.res_cs + co + e];
          t = shdr::act_apply(t, a.act2);
          a.y[(size_t)pix * a.y_cs + co + e] = t;
          if (a.yr) ym = fmaxf(ym, fabsf(t));
        }
      }
    }
  }
  if (a.yr) conv_range_out<RANGE_BLOCK>(a.yr, ym, yr_seen);
}

// Row-contiguous epilogue for the LDS-DMA kernel (BN >= 32, every output channel stored): in the MFMA accumulator layout a
// lane holds 4 couts of one pixel, so a wave store instruction writes 64-byte pieces at a stride of Cout*4 bytes -- the
// output-bound layers (1x1 convs with few input channels, the 4 -> 64 image layers) ran at 1.0-1.4 TB/s of stores.  Here the
// block's 128 x BN tile is staged through LDS (the pipeline buffers are dead), and each thread then owns consecutive 16-byte
// pieces of one pixel row: a wave instruction writes 1 KiB made of BN*4-byte contiguous runs, and bias / scale / shift /
// residual are read the same coalesced way.
template <int BM, int BN, int MT, int NT>
__device__ __forceinline__ void conv_epilogue_staged(const ConvArgs& a, f32x4 (&acc)[MT][NT], float* stage, int img, int oh0,
                                                     int ow0, int n0, int wm, int wn, int fi, int fg, int tid) {
  constexpr int RS = BN + 4;                           // row stride in floats (16-byte aligned, rows land on shifted banks)
  __syncthreads();                                     // every wave is done with the pipeline buffers
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int r = wm * MT * 16 + mi * 16 + fi;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
      *reinterpret_cast<f32x4*>(stage + r * RS + wn * NT * 16 + ni * 16 + 4 * fg) = acc[mi][ni];
  }
  __syncthreads();
  constexpr int QR = BN / 4;                           // float4 per tile row
  constexpr int TOTAL = BM * QR;
  static_assert(256 % QR == 0, "a thread keeps its cout quad over the store loop");
  // 256 % QR == 0: the thread's cout quad q is the same in every iteration, so bias / scale / shift are loaded ONCE in front of the
  // store loop (a load inside it is followed by s_waitcnt vmcnt(0), which also sits out the stores issued before it)
  const int q = tid % QR, co = n0 + 4 * q;
  const float4 b4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 s4 = a.scale ? *reinterpret_cast<const float4*>(a.scale + co) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 t4 = a.scale ? *reinterpret_cast<const float4*>(a.shift + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  float ym = 0.0f;
  const unsigned yr_seen = conv_range_seen(a.yr);
#pragma unroll 4
  for (int e = tid; e < TOTAL; e += 256) {
    const int r = e / QR;
    const int oh = oh0 + (r >> 4), ow = ow0 + (r & 15);
    if (oh >= a.Ho || ow >= a.Wo) continue;
    const size_t pix = ((size_t)img * a.YH + oh * a.ys + a.yoh) * a.YW + ow * a.ys + a.yow;
    float4 v = *reinterpret_cast<const float4*>(stage + r * RS + 4 * q);
    if (a.bias) { v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
    if (a.act1 != SHDR_ACT_NONE) {
      v.x = shdr::act_apply(v.x, a.act1); v.y = shdr::act_apply(v.y, a.act1);
      v.z = shdr::act_apply(v.z, a.act1); v.w = shdr::act_apply(v.w, a.act1);
    }
    if (a.scale) {
      v.x = v.x * s4.x + t4.x; v.y = v.y * s4.y + t4.y; v.z = v.z * s4.z + t4.z; v.w = v.w * s4.w + t4.w;
    }
    if (a.res) {
      const float4 r4 = *reinterpret_cast<const float4*>(a.res + pix * a.res_cs + co);
      v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
    }
    if (a.act2 != SHDR_ACT_NONE) {
      v.x = shdr::act_apply(v.x, a.act2); v.y = shdr::act_apply(v.y, a.act2);
      v.z = shdr::act_apply(v.z, a.act2); v.w = shdr::act_apply(v.w, a.act2);
    }
    *reinterpret_cast<float4*>(a.y + pix * a.y_cs + co) = v;
    if (a.yr) ym = fmaxf(fmaxf(fmaxf(fmaxf(ym, fabsf(v.x)), fabsf(v.y)), fabsf(v.z)), fabsf(v.w));
  }
  if (a.yr) conv_range_out<true>(a.yr, ym, yr_seen);
}

// FAST: (C1+C2) % 32 == 0 and the two sources split on a 32-channel boundary, so the tap,
// the channel chunk and the source of a k-chunk are wave-uniform scalars and chunks run
// channel-chunk-outer / tap-inner (the taps of one 128-byte channel slice re-hit L1/L2).
// !FAST: natural k order with per-thread (tap, channel) state; any C1 % 4 == C2 % 4 == 0.
template <int BM, int BN, int WM, int WN, bool FAST>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int TH = BM / 16;          // pixel tile = TH rows x 16 columns
  constexpr int MT = BM / WM / 16;     // 16-pixel groups per wave
  constexpr int NT = BN / WN / 16;     // 16-cout groups per wave
  constexpr int SB = sb_stride(BN);
  constexpr int AROWS = BM / 32;       // A quads per thread per chunk
  constexpr int BQ = BN / 4;           // quads per B row
  constexpr int BROWS_PER_PASS = 256 / BQ;
  constexpr int BPASS = (BK + BROWS_PER_PASS - 1) / BROWS_PER_PASS;
  constexpr int KSTEPS = BK / 4;
  constexpr bool BFULL = (BROWS_PER_PASS * BPASS == BK);  // every thread's B rows are inside the chunk

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][BM][SA]
  float* Bs = smem + 2 * BM * SA;      // [2][BK][SB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int L = xcd_remap(blockIdx.x, a.nblk_m * a.nblk_n);
  const int pn = L % a.nblk_n;
  int pm = L / a.nblk_n;
  const int tx = pm % a.tiles_x;
  pm /= a.tiles_x;
  const int ty = pm % a.tiles_y;
  const int img = pm / a.tiles_y;
  const int n0 = pn * BN;
  const int oh0 = ty * TH, ow0 = tx * 16;

  // ---- per-thread load geometry, fixed over the K loop ----------------------
  const int aj = tid & 7;    // quad slot inside the 32-channel chunk
  const int ar0 = tid >> 3;  // 0..31
  int ihb[AROWS], iwb[AROWS];
  unsigned rowoff1[AROWS], rowoff2[AROWS];  // element offset of (pixel, channel 0) in x1 / x2
  const unsigned img_base = (unsigned)img * (unsigned)(a.H * a.W);
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int r = ar0 + 32 * i;
    const int oh = oh0 + (r >> 4), ow = ow0 + (r & 15);
    const bool ok = (oh < a.Ho) && (ow < a.Wo);
    ihb[i] = ok ? oh * a.stride - a.pad_t : -(1 << 28);
    iwb[i] = ow * a.stride - a.pad_l;
    const unsigned pix = ok ? img_base + (unsigned)(ihb[i] * a.W + iwb[i]) : 0u;  // may wrap for taps < 0: only used when in bounds
    rowoff1[i] = pix * (unsigned)a.C1;
    rowoff2[i] = pix * (unsigned)a.C2;
  }
  const int bq = tid % BQ, bk0 = tid / BQ;
  unsigned wrow[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) wrow[i] = (unsigned)(bk0 + BROWS_PER_PASS * i) * (unsigned)a.Cout + (unsigned)(n0 + 4 * bq);

  // Register staging of the next chunk: unconditional loads from clamped addresses issue
  // back to back; validity and the x2 scale are applied when the registers go to LDS.
  float4 areg[AROWS];
  float4 breg[BPASS];
  bool aok[AROWS], bok[BPASS];
  float ascale = 1.0f;
  // position of the next chunk: scalars when FAST, per-thread otherwise
  int nx_kh = 0, nx_kw = 0, nx_c = FAST ? 0 : 4 * aj, nx_kc = 0;
  if (!FAST) {  // Ct < 32: the first chunk already spans several taps
    while (nx_c >= a.Ct && nx_kh < a.KH) {
      nx_c -= a.Ct;
      if (++nx_kw == a.KW) { nx_kw = 0; ++nx_kh; }
    }
  }

  auto load_next = [&]() {
    const int kh = nx_kh, kw = nx_kw;
    bool kvalid, second;
    unsigned krow0;
    int cc;
    if (FAST) {
      kvalid = true;
      second = nx_c >= a.C1;                          // uniform
      cc = (second ? nx_c - a.C1 : nx_c) + 4 * aj;
      krow0 = (unsigned)((kh * a.KW + kw) * a.Ct + nx_c);
      if (++nx_kw == a.KW) {
        nx_kw = 0;
        if (++nx_kh == a.KH) { nx_kh = 0; nx_c += BK; }
      }
    } else {
      kvalid = kh < a.KH;
      second = kvalid && (nx_c >= a.C1);
      cc = kvalid ? (second ? nx_c - a.C1 : nx_c) : 0;
      krow0 = (unsigned)(nx_kc * BK);
      nx_c += BK;                                      // advance my quad by 32 k positions
      while (nx_c >= a.Ct && nx_kh < a.KH) {
        nx_c -= a.Ct;
        if (++nx_kw == a.KW) { nx_kw = 0; ++nx_kh; }
      }
    }
    ++nx_kc;
    const float* src = second ? a.x2 : a.x1;
    const unsigned delta = (unsigned)((kh * a.W + kw) * (second ? a.C2 : a.C1) + cc);
    ascale = second ? a.x2_scale : 1.0f;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      aok[i] = kvalid && (unsigned)(ihb[i] + kh) < (unsigned)a.H && (unsigned)(iwb[i] + kw) < (unsigned)a.W;
      const unsigned off = aok[i] ? (second ? rowoff2[i] : rowoff1[i]) + delta : 0u;
      areg[i] = *reinterpret_cast<const float4*>(src + (size_t)off);
    }
    const unsigned wbase = krow0 * (unsigned)a.Cout;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int bk = bk0 + BROWS_PER_PASS * i;
      bok[i] = (BFULL || bk < BK) && (FAST || (int)krow0 + bk < a.K);
      breg[i] = *reinterpret_cast<const float4*>(a.w + (size_t)img * a.w_bstride + (size_t)(bok[i] ? wbase + wrow[i] : 0u));
    }
  };

  auto store_chunk = [&](int buf) {
    float* Ab = As + buf * BM * SA;
    float* Bb = Bs + buf * BK * SB;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      float* p = Ab + (ar0 + 32 * i) * SA + 4 * aj;
      const float4 v = areg[i];
      *reinterpret_cast<float2*>(p) = aok[i] ? make_float2(v.x * ascale, v.y * ascale) : make_float2(0.f, 0.f);
      *reinterpret_cast<float2*>(p + 2) = aok[i] ? make_float2(v.z * ascale, v.w * ascale) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int bk = bk0 + BROWS_PER_PASS * i;
      if (BFULL || bk < BK)
        *reinterpret_cast<float4*>(Bb + bk * SB + 4 * bq) = bok[i] ? breg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fg = lane >> 4;
  const int a_off = (wm * MT * 16 + fi) * SA + fg;   // + mi*16*SA + 4*s
  const int b_off = fg * SB + wn * NT * 16 + fi;     // + 4*s*SB + ni*16

  // One k-chunk of MFMAs on LDS buffer `buf`.  WITH_NEXT: the next chunk's global loads are
  // issued in the shadow of k-step 0's MFMAs and written to the other LDS buffer in the
  // shadow of k-step KSTEPS-2's, so the loop body is straight-line code.
  auto compute_chunk = [&](int buf, auto with_next) {
    constexpr bool WITH_NEXT = decltype(with_next)::value;
    const float* Ab = As + buf * BM * SA + a_off;
    const float* Bb = Bs + buf * BK * SB + b_off;
    float af[2][MT], bf[2][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) af[0][mi] = Ab[mi * 16 * SA];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) bf[0][ni] = Bb[ni * 16];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int cur = s & 1;
      if (s + 1 < KSTEPS) {  // fragments of k-step s+1 are in flight while step s runs on the matrix pipe
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) af[cur ^ 1][mi] = Ab[mi * 16 * SA + 4 * (s + 1)];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) bf[cur ^ 1][ni] = Bb[4 * (s + 1) * SB + ni * 16];
      }
      if constexpr (WITH_NEXT) {
        if (s == 1) load_next();
        if (s == KSTEPS - 1) store_chunk(buf ^ 1);
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[cur][ni], af[cur][mi], acc[mi][ni], 0, 0, 0);
    }
  };

  load_next();
  store_chunk(0);
  __syncthreads();
#pragma unroll 1
  for (int kc = 0; kc + 1 < a.nchunks; ++kc) {
    compute_chunk(kc & 1, std::true_type{});
    __syncthreads();
  }
  compute_chunk((a.nchunks - 1) & 1, std::false_type{});

  conv_epilogue<MT, NT>(a, acc, img, oh0, ow0, n0, wm, wn, fi, fg);
}

// ---------------------------------------------------------------------------
// LDS-DMA variant (global_load_lds_dwordx4): the k-chunk goes HBM/L2 -> LDS with no
// register staging, no ds_write and no per-element select.
//   * LDS images are lane-linear (a wave instruction writes 1 KiB in lane order), so the
//     bank-conflict swizzle is applied on the SOURCE side:
//       A  [BM rows][8 quads]: physical quad slot p of row r holds logical quad p ^ ((r>>1)&7);
//          fragments are read with ds_read_b128 (conflict-free for the 16-lane b128 groups);
//          lane group g owns k = {4g..4g+3, 16+4g..16+4g+3} of the chunk (any k permutation is
//          legal as long as A and B agree).
//       B  [32 rows][BN]: physical column = n ^ (16 * ((k>>2)&1))  (BN >= 32) -> the two k rows
//          of a half-wave hit disjoint 16-bank halves.
//   * padding / K tail: invalid quads read from a 16-byte zero page.
//   * needs x2_scale == 1 (the host pre-scales the x2 rows of the filter instead).
// ---------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

typedef __attribute__((address_space(1))) const void* shdr_gptr_t;
typedef __attribute__((address_space(3))) void* shdr_lptr_t;

template <int BM, int BN>
__host__ __device__ constexpr int conv_dma_lds_bytes() {
  constexpr int pipe = 2 * (BM * BK + BK * BN) * 4;
  constexpr int stage = BN >= 32 ? BM * (BN + 4) * 4 : 0;       // output tile staged for row-contiguous stores (BN >= 32)
  return pipe > stage ? pipe : stage;
}

//   * PREC = 1 / 2 (fp16 / bf16 MFMA operands, BASELINE configs[4]): the LDS image stays fp32 -- the same DMA,
//     the same fragment reads -- and the 8 k values a lane group owns in a chunk (4g..4g+3, 16+4g..16+4g+3)
//     are rounded to nearest-even and packed into ONE v_mfma_f32_16x16x32_{f16,bf16} operand: 16 MFMAs per chunk
//     instead of 128, accumulation in fp32.  The kernel is then bound by the LDS / L2 feed, not by the matrix pipe.
template <int BM, int BN, int WM, int WN, bool FAST, int PREC = 0>
__global__ __launch_bounds__(256, 2) void conv_mfma_dma_kernel(const ConvArgs a) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int TH = BM / 16;
  constexpr int MT = BM / WM / 16;
  constexpr int NT = BN / WN / 16;
  constexpr int KSTEPS = BK / 4;
  constexpr int AI = BM / 32;                      // A DMA instructions per wave per chunk (8 rows each)
  constexpr int BQ = BN / 4;                       // quads per B row
  constexpr int BI = (BK * BQ / 64 + 3) / 4;       // B DMA instructions per wave per chunk
  constexpr int B_WAVES = (BK * BQ / 64) >= 4 ? 4 : (BK * BQ / 64);  // waves that carry B (BN = 16: 2)
  constexpr bool BSWZ = BN >= 32;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][BM][32]
  float* Bs = smem + 2 * BM * BK;    // [2][32][BN]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int L = xcd_remap(blockIdx.x, a.nblk_m * a.nblk_n);
  const int pn = L % a.nblk_n;
  int pm = L / a.nblk_n;
  const int tx = pm % a.tiles_x;
  pm /= a.tiles_x;
  const int ty = pm % a.tiles_y;
  const int img = pm / a.tiles_y;
  const int n0 = pn * BN;
  const int oh0 = ty * TH, ow0 = tx * 16;
  const float* zero = g_zero_page;
  const float* wimg = a.w + (size_t)img * a.w_bstride;

  // ---- A geometry: instruction i of this wave fills rows (wave*AI + i)*8 .. +7 -------------
  int ihb[AI], iwb[AI], aq[AI];
  unsigned rowoff1[AI], rowoff2[AI];
  const unsigned img_base = (unsigned)img * (unsigned)(a.H * a.W);
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int r = (wave * AI + i) * 8 + (lane >> 3);
    aq[i] = (lane & 7) ^ ((r >> 1) & 7);            // logical quad fetched into physical slot lane&7
    const int oh = oh0 + (r >> 4), ow = ow0 + (r & 15);
    const bool ok = (oh < a.Ho) && (ow < a.Wo);
    ihb[i] = ok ? oh * a.stride - a.pad_t : -(1 << 28);
    iwb[i] = ow * a.stride - a.pad_l;
    const unsigned pix = ok ? img_base + (unsigned)(ihb[i] * a.W + iwb[i]) : 0u;
    rowoff1[i] = pix * (unsigned)a.C1 + (FAST ? 4u * (unsigned)aq[i] : 0u);
    rowoff2[i] = pix * (unsigned)a.C2 + (FAST ? 4u * (unsigned)aq[i] : 0u);
  }
  // ---- B geometry ----------------------------------------------------------------------------
  unsigned woff[BI];
  int wk[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int Q = (wave * BI + j) * 64 + lane;     // linear quad index inside the B tile
    const int k = Q / BQ, pq = Q % BQ;
    const int lq = BSWZ ? (pq ^ (4 * ((k >> 2) & 1))) : pq;
    wk[j] = k;
    woff[j] = (unsigned)k * (unsigned)a.Cout + (unsigned)(n0 + 4 * lq);
  }

  // position of the next chunk: scalars when FAST; per-instruction (tap, channel) otherwise
  int nx_kh = 0, nx_kw = 0, nx_c = 0, nx_kc = 0;
  int t_kh[AI], t_kw[AI], t_c[AI];
  if (!FAST) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      t_kh[i] = 0; t_kw[i] = 0; t_c[i] = 4 * aq[i];
      while (t_c[i] >= a.Ct && t_kh[i] < a.KH) {
        t_c[i] -= a.Ct;
        if (++t_kw[i] == a.KW) { t_kw[i] = 0; ++t_kh[i]; }
      }
    }
  }

  auto dma_next = [&](int buf) {
    float* Ab = As + buf * BM * BK + (wave * AI) * 8 * BK;   // wave-uniform
    float* Bb = Bs + buf * BK * BN + (wave * BI) * 256;
    unsigned krow0;
    if (FAST) {
      const int kh = nx_kh, kw = nx_kw;
      const bool second = nx_c >= a.C1;
      const float* src = second ? a.x2 : a.x1;
      const unsigned delta = (unsigned)((kh * a.W + kw) * (second ? a.C2 : a.C1) + (second ? nx_c - a.C1 : nx_c));
      krow0 = (unsigned)((kh * a.KW + kw) * a.Ct + nx_c);
      if (++nx_kw == a.KW) {
        nx_kw = 0;
        if (++nx_kh == a.KH) { nx_kh = 0; nx_c += BK; }
      }
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const bool ok = (unsigned)(ihb[i] + kh) < (unsigned)a.H && (unsigned)(iwb[i] + kw) < (unsigned)a.W;
        const float* p = ok ? src + (size_t)((second ? rowoff2[i] : rowoff1[i]) + delta) : zero;
        __builtin_amdgcn_global_load_lds((shdr_gptr_t)p, (shdr_lptr_t)(Ab + i * 8 * BK), 16, 0, 0);
      }
    } else {
      krow0 = (unsigned)(nx_kc * BK);
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int kh = t_kh[i], kw = t_kw[i];
        const bool kvalid = kh < a.KH;
        const bool second = kvalid && (t_c[i] >= a.C1);
        const float* src = second ? a.x2 : a.x1;
        const bool ok = kvalid && (unsigned)(ihb[i] + kh) < (unsigned)a.H && (unsigned)(iwb[i] + kw) < (unsigned)a.W;
        const unsigned off = (second ? rowoff2[i] : rowoff1[i]) +
                             (unsigned)((kh * a.W + kw) * (second ? a.C2 : a.C1) + (second ? t_c[i] - a.C1 : t_c[i]));
        const float* p = ok ? src + (size_t)off : zero;
        __builtin_amdgcn_global_load_lds((shdr_gptr_t)p, (shdr_lptr_t)(Ab + i * 8 * BK), 16, 0, 0);
        t_c[i] += BK;
        while (t_c[i] >= a.Ct && t_kh[i] < a.KH) {
          t_c[i] -= a.Ct;
          if (++t_kw[i] == a.KW) { t_kw[i] = 0; ++t_kh[i]; }
        }
      }
    }
    ++nx_kc;
    if (wave < B_WAVES) {
      const unsigned wbase = krow0 * (unsigned)a.Cout;
#pragma unroll
      for (int j = 0; j < BI; ++j) {
        const bool ok = FAST || ((int)krow0 + wk[j] < a.K);
        const float* p = ok ? wimg + (size_t)(wbase + woff[j]) : zero;
        __builtin_amdgcn_global_load_lds((shdr_gptr_t)p, (shdr_lptr_t)(Bb + j * 256), 16, 0, 0);
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fg = lane >> 4;
  int a_rd[MT][2];   // dword offsets of this lane's two A quads (k = 4fg.., 16+4fg..) per 16-pixel group
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int row = wm * MT * 16 + mi * 16 + fi;
    const int f = (row >> 1) & 7;
    a_rd[mi][0] = row * BK + 4 * (fg ^ f);
    a_rd[mi][1] = row * BK + 4 * ((fg + 4) ^ f);
  }
  int b_col[NT];
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    const int col = wn * NT * 16 + ni * 16 + fi;
    b_col[ni] = BSWZ ? (col ^ (16 * (fg & 1))) : col;
  }
  // B row of k-step s for lane group fg: 16*(s>>2) + 4*fg + (s&3)
  const int b_row0 = 4 * fg * BN;

  auto compute_chunk = [&](int buf, auto with_next) {
    constexpr bool WITH_NEXT = decltype(with_next)::value;
    const float* Ab = As + buf * BM * BK;
    const float* Bb = Bs + buf * BK * BN + b_row0;
    if constexpr (PREC != 0) {
      using frag_t = std::conditional_t<PREC == 1, f16x8, bf16x8>;
      using elem_t = std::conditional_t<PREC == 1, _Float16, __bf16>;
      float4 qa[MT][2];
      float bv[8][NT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        qa[mi][0] = *reinterpret_cast<const float4*>(Ab + a_rd[mi][0]);
        qa[mi][1] = *reinterpret_cast<const float4*>(Ab + a_rd[mi][1]);
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) bv[s][ni] = Bb[(16 * (s >> 2) + (s & 3)) * BN + b_col[ni]];
      if constexpr (WITH_NEXT) dma_next(buf ^ 1);
      frag_t pa[MT], wb[NT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        pa[mi][0] = (elem_t)qa[mi][0].x; pa[mi][1] = (elem_t)qa[mi][0].y;
        pa[mi][2] = (elem_t)qa[mi][0].z; pa[mi][3] = (elem_t)qa[mi][0].w;
        pa[mi][4] = (elem_t)qa[mi][1].x; pa[mi][5] = (elem_t)qa[mi][1].y;
        pa[mi][6] = (elem_t)qa[mi][1].z; pa[mi][7] = (elem_t)qa[mi][1].w;
      }
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int s = 0; s < 8; ++s) wb[ni][s] = (elem_t)bv[s][ni];
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          if constexpr (PREC == 1)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], pa[mi], acc[mi][ni], 0, 0, 0);
          else
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ni], pa[mi], acc[mi][ni], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
      return;
    }
    float4 qa[MT][2];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) qa[mi][0] = *reinterpret_cast<const float4*>(Ab + a_rd[mi][0]);
    float bf[2][NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) bf[0][ni] = Bb[b_col[ni]];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) qa[mi][1] = *reinterpret_cast<const float4*>(Ab + a_rd[mi][1]);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int cur = s & 1;
      if (s + 1 < KSTEPS) {
        const int s1 = s + 1;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) bf[cur ^ 1][ni] = Bb[(16 * (s1 >> 2) + (s1 & 3)) * BN + b_col[ni]];
      }
      if constexpr (WITH_NEXT) {
        if (s == 1) dma_next(buf ^ 1);
      }
      __builtin_amdgcn_s_setprio(1);   // the wave that is feeding the matrix pipe wins issue arbitration (+1.5 %)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const float4 q = qa[mi][s >> 2];
        const float av = (s & 3) == 0 ? q.x : (s & 3) == 1 ? q.y : (s & 3) == 2 ? q.z : q.w;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[cur][ni], av, acc[mi][ni], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
  };

  dma_next(0);
  __syncthreads();
#pragma unroll 1
  for (int kc = 0; kc + 1 < a.nchunks; ++kc) {
    compute_chunk(kc & 1, std::true_type{});
    __syncthreads();
  }
  compute_chunk((a.nchunks - 1) & 1, std::false_type{});

  if constexpr (BN >= 32) {
    if (a.cout_valid == a.Cout && !a.legacy_epilogue) {           // block-uniform
      conv_epilogue_staged<BM, BN, MT, NT>(a, acc, smem, img, oh0, ow0, n0, wm, wn, fi, fg, tid);
      return;
    }
  }
  conv_epilogue<MT, NT>(a, acc, img, oh0, ow0, n0, wm, wn, fi, fg);
}

// ---------------------------------------------------------------------------
// Register-A variant for the narrow full-resolution layers of the U-Nets (Dequantization- / Refinement-Net: 7x7 16->16,
// 7x7 4->16, 5x5 16->32, 3x3 32->16, 3x3 16+16->16, ...).  With Cout = 16 / 32 every activation value feeds exactly one / two
// MFMAs, and staging it through LDS-DMA runs into the LDS-DMA throughput of a CU (~30 GB/s, the same ceiling the filter
// stream of the fused Winograd kernel hit): conv_mfma_dma_kernel<128,16> moves 137 B per kFLOP and sits at 41-51 TFLOP/s.
// Here the A operand never touches LDS: lane (fi = pixel of a 16-pixel row segment, fg = channel group) loads its own
// float4 = channels 4fg..4fg+3 of pixel fi + tap straight from global memory (one fully coalesced 1 KiB wave load per
// (row segment, tap, 16-channel group)); its component s is the A value of k-step s.  The whole filter lives in LDS for
// the lifetime of a PERSISTENT block (loaded once; image [tap][16-channel group][s][fg][cout] so the four lane groups of
// a B read hit 64 consecutive floats).  Padding: rows outside the image skip the tap (wave-uniform), columns are zeroed
// per lane.  Shares ConvArgs and the fused epilogue with the other MFMA kernels.
//   CT = channels per tap: 4 (3-channel images zero-padded to 4), 16, or 32 (one source, or 16 + 16 concatenated sources).
// ---------------------------------------------------------------------------
// KK = 3 / 5 / 7 (square filter, compile-time): one activation STRIP per filter row -- the 16 pixels of the segment plus the
// KK-1 halo pixels to their right -- is loaded once (2 wave loads instead of KK) and the operand of tap kw is the strip shifted
// by kw lanes inside each 16-lane DPP row (v_mov_b32 dpp row_shl:kw, the lanes that run off the row take the halo register via
// row_shr:16-kw).  KK = 0: any filter shape, one load per tap.
template <int KK>
__device__ __forceinline__ float strip_shift(float a, float b, std::integral_constant<int, KK>) {
  if constexpr (KK == 0) {
    return a;
  } else {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x110 + (16 - KK), 0xF, 0xF, false);     // row_shr:16-kw
    return __int_as_float(__builtin_amdgcn_update_dpp(t, __float_as_int(a), 0x100 + KK, 0xF, 0xF, false));   // row_shl:kw
  }
}

template <int MT, int NT, int CT, int KK>
__global__ __launch_bounds__(256) void conv_rega_kernel(const ConvArgs a) {
  static_assert(CT == 4 || CT == 16 || CT == 32, "channels per tap");
  constexpr int G = CT == 4 ? 1 : CT / 16;           // 16-channel groups per tap
  constexpr int KS = CT == 4 ? 1 : 4;                // k-steps per group
  constexpr int COUT = NT * 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // filter image [ntaps][G][KS][4][COUT]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fg = lane >> 4;

  // ---- filter -> LDS, once per block ---------------------------------------------------------------------------------
  {
    const int total = a.ntaps * CT * COUT;
    for (int e = tid; e < total; e += 256) {
      const int co = e % COUT;
      int r = e / COUT;
      const int ch = r % CT, tap = r / CT;
      const int g = CT == 4 ? 0 : ch >> 4, c16 = ch & 15;
      const int s = CT == 4 ? 0 : c16 & 3, grp = CT == 4 ? ch : c16 >> 2;
      float v = 0.0f;
      if (co < a.Cout && ch < a.Ct)
        v = a.w[((size_t)tap * a.Ct + ch) * a.Cout + co] * ((CT == 32 && a.C2 > 0 && ch >= a.C1) ? a.x2_scale : 1.0f);
      smem[(((tap * G + g) * KS + s) * 4 + grp) * COUT + co] = v;
    }
  }
  __syncthreads();

  constexpr int TH = 4 * MT;                          // pixel tile: TH rows x 16 columns, wave w owns rows w*MT .. w*MT+MT-1
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int pm = tile;
    const int tx = pm % a.tiles_x;
    pm /= a.tiles_x;
    const int ty = pm % a.tiles_y;
    const int img = pm / a.tiles_y;
    const int oh0 = ty * TH, ow0 = tx * 16;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int iw0 = ow0 + fi - a.pad_l;               // input column of tap kw = 0
    const size_t img_base = (size_t)img * a.H * a.W;
    if constexpr (KK != 0) {
      for (int kh = 0; kh < KK; ++kh) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float* src = (CT == 32 && g == 1 && a.C2 > 0) ? a.x2 : a.x1;
          const int cs = (CT == 32 && g == 1 && a.C2 > 0) ? a.C2 : a.C1;
          const int coff = (CT == 32 && g == 1 && a.C2 == 0) ? 16 : 0;
          float4 sa[MT], sb[MT];                      // strip: pixels iw0 + fi and iw0 + 16 + fi (halo, lanes fi < KK-1)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) {
            const int ih = oh0 + wave * MT + mi + kh - a.pad_t;        // wave-uniform
            const bool row_ok = (unsigned)ih < (unsigned)a.H && (CT != 16 || 4 * fg < cs);
            const bool ok_a = row_ok && (unsigned)iw0 < (unsigned)a.W;
            const bool ok_b = row_ok && fi < KK - 1 && (unsigned)(iw0 + 16) < (unsigned)a.W;
            const size_t row = img_base + (size_t)(row_ok ? ih : 0) * a.W;
            if (CT == 4) {
              const float ta = src[(row + (ok_a ? iw0 : 0)) * cs + fg], tb = src[(row + (ok_b ? iw0 + 16 : 0)) * cs + fg];
              sa[mi] = make_float4(ok_a ? ta : 0.f, 0.f, 0.f, 0.f);
              sb[mi] = make_float4(ok_b ? tb : 0.f, 0.f, 0.f, 0.f);
            } else {
              const float4 ta = *reinterpret_cast<const float4*>(src + (row + (ok_a ? iw0 : 0)) * cs + coff + (ok_a ? 4 * fg : 0));
              const float4 tb = *reinterpret_cast<const float4*>(src + (row + (ok_b ? iw0 + 16 : 0)) * cs + coff + (ok_b ? 4 * fg : 0));
              sa[mi] = ok_a ? ta : make_float4(0.f, 0.f, 0.f, 0.f);
              sb[mi] = ok_b ? tb : make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
          auto tap = [&](auto kwc) {
            constexpr int KW_ = decltype(kwc)::value;
            const float* wl = smem + (((kh * KK + KW_) * G + g) * KS * 4) * COUT + fg * COUT + fi;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
              float bw[NT];
#pragma unroll
              for (int ni = 0; ni < NT; ++ni) bw[ni] = wl[(s * 4) * COUT + ni * 16];
#pragma unroll
              for (int mi = 0; mi < MT; ++mi) {
                const float va = s == 0 ? sa[mi].x : s == 1 ? sa[mi].y : s == 2 ? sa[mi].z : sa[mi].w;
                const float vb = s == 0 ? sb[mi].x : s == 1 ? sb[mi].y : s == 2 ? sb[mi].z : sb[mi].w;
                const float x = strip_shift(va, vb, kwc);
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[ni], x, acc[mi][ni], 0, 0, 0);
              }
            }
          };
          tap(std::integral_constant<int, 0>{});
          tap(std::integral_constant<int, 1>{});
          tap(std::integral_constant<int, 2>{});
          if constexpr (KK >= 5) { tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{}); }
          if constexpr (KK >= 7) { tap(std::integral_constant<int, 5>{}); tap(std::integral_constant<int, 6>{}); }
        }
      }
    } else {
      // (an explicit next-tap register prefetch measured 5-10 % slower than leaving the schedule to the compiler and the
      //  8-12 resident waves per CU; the same holds for a double-buffered strip in the KK != 0 branch above)
      for (int kh = 0; kh < a.KH; ++kh) {
        for (int kw = 0; kw < a.KW; ++kw) {
          const int iw = iw0 + kw;
          const bool col_ok = (unsigned)iw < (unsigned)a.W;
          const int iwc = col_ok ? iw : 0;
          const float* wl = smem + ((kh * a.KW + kw) * G) * KS * 4 * COUT + fg * COUT + fi;
  #pragma unroll
          for (int g = 0; g < G; ++g) {
            // this lane's channels of the tap: 4fg..4fg+3 of group g (CT = 4: channel fg)
            const float* src = (CT == 32 && g == 1 && a.C2 > 0) ? a.x2 : a.x1;
            const int cs = (CT == 32 && g == 1 && a.C2 > 0) ? a.C2 : a.C1;
            const int coff = (CT == 32 && g == 1 && a.C2 == 0) ? 16 : 0;
            float4 av[MT];
  #pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
              const int ih = oh0 + wave * MT + mi + kh - a.pad_t;        // wave-uniform
              const bool ok = col_ok && (unsigned)ih < (unsigned)a.H && (CT != 16 || 4 * fg < cs);
              const size_t pix = img_base + (size_t)(ok ? ih : 0) * a.W + iwc;
              if (CT == 4) {
                const float t = src[pix * cs + fg];
                av[mi] = make_float4(ok ? t : 0.f, 0.f, 0.f, 0.f);
              } else {
                const float4 t = *reinterpret_cast<const float4*>(src + pix * cs + coff + (ok ? 4 * fg : 0));
                av[mi] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
              }
            }
  #pragma unroll
            for (int s = 0; s < KS; ++s) {
              float bw[NT];
  #pragma unroll
              for (int ni = 0; ni < NT; ++ni) bw[ni] = wl[((g * KS + s) * 4) * COUT + ni * 16];
  #pragma unroll
              for (int mi = 0; mi < MT; ++mi) {
                const float x = s == 0 ? av[mi].x : s == 1 ? av[mi].y : s == 2 ? av[mi].z : av[mi].w;
  #pragma unroll
                for (int ni = 0; ni < NT; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[ni], x, acc[mi][ni], 0, 0, 0);
              }
            }
          }
        }
      }
    }
    conv_epilogue<MT, NT, false>(a, acc, img, oh0, ow0, 0, wave, 0, fi, fg);       // (per tile of the loop: per-wave range atomics)
  }
}

// ---------------------------------------------------------------------------
// VALU direct convolution: one thread = one output pixel x CPT couts.
// Filter taps are indexed uniformly across the block -> scalar loads.
// ---------------------------------------------------------------------------
template <int CPT>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvArgs a) {
  const long npix = (long)a.N * a.Ho * a.Wo;
  const int co0 = blockIdx.y * CPT;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long)gridDim.x * 256) {
    const int ow = pix % a.Wo;
    const long t = pix / a.Wo;
    const int oh = t % a.Ho;
    const int img = t / a.Ho;
    const long opix = ((long)img * a.YH + oh * a.ys + a.yoh) * a.YW + ow * a.ys + a.yow;
    float acc[CPT];
#pragma unroll
    for (int n = 0; n < CPT; ++n) acc[n] = 0.f;
    for (int kh = 0; kh < a.KH; ++kh) {
      const int ih = oh * a.stride - a.pad_t + kh;
      if ((unsigned)ih >= (unsigned)a.H) continue;
      for (int kw = 0; kw < a.KW; ++kw) {
        const int iw = ow * a.stride - a.pad_l + kw;
        if ((unsigned)iw >= (unsigned)a.W) continue;
        const long ipix = ((long)img * a.H + ih) * a.W + iw;
        const float* wt = a.w + (long)((kh * a.KW + kw) * a.Ct) * a.Cout + co0;
        const float* p1 = a.x1 + ipix * a.C1;
        for (int c = 0; c < a.C1; ++c) {
          const float xv = p1[c];
#pragma unroll
          for (int n = 0; n < CPT; ++n)
            if (co0 + n < a.Cout) acc[n] = fmaf(xv, wt[(long)c * a.Cout + n], acc[n]);
        }
        if (a.C2 > 0) {
          const float* p2 = a.x2 + ipix * a.C2;
          const float* wt2 = wt + (long)a.C1 * a.Cout;
          for (int c = 0; c < a.C2; ++c) {
            const float xv = p2[c] * a.x2_scale;
#pragma unroll
            for (int n = 0; n < CPT; ++n)
              if (co0 + n < a.Cout) acc[n] = fmaf(xv, wt2[(long)c * a.Cout + n], acc[n]);
          }
        }
      }
    }
#pragma unroll
    for (int n = 0; n < CPT; ++n) {
      const int co = co0 + n;
      if (co >= a.Cout) break;
      float v = acc[n];
      if (a.bias) v += a.bias[co];
      v = shdr::act_apply(v, a.act1);
      if (a.scale) v = v * a.scale[co] + a.shift[co];
      if (a.res) v += a.res[opix * a.res_cs + co];
      v = shdr::act_apply(v, a.act2);
      a.y[opix * a.y_cs + co] = v;
    }
  }
}

template <int BM, int BN, int WM, int WN, bool FAST, int PREC = 0>
int launch_mfma_dma_impl(ConvArgs& a, hipStream_t st) {
  constexpr int TH = BM / 16;
  a.tiles_x = (a.Wo + 15) / 16;
  a.tiles_y = (a.Ho + TH - 1) / TH;
  a.nblk_m = a.N * a.tiles_y * a.tiles_x;
  a.nblk_n = a.Cout / BN;
  constexpr int lds = conv_dma_lds_bytes<BM, BN>();
  static bool attr_done[shdr::kMaxDevices] = {};
  const int dev_slot = shdr::device_slot();
  if (!attr_done[dev_slot]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_dma_kernel<BM, BN, WM, WN, FAST, PREC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done[dev_slot] = true;
  }
  const long nblk = (long)a.nblk_m * a.nblk_n;
  if (nblk <= 0 || nblk > 0x7fffffffL) return shdr::fail(SHDR_E_SHAPE, "conv2d: grid of %ld blocks", nblk);
  hipLaunchKernelGGL((conv_mfma_dma_kernel<BM, BN, WM, WN, FAST, PREC>), dim3((unsigned)nblk), dim3(256), lds, st, a);
  return shdr::check_launch("conv_mfma_dma_kernel");
}

template <int BM, int BN, int WM, int WN, bool FAST>
int launch_mfma_impl(ConvArgs& a, hipStream_t st) {
  constexpr int TH = BM / 16;
  a.tiles_x = (a.Wo + 15) / 16;
  a.tiles_y = (a.Ho + TH - 1) / TH;
  a.nblk_m = a.N * a.tiles_y * a.tiles_x;
  a.nblk_n = a.Cout / BN;
  constexpr int lds = conv_lds_bytes<BM, BN>();
  static bool attr_done[shdr::kMaxDevices] = {};
  const int dev_slot = shdr::device_slot();
  if (!attr_done[dev_slot]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<BM, BN, WM, WN, FAST>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done[dev_slot] = true;
  }
  const long nblk = (long)a.nblk_m * a.nblk_n;
  if (nblk <= 0 || nblk > 0x7fffffffL) return shdr::fail(SHDR_E_SHAPE, "conv2d: grid of %ld blocks", nblk);
  hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, FAST>), dim3((unsigned)nblk), dim3(256), lds, st, a);
  return shdr::check_launch("conv_mfma_kernel");
}

template <int BM, int BN, int WM, int WN>
int launch_mfma(ConvArgs& a, hipStream_t st) {
  const bool fast = (a.Ct % BK == 0) && (a.C2 == 0 || a.C1 % BK == 0);
  a.chunked = fast;
  a.nchunks = fast ? a.ntaps * (a.Ct / BK) : (a.K + BK - 1) / BK;
  if (a.prec != 0) {
    if (a.x2_scale != 1.0f)
      return shdr::fail(SHDR_E_SHAPE, "conv2d: the fp16/bf16 MFMA path needs x2_scale == 1 (fold it into the filter)");
    if (a.prec == 1)
      return fast ? launch_mfma_dma_impl<BM, BN, WM, WN, true, 1>(a, st) : launch_mfma_dma_impl<BM, BN, WM, WN, false, 1>(a, st);
    return fast ? launch_mfma_dma_impl<BM, BN, WM, WN, true, 2>(a, st) : launch_mfma_dma_impl<BM, BN, WM, WN, false, 2>(a, st);
  }
  if (a.x2_scale == 1.0f && !a.no_dma)
    return fast ? launch_mfma_dma_impl<BM, BN, WM, WN, true>(a, st) : launch_mfma_dma_impl<BM, BN, WM, WN, false>(a, st);
  return fast ? launch_mfma_impl<BM, BN, WM, WN, true>(a, st) : launch_mfma_impl<BM, BN, WM, WN, false>(a, st);
}

template <int MT, int NT, int CT, int KK>
int launch_rega(ConvArgs& a, hipStream_t st) {
  constexpr int TH = 4 * MT;
  a.tiles_x = (a.Wo + 15) / 16;
  a.tiles_y = (a.Ho + TH - 1) / TH;
  const long ntiles = (long)a.N * a.tiles_y * a.tiles_x;
  const int lds = a.ntaps * CT * NT * 16 * 4;
  const int dev_slot = shdr::device_slot();
  static int attr_lds_dev[shdr::kMaxDevices] = {};
  int& attr_lds = attr_lds_dev[dev_slot];
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rega_kernel<MT, NT, CT, KK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_lds = lds;
  }
  // persistent grid = the blocks that are actually co-resident (VGPR- or LDS-limited): a grid that is not a multiple of the
  // residency runs its last blocks alone at low occupancy (measured: 768 blocks with 2 resident per CU cost +20 %)
  static int occ_lds_dev[shdr::kMaxDevices] = {}, occ_dev[shdr::kMaxDevices] = {};
  int &occ_lds = occ_lds_dev[dev_slot], &occ = occ_dev[dev_slot];     // (LDS bytes are never 0 here: 0 = not queried yet)
  if (lds != occ_lds) {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_rega_kernel<MT, NT, CT, KK>), 256, lds);
    if (e != hipSuccess || nb < 1) nb = 1;
    occ = nb > 8 ? 8 : nb;
    occ_lds = lds;
  }
  long grid = 256L * occ;
  if (const char* e = SHDR_ENV("SHDR_REGA_PER_CU")) grid = 256L * atoi(e);
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL((conv_rega_kernel<MT, NT, CT, KK>), dim3((unsigned)grid), dim3(256), lds, st, a);
  return shdr::check_launch("conv_rega_kernel");
}

// narrow stride-1 layers the register-A kernel takes: Cout <= 32, 4 / 16 / 32 channels per tap, filter <= 100 KB of LDS
inline bool rega_ok(const ConvArgs& a) {
  if (a.stride != 1 || a.w_bstride != 0 || a.Cout % 16 != 0) return false;
  // 64 output channels only from a 4-channel (RGB + zero) source: the first layer of the VGG-shaped encoders (0.76 -> 0.54 ms
  // at 16 x 512^2; a per-wave LDS-staged, row-contiguous store on top of it measured no further gain)
  if (a.Cout > 32 && !(a.Cout == 64 && a.C2 == 0 && a.C1 == 4 && a.KH == 3 && a.KW == 3 && SHDR_ENV("SHDR_NO_REGA64") == nullptr)) return false;
  const bool ct4 = a.C2 == 0 && a.C1 == 4, ct16 = a.C2 == 0 && (a.C1 == 8 || a.C1 == 12 || a.C1 == 16);
  const bool ct32 = (a.C2 == 0 && a.C1 == 32) || (a.C1 == 16 && a.C2 == 16);
  if (!(ct4 || ct16 || ct32)) return false;
  if (a.C2 == 0 && a.C1 == 32) return false;       // measured: one 32-channel source is faster on the LDS-DMA kernel
  return (long)a.ntaps * a.Ct * a.Cout * 4 <= 100 * 1024;
}

template <int NT, int CT>
int dispatch_rega_k(ConvArgs& a, hipStream_t st) {
  const bool square = a.KH == a.KW && SHDR_ENV("SHDR_REGA_NO_DPP") == nullptr;
  if (square && a.KH == 3) return launch_rega<4, NT, CT, 3>(a, st);
  if (square && a.KH == 5) return launch_rega<4, NT, CT, 5>(a, st);
  if (square && a.KH == 7) return launch_rega<4, NT, CT, 7>(a, st);
  return launch_rega<4, NT, CT, 0>(a, st);
}

template <int NT>
int dispatch_rega(ConvArgs& a, hipStream_t st) {
  if (a.Ct == 4) return dispatch_rega_k<NT, 4>(a, st);
  if (a.Ct <= 16) return dispatch_rega_k<NT, 16>(a, st);      // 8 / 12 channels: the missing lane groups read zeros
  return dispatch_rega_k<NT, 32>(a, st);
}

template <int CPT>
int launch_direct(ConvArgs& a, hipStream_t st) {
  const long npix = (long)a.N * a.Ho * a.Wo;
  long gx = (npix + 255) / 256;
  if (gx > 65535L * 16) gx = 65535L * 16;
  dim3 grid((unsigned)gx, (unsigned)((a.Cout + CPT - 1) / CPT));
  hipLaunchKernelGGL((conv_direct_kernel<CPT>), grid, dim3(256), 0, st, a);
  return shdr::check_launch("conv_direct_kernel");
}

}  // namespace

extern "C" int shdr_same_pad(int in_size, int k, int stride, int* out_size, int* pad_before) {
  if (in_size <= 0 || k <= 0 || stride <= 0) return shdr::fail(SHDR_E_SHAPE, "same_pad: bad args");
  const int out = (in_size + stride - 1) / stride;
  int total = (out - 1) * stride + k - in_size;
  if (total < 0) total = 0;
  if (out_size) *out_size = out;
  if (pad_before) *pad_before = total / 2;
  return SHDR_OK;
}

// y_range (or NULL): range slot that receives max |y| (atomicMax; see shdr_conv2d_fwd_prepared_ranged_f32) from the kernels' epilogues
extern "C" int shdr_conv2d_fwd_yrange_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2,
                                   const float* w, const float* bias, const float* scale,
                                   const float* shift, const float* residual, float* y,
                                   float* y_range, void* stream) {
  SHDR_REQUIRE(d && x1 && w && y, SHDR_E_NULL, "conv2d: null desc/x1/w/y");
  SHDR_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C1 > 0 && d->C2 >= 0 && d->Cout > 0 &&
                   d->KH > 0 && d->KW > 0 && d->stride > 0 && d->Ho > 0 && d->Wo > 0,
               SHDR_E_SHAPE, "conv2d: non-positive dimension");
  SHDR_REQUIRE((d->C2 == 0) == (x2 == nullptr), SHDR_E_NULL, "conv2d: x2 must be given iff C2 > 0");
  SHDR_REQUIRE((scale == nullptr) == (shift == nullptr), SHDR_E_NULL,
               "conv2d: scale and shift must be given together");
  SHDR_REQUIRE(d->pad_t >= 0 && d->pad_l >= 0 && d->pad_t < d->KH && d->pad_l < d->KW, SHDR_E_SHAPE,
               "conv2d: pad (%d,%d) outside kernel %dx%d", d->pad_t, d->pad_l, d->KH, d->KW);
  // every output pixel must map to a window that starts inside the padded input
  SHDR_REQUIRE((long)(d->Ho - 1) * d->stride - d->pad_t < d->H &&
                   (long)(d->Wo - 1) * d->stride - d->pad_l < d->W,
               SHDR_E_SHAPE, "conv2d: output %dx%d too large for input %dx%d", d->Ho, d->Wo, d->H, d->W);
  const int cout_valid = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  SHDR_REQUIRE(cout_valid <= d->Cout, SHDR_E_SHAPE, "conv2d: cout_valid > Cout");
  const int y_cs = d->y_cstride > 0 ? d->y_cstride : cout_valid;
  SHDR_REQUIRE(y_cs >= cout_valid, SHDR_E_SHAPE, "conv2d: y_cstride < stored channels");
  SHDR_REQUIRE(!residual || d->res_cstride >= cout_valid, SHDR_E_SHAPE, "conv2d: res_cstride < stored channels");
  SHDR_REQUIRE((long)d->N * d->H * d->W < (1L << 31) && (long)d->N * d->Ho * d->Wo < (1L << 31),
               SHDR_E_SHAPE, "conv2d: more than 2^31 pixels");
  SHDR_REQUIRE((long)d->N * d->H * d->W * (d->C1 > d->C2 ? d->C1 : d->C2) < (1L << 32) &&
                   (long)d->KH * d->KW * (d->C1 + d->C2) * d->Cout < (1L << 32) && d->w_batch_stride >= 0,
               SHDR_E_SHAPE, "conv2d: tensor with more than 2^32 elements");

  ConvArgs a{};
  a.x1 = x1; a.x2 = x2; a.w = w; a.bias = bias; a.scale = scale; a.shift = shift;
  a.res = residual; a.y = y;
  a.yr = reinterpret_cast<unsigned*>(y_range);
  a.N = d->N; a.H = d->H; a.W = d->W; a.C1 = d->C1; a.C2 = d->C2; a.Ct = d->C1 + d->C2;
  a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride;
  a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.Ho = d->Ho; a.Wo = d->Wo;
  a.ntaps = a.KH * a.KW;
  a.K = a.ntaps * a.Ct;
  a.chunked = (a.Ct % BK) == 0;
  a.nchunks = a.chunked ? a.ntaps * (a.Ct / BK) : (a.K + BK - 1) / BK;
  a.x2_scale = d->C2 > 0 ? d->x2_scale : 1.0f;
  a.act1 = d->act1; a.act2 = d->act2;
  a.res_cs = d->res_cstride; a.y_cs = y_cs; a.cout_valid = cout_valid;
  a.w_bstride = d->w_batch_stride;
  a.YH = a.Ho; a.YW = a.Wo; a.ys = 1; a.yoh = 0; a.yow = 0;
  if (d->y_pix_stride > 1) {
    SHDR_REQUIRE(d->y_H > 0 && d->y_W > 0 && d->y_off_h >= 0 && d->y_off_w >= 0 &&
                     (long)(d->Ho - 1) * d->y_pix_stride + d->y_off_h < d->y_H && (long)(d->Wo - 1) * d->y_pix_stride + d->y_off_w < d->y_W,
                 SHDR_E_SHAPE, "conv2d: strided output %dx%d (stride %d, offset %d,%d) does not fit y %dx%d", d->Ho, d->Wo,
                 d->y_pix_stride, d->y_off_h, d->y_off_w, d->y_H, d->y_W);
    SHDR_REQUIRE((long)d->N * d->y_H * d->y_W < (1L << 31), SHDR_E_SHAPE, "conv2d: more than 2^31 output pixels");
    a.YH = d->y_H; a.YW = d->y_W; a.ys = d->y_pix_stride; a.yoh = d->y_off_h; a.yow = d->y_off_w;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  a.legacy_epilogue = SHDR_ENV("SHDR_CONV_LEGACY_EPILOGUE") != nullptr;

  const bool ragged = (cout_valid % 4) != 0;  // scalar epilogue: no alignment demands on y/res/bias
  const bool mfma_ok = (a.C1 % 4 == 0) && (a.C2 % 4 == 0) && (a.Cout % 16 == 0) &&
                       (ragged || ((y_cs % 4 == 0) && (!residual || d->res_cstride % 4 == 0))) && shdr::aligned16(x1) &&
                       (!x2 || shdr::aligned16(x2)) && shdr::aligned16(w) && shdr::aligned16(y) &&
                       (!bias || shdr::aligned16(bias)) && (!scale || shdr::aligned16(scale)) &&
                       (!shift || shdr::aligned16(shift)) && (!residual || shdr::aligned16(residual));
  int algo = d->algo;
  if (algo == SHDR_ALGO_AUTO || algo == SHDR_ALGO_AUTO_EXACT) algo = mfma_ok ? SHDR_ALGO_MFMA : SHDR_ALGO_DIRECT;
  if (algo == SHDR_ALGO_AUTO_F16 || algo == SHDR_ALGO_AUTO_BF16) {   // reduced-precision operands where the MFMA path applies
    a.prec = (mfma_ok && a.x2_scale == 1.0f) ? (algo == SHDR_ALGO_AUTO_F16 ? 1 : 2) : 0;
    algo = mfma_ok ? SHDR_ALGO_MFMA : SHDR_ALGO_DIRECT;
  }
  if (algo == SHDR_ALGO_MFMA_F16 || algo == SHDR_ALGO_MFMA_BF16) { a.prec = algo == SHDR_ALGO_MFMA_F16 ? 1 : 2; algo = SHDR_ALGO_MFMA; }
  if (algo == SHDR_ALGO_MFMA_REG) { a.no_dma = 1; algo = SHDR_ALGO_MFMA; }
  if (algo == SHDR_ALGO_MFMA) {
    SHDR_REQUIRE(mfma_ok, SHDR_E_ALIGN,
                 "conv2d: MFMA path needs C1%%4==0, C2%%4==0, Cout%%16==0, 16-byte aligned tensors");
    // short-K layers (1x1 expansions 64 -> 256, 128 -> 512 of the ResNet blocks) are bound by their output stream: the 128 x 64
    // tile keeps three blocks per CU in flight instead of two (0.32 -> 0.27 ms with the fused residual, 0.19 -> 0.16 without)
    if (a.Cout % 128 == 0 && a.K > 128) return launch_mfma<128, 128, 2, 2>(a, st);
    // In the reduced-precision AUTO modes the exact-fp32 register-A kernel still takes the single-source layers with 4 / <= 16
    // channels per tap and 16 (64) couts: it is faster there than the fp16-operand LDS-DMA kernel (16 x 512^2: 7x7 4->16 0.38
    // vs 0.89 ms, 3x3 16->16 0.28 vs 0.38, 7x7 16->16 1.22 vs 1.37) and errs on the accurate side; 16+16 and 32-cout layers
    // stay on the fp16 kernel (0.53 vs 0.57, 0.24 vs 0.30 ms).
    const bool auto_reduced = d->algo == SHDR_ALGO_AUTO_F16 || d->algo == SHDR_ALGO_AUTO_BF16;
    const bool rega_prec = a.prec == 0 || (auto_reduced && a.C2 == 0 && (a.Cout == 16 || a.Cout == 64));
    const bool rega = rega_prec && !a.no_dma && d->algo != SHDR_ALGO_MFMA && rega_ok(a) && SHDR_ENV("SHDR_NO_REGA") == nullptr;
    if (rega) a.prec = 0;
    if (rega && a.Cout == 64) return launch_rega<4, 4, 4, 3>(a, st);
    if (a.Cout % 64 == 0) return launch_mfma<128, 64, 4, 1>(a, st);
    if (rega) return a.Cout == 32 ? dispatch_rega<2>(a, st) : dispatch_rega<1>(a, st);
    if (a.Cout % 32 == 0) return launch_mfma<128, 32, 4, 1>(a, st);
    return launch_mfma<128, 16, 4, 1>(a, st);
  }
  if (algo == SHDR_ALGO_DIRECT) {
    SHDR_REQUIRE(cout_valid == a.Cout, SHDR_E_SHAPE, "conv2d: direct path does not take padded filters");
    SHDR_REQUIRE(a.w_bstride == 0, SHDR_E_SHAPE, "conv2d: direct path does not take per-image filters");
    int rc;
    if (a.Cout <= 3) rc = launch_direct<3>(a, st);
    else if (a.Cout % 16 == 0) rc = launch_direct<16>(a, st);
    else rc = launch_direct<8>(a, st);
    if (rc || !y_range) return rc;
    SHDR_REQUIRE(d->y_pix_stride <= 1 && y_cs == cout_valid, SHDR_E_SHAPE, "conv2d: y_range needs a dense output");
    return shdr_absmax_f32(y, (int64_t)d->N * d->Ho * d->Wo * cout_valid, y_range, stream);     // the VALU kernel's epilogue does not track it
  }
  return shdr::fail(SHDR_E_SHAPE, "conv2d: unknown algo %d", d->algo);
}

extern "C" int shdr_conv2d_fwd_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* w, const float* bias,
                                   const float* scale, const float* shift, const float* residual, float* y, void* stream) {
  return shdr_conv2d_fwd_yrange_f32(d, x1, x2, w, bias, scale, shift, residual, y, nullptr, stream);
}
