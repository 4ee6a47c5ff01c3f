// fp32 3x3 / stride-1 convolution on the fp16 matrix pipe: every fp32 operand is split into two fp16 terms and a product is the sum of
// three v_mfma_f32_16x16x32_f16 (fp32 accumulation) -- "x3".  Replaces the same tf.keras.layers.Conv2D call sites as the fused
// Winograd kernel (hallucination_net.py:43-75,115-144, dequantization_net.py:35-46, refinement_net.py, vgg16.py:72-83: the 3x3
// layers with a multiple of 32 input channels per source and a multiple of 64 output channels), forward and input gradient.
//
// Arithmetic.  x = xh + xl * 2^-11 with xh = fp16(x), xl = fp16((x - xh) * 2^11): 22 mantissa bits, no underflow of the low term
// (it is stored scaled).  w' = w * 2^S (S per layer: max |w'| in [2^13, 2^14), so the low term of a weight is a normal fp16 number
// down to |w| = max |w| * 2^-17) = wh + wl.  Then
//     x * w' = xh * wh + xl * (wh * 2^-11) + xh * wl + (dropped: xl * wl * 2^-11, < 2^-22 |x w'|)
// i.e. three MFMAs with the filter operands wh, wh * 2^-11 (exact: a power-of-two scaling, formed in registers from the wh fragment with
// four v_pk_mul_f16 -- one LDS image and a third of the filter bytes less than a stored copy) and wl, all into ONE fp32 accumulator; the
// epilogue multiplies by 2^-S (exact).  Per-product error <= 3 * 2^-22 = 7e-7 relative, of random sign -- the level of the fp32
// rounding of a K >= 288 dot product itself; measured against the float64 oracle in tests/test_gpu_ops.py (same 1e-5 bar as the
// exact-fp32 kernels, errors reported next to the Winograd kernel's).
// Range.  fp16(x) overflows at |x| >= 65504 and loses the high term's mantissa below 2^-14, which an fp32 convolution does not.  Every
// launch therefore takes a RANGE SLOT per source -- a device word holding (the bit pattern of) an upper bound of max |x| of the tensor,
// written by the producing kernel's epilogue (y_range of the conv kernels), by shdr_absmax_f32, or known to the host -- and multiplies
// the input by the power of two 2^T that brings that bound into [2^10, 2^11) while it splits (one v_fma_mix per term: the scaling costs
// no instruction), the epilogue multiplies by 2^-T: both exact.  Elements down to 2^-25 of the tensor's maximum keep all 22 bits, smaller
// ones an absolute precision of 2^-46 of the maximum; non-finite inputs give non-finite outputs on their receptive field, as the fp32
// kernels do.  A launch without a slot (the low-level entry point only) runs unscaled.
//
// Kernel (the layout of conv_f16_w3.hip with fp32 tensors in HBM): block = 16 x 16 pixels x 64 couts, 4 waves; per 32-channel chunk
// the raw 18 x 18 fp32 patch is loaded ONCE into registers (coalesced 128-byte lines, issued under the MFMAs of the previous
// chunk), split, and written as two fp16 images of 64-byte rows (16-byte slot XOR-swizzled: conflict-free ds_read_b128); the three
// two filter images stream through LDS per tap (8 KB, register-staged one tap ahead, double-buffered).  Per tap and wave: 16 operand
// reads feed 48 MFMAs.  58 KB of LDS, two blocks per CU.
// Fused around it: MaxPool2D(2) in the epilogue (the conv + max-pool pairs of hallucination_net.py:43-75 / vgg16.py:72-83: the 2 x 2
// window is two accumulator rows of a lane and its neighbour lane) and tf.image.resize(x, 2x, BILINEAR) in the prologue (UP = true;
// hallucination_net.py:86-88, dequantization_net.py:25-27): the block loads the 10 x 10 LOW-RES patch of a chunk (a quarter of the
// bytes), parks it in 12.5 KB of LDS and builds the 18 x 18 up-sampled patch from it with resize2x_kernel's arithmetic and order
// (bit-identical to the two-kernel path) on the way into the fp16 images.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include <type_traits>

#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int X3_HEADER_FLOATS = 16;                           // [0] max |w| (bits), [1] 2^-S; 64 bytes keep the images 16-byte aligned
constexpr int BN = 64, NT = 4, MT = 4;
constexpr int IMG_HALVES = BN * 32;                            // one filter image of a tap: [64 couts][32 channels]
constexpr int UNIT_HALVES = 2 * IMG_HALVES;                    // wh, wl
constexpr int LRW = 10, LRPIX = LRW * LRW;                     // low-res patch of the up-sampling prologue
constexpr int LRJ = (LRPIX * 8 + 255) / 256;                   // float4 pieces per thread (4)
// KH x KW taps over a 16 x 16 output tile: raw patch (16 + KH - 1) x (16 + KW - 1).  3 x 3 = the stride-1 layers; 4 x 4, 4 x 3, 3 x 4
// (and 3 x 3) = the four phases of a 7 x 7 / stride-2 layer (see shdr_conv2d_fwd_x3_f32)
template <int KH, int KW>
struct X3G {
  static constexpr int PH = 16 + KH - 1, PWID = 16 + KW - 1, PPIX = PH * PWID;
  static constexpr int PJ = (PPIX * 8 + 255) / 256;           // float4 pieces per thread and chunk
  static constexpr int PATCH_HALVES = PPIX * 32;              // one fp16 image of the patch
  static constexpr int LDS_BYTES = (2 * PATCH_HALVES + 2 * UNIT_HALVES) * 2;
  static constexpr int LDS_BYTES_UP = LDS_BYTES + LRPIX * 32 * 4;
};

struct X3Args {
  const float* x1;
  const float* x2;
  const _Float16* wp;      // packed [Cout / 64][units = Ct / 32 * taps][2 images: wh, wl][64][32]
  const float* hdr;        // packed header: hdr[1] = 2^-S
  const float* bias;
  const float* scale;
  const float* shift;
  float* y;                // [N,H,W,Cout] (or null when only the pooled tensor is wanted)
  float* yp;               // [N,H/2,W/2,Cout] = MaxPool2D(2)(y), or null
  const float* yin;        // partial sums of earlier phases to add (same layout as y), or null
  const float* res;        // residual added between the affine and act2 (the ResNet joins of linearization_net.py:6-48), or null
  int res_cs;              // its channels per pixel
  int N, H, W, C1, C2, Cout, tiles_x, tiles_y, nblk_m, nblk_n, act1, act2;
  int Hl, Wl;              // UP: x1 is the low-res tensor [N,Hl,Wl,C1], H = 2 Hl, W = 2 Wl
  int Hin, Win;            // input tensor [N,Hin,Win,C]; tap (kh, kw) of output (oh, ow) reads input (in_s (oh + kh) + bh, in_s (ow + kw) + bw)
  int in_s, bh, bw;        // stride-1 3x3 SAME: in_s = 1, bh = bw = -1
  int final;               // 0: store the raw partial sum (another phase follows), 1: the epilogue
  int pool_avg;            // yp = AveragePooling2D(2)(y) instead of MaxPool2D(2)(y)
  const unsigned* xr1;     // range slots of the two sources (bits of an upper bound of max |x|; null: no scaling)
  const unsigned* xr2;
  unsigned* yr;            // range slot of the output: atomicMax of max |y| (null: not wanted)
  int nphase;              // MP kernel (the 7x7 / stride-2 stem in ONE launch): phase p runs pth[p] x ptw[p] taps at input offset (pbh[p], pbw[p]) with
  int pth[4], ptw[4], pbh[4], pbw[4];      // the packed sub-filter pwp[p]; all phases accumulate in the block's registers
  const _Float16* pwp[4];
  const float* proj;       // [3][64] or null: a linear map applied to the 64 output channels of every pixel in the epilogue ...
  float* yproj;            // ... yproj[N,H,W,3][j] = sum_c proj[j][c] y[c] (shdr_conv2d_fwd_x3_projected_f32; Cout = 64)
};

// 2^T and 2^-T for the power of two that brings the larger of the two bounds into [2^10, 2^11); 1 for an empty, zero or non-finite bound
__device__ __forceinline__ void x3_range_scale(const unsigned* r1, const unsigned* r2, float& xs, float& ixs) {
  xs = 1.0f;
  ixs = 1.0f;
  unsigned b = r1 ? *r1 : 0u;
  if (r2) {
    const unsigned b2 = *r2;
    b = b2 > b ? b2 : b;                                       // non-negative floats order like their bit patterns
  }
  if (b != 0u && b < 0x7f800000u) {
    int ex;
    frexpf(__uint_as_float(b), &ex);                           // bound in [2^(ex-1), 2^ex)
    int T = 11 - ex;
    T = T < -126 ? -126 : (T > 126 ? 126 : T);
    xs = ldexpf(1.0f, T);
    ixs = ldexpf(1.0f, -T);
  }
}
// one float4 -> (fp16(v xs), fp16((v xs - fp16(v xs)) 2^11)) as two packed register pairs: v_fma_mixlo/hi_f16 multiply, round and pack in
// one instruction, v_fma_mix_f32 forms v xs - (float)h exactly (the product is exact, the difference representable): 12 vector
// instructions per float4 (the compiler's cast / convert / subtract / multiply / pack sequence: 19)
__device__ __forceinline__ void x3_split4(const f32x4 v, float xs, unsigned (&h)[2], unsigned (&l)[2]) {
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
// max |y| of a block -> the range slot: the waves' maxima meet in LDS (dead after the tap loop's last barrier) and ONE thread issues a
// no-return atomicMax, and only when the block's maximum exceeds what the slot already holds.  Atomics execute at the memory side, one
// after the other per address (measured ~4 ns each): one per WAVE cost the small layers of the U-Nets 15 - 35 us per launch, all of it
// in the first round of blocks, which finish together and all still see the slot empty.
// `seen` = the slot's value loaded at the START of the epilogue (the load's round trip to the memory side runs under the stores)
__device__ __forceinline__ void x3_range_out(unsigned* slot, float m, int lane, int wave, unsigned seen, unsigned* lds) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) lds[wave] = __float_as_uint(m);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (not __syncthreads(): that would also sit out the output stores)
  if (threadIdx.x == 0) {
    const unsigned b01 = lds[0] > lds[1] ? lds[0] : lds[1], b23 = lds[2] > lds[3] ? lds[2] : lds[3];
    const unsigned b = b01 > b23 ? b01 : b23;                  // non-negative floats order like their bit patterns
#ifndef SHDR_ABL_NO_RANGE_ATOMIC
    if (b > seen) atomicMax(slot, b);
#endif
  }
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__host__ __device__ inline int f4(int row) { return (-(row >> 2)) & 3; }      // filter images: rows = couts, 16-row aligned fragments
// Patch images: the 16-byte slot of channel group kg of patch column c is kg ^ sx(c).  A ds_read_b128 is served in lane groups of
// {8 lanes of one kg, 8 lanes of kg ^ 1}; with the lane -> tile-column permutation pcol() below each set of 8 reads 8 CONSECUTIVE
// columns, so (column mod 8, kg) -- and with it (64-byte quarter, slot) -- is distinct for every tap shift: conflict-free at any
// alignment (the row-keyed swizzle of the filter images 2-way-conflicted on 23 % of the LDS cycles here), and the address splits
// into a per-lane column term (one register per kw) + a scalar row term: one VALU add per tap instead of six per operand read.
__host__ __device__ inline int sx(int col) { return ((col >> 2) & 1) * 2; }
__device__ __forceinline__ int pcol(int fi) { return fi < 4 ? fi : (fi >= 12 ? fi - 8 : fi + 4); }

// The input is multiplied by the power of two that brings its range bound (a.xr1 / a.xr2) to [2^10, 2^11) before the split and the result
// divided by it -- both exact.  This matters at both ends: activations beyond the fp16 range, and the output gradients dz of the
// training steps far below it (max |dz| 3e-8 ... 2e-2 in the joint step: unscaled, their high terms are fp16 subnormals and the
// parameter gradient differed from the exact-fp32 kernels' by 3e-5; scaled, by the run-to-run noise).
// MP (with KH = KW = 4, the largest phase): the four parity phases of a 7x7 / stride-2 layer in ONE launch -- a phase loop around the chunk loop
// with the tap counts, the input offset and the sub-filter of each phase taken from the argument arrays; the partial sums stay in the
// accumulators (the four-launch form wrote and re-read them three times: 3 x 2 x N Ho Wo Cout floats).  The patch geometry is that of the
// 4 x 4 phase for every phase: a 3-tap dimension loads one row / column it does not use.
template <bool UP, int KH, int KW, bool MP = false>
__global__ __launch_bounds__(256, 2) void conv_x3_kernel(const X3Args a) {
  using G = X3G<KH, KW>;
  constexpr int PWID = G::PWID, PPIX = G::PPIX, PJ = G::PJ, PATCH_HALVES = G::PATCH_HALVES, NTAPS = KH * KW;
  static_assert(!UP || (KH == 3 && KW == 3), "the up-sampling prologue belongs to the 3 x 3 stride-1 form");
  static_assert(!MP || (!UP && KH == 4 && KW == 4), "the phase loop is built on the 4 x 4 patch geometry");
  extern __shared__ __attribute__((aligned(16))) _Float16 xsm[];
  // LDS regions as expressions of the __shared__ symbol (pointer VARIABLES captured by the lambdas below lost their address space: the
  // compiler kept them as 64-bit generic pointers in scratch and reloaded them inside the tap loop)
#define patch_h (xsm)                                          /* [PATCH_HALVES] */
#define patch_l (xsm + PATCH_HALVES)
#define filt (xsm + 2 * PATCH_HALVES)                          /* [2][UNIT_HALVES] */
#define lrs (reinterpret_cast<float*>(xsm + 2 * PATCH_HALVES + 2 * UNIT_HALVES))       /* UP: [LRPIX][32] fp32 */

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

  // ---- patch geometry (fixed per block): piece p = tid + 256 j -> (patch pixel, float4 of the 32-channel chunk) ----------------
  int ppix[PJ];                                                // pixel index in the image tensor, -1: padding / beyond the patch
  int pdst[PJ];                                                // half offset of the 8-byte destination inside an image
  auto patch_geometry = [&](int bh, int bw) __attribute__((always_inline)) {                  // (MP: once per phase)
#pragma unroll
    for (int j = 0; j < (UP ? 0 : PJ); ++j) {
      const int p = tid + 256 * j;
      const int pix = p >> 3, q = p & 7;
      const int py = pix / PWID, px = pix - py * PWID;
      const int ih = a.in_s * (oh0 + py) + bh, iw = a.in_s * (ow0 + px) + bw;
      const bool ok = pix < PPIX && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      ppix[j] = ok ? (img * a.Hin + ih) * a.Win + iw : -1;
      pdst[j] = pix < PPIX ? pix * 32 + 8 * ((q >> 1) ^ sx(px)) + 4 * (q & 1) : -1;
    }
  };
  if (!MP) patch_geometry(a.bh, a.bw);
  // UP: low-res pieces of this thread: piece p = tid + 256 j -> (low-res patch pixel, float4)
  int lpix[LRJ];
  if (UP) {
#pragma unroll
    for (int j = 0; j < LRJ; ++j) {
      const int p = tid + 256 * j;
      const int pix = p >> 3;
      const int ly = pix / LRW, lx = pix - ly * LRW;
      const int r = (oh0 >> 1) - 1 + ly, c = (ow0 >> 1) - 1 + lx;
      lpix[j] = (pix < LRPIX && (unsigned)r < (unsigned)a.Hl && (unsigned)c < (unsigned)a.Wl) ? (img * a.Hl + r) * a.Wl + c : -1;
    }
  }
  const int nch1 = a.C1 >> 5, nch = (a.C1 + a.C2) >> 5;
  int nunits = nch * NTAPS;                                    // (MP: per phase)
  // 1 x 1 layers run ONE tap (48 MFMAs, ~0.4 us) per chunk: a patch load issued one chunk ahead still has most of its HBM latency in
  // front of it when the chunk is needed -- they prefetch TWO chunks ahead into two register sets (the chunk loop below alternates them)
  constexpr int DEPTH = (NTAPS == 1 && !UP && !MP) ? 2 : 1;
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, DEPTH - 1>;
  f32x4 pr_sets[DEPTH][PJ];
#define pr (pr_sets[0])                                        /* the UP paths and the single-set kernels */
  auto load_lr = [&](int c) __attribute__((always_inline)) {                                  // UP: low-res chunk c -> the first LRJ registers
#pragma unroll
    for (int j = 0; j < LRJ; ++j) {
      const int q = (tid + 256 * j) & 7;
      pr[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (lpix[j] >= 0) pr[j] = *reinterpret_cast<const f32x4*>(a.x1 + (size_t)(unsigned)lpix[j] * (unsigned)a.C1 + (c << 5) + 4 * q);
    }
  };
  auto park_lr = [&]() __attribute__((always_inline)) {                                       // UP: registers -> LDS scratch [pixel][32]
#pragma unroll
    for (int j = 0; j < LRJ; ++j)
      if (tid + 256 * j < LRPIX * 8) *reinterpret_cast<f32x4*>(lrs + 4 * (tid + 256 * j)) = pr[j];
  };
  float xs = 1.0f, ixs = 1.0f;                                 // input scale 2^T and its inverse: set BEHIND the first patch / filter loads below (the
                                                               // slot is a dependent scalar load: in front of them it delayed every block's first fetch)
  auto split_store = [&](int dst, const f32x4 v4) __attribute__((always_inline)) {            // one float4 -> 8 bytes in each fp16 image
    unsigned h[2], l[2];
    x3_split4(v4, xs, h, l);
    *reinterpret_cast<uint2*>(patch_h + dst) = make_uint2(h[0], h[1]);
    *reinterpret_cast<uint2*>(patch_l + dst) = make_uint2(l[0], l[1]);
  };
  auto expand_store = [&]() __attribute__((always_inline)) {   // UP: scratch -> up-sampled 18 x 18 patch (the arithmetic of resize2x_kernel) -> images
    // the geometry of a piece is recomputed per chunk (a few integer operations) instead of being held in 22 registers: the
    // expansion needs them for its four taps
    const int r0 = (oh0 >> 1) - 1, c0 = (ow0 >> 1) - 1;
    int t0 = tid;
    asm volatile("" : "+v"(t0));                               // computed HERE: hoisted out of the chunk loop the 11 geometries are 90 live registers (spills)
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      const int p = t0 + 256 * j;
      const int pix = p >> 3, q = p & 7;
      if (pix >= PPIX) continue;
      const int py = pix / PWID, px = pix - py * PWID;
      const int ih = oh0 - 1 + py, iw = ow0 - 1 + px;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if ((unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win) {
        const int mr = ih >> 1, mc = iw >> 1;
        const int ra = (ih & 1) ? mr : max(mr - 1, 0), rb = (ih & 1) ? min(mr + 1, a.Hl - 1) : mr;
        const int ca = (iw & 1) ? mc : max(mc - 1, 0), cb = (iw & 1) ? min(mc + 1, a.Wl - 1) : mc;
        const float* s00 = lrs + ((ra - r0) * LRW + (ca - c0)) * 32 + 4 * q;
        const int dx = (cb - ca) * 32, dy = (rb - ra) * LRW * 32;
        const float wx = (iw & 1) ? 0.25f : 0.75f, wy = (ih & 1) ? 0.25f : 0.75f;
        const f32x4 q00 = *reinterpret_cast<const f32x4*>(s00), q01 = *reinterpret_cast<const f32x4*>(s00 + dx);
        const f32x4 q10 = *reinterpret_cast<const f32x4*>(s00 + dy), q11 = *reinterpret_cast<const f32x4*>(s00 + dy + dx);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = q00[e] + (q01[e] - q00[e]) * wx;      // horizontal first, then vertical
          const float u = q10[e] + (q11[e] - q10[e]) * wx;
          v[e] = t + (u - t) * wy;
        }
      }
      split_store(pix * 32 + 8 * ((q >> 1) ^ sx(px)) + 4 * (q & 1), v);
      if (j & 1) __builtin_amdgcn_sched_barrier(0);            // two pieces (32 registers of taps) in flight, not eleven: no spills
    }
  };
  auto load_patch = [&](int c, auto setc) __attribute__((always_inline)) {                    // chunk c -> register set setc
    constexpr int S = decltype(setc)::value;
    if (UP) { load_lr(c); return; }
    const bool second = c >= nch1;
    const float* src = second ? a.x2 : a.x1;
    const int Cs = second ? a.C2 : a.C1;
    const int c0 = (second ? c - nch1 : c) << 5;
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      const int q = (tid + 256 * j) & 7;
      pr_sets[S][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ppix[j] >= 0) pr_sets[S][j] = *reinterpret_cast<const f32x4*>(src + (size_t)(unsigned)ppix[j] * (unsigned)Cs + c0 + 4 * q);
    }
  };
  auto store_patch = [&](auto setc) __attribute__((always_inline)) {                          // register set setc -> the two fp16 images
    constexpr int S = decltype(setc)::value;
    if (UP) { expand_store(); return; }
#pragma unroll
    for (int j = 0; j < PJ; ++j)
      if (pdst[j] >= 0) split_store(pdst[j], pr_sets[S][j]);
  };
  // ---- filter units: 8 KB per tap = 512 pieces of 16 bytes, two per thread --------------------------------------------------------
  constexpr int FJ = 2;
  const _Float16* wbase = a.wp + (size_t)pn * nunits * UNIT_HALVES;      // (MP: per phase)
  int fdst[FJ];
#pragma unroll
  for (int j = 0; j < FJ; ++j) {
    const int p = tid + 256 * j;
    const int im = p >> 8, co = (p & 255) >> 2, slot = p & 3;
    fdst[j] = im * IMG_HALVES + co * 32 + 8 * (slot ^ f4(co));
  }
  u32x4 fr[FJ];
  auto load_filt = [&](int u) __attribute__((always_inline)) {
    const u32x4* g = reinterpret_cast<const u32x4*>(wbase + (size_t)u * UNIT_HALVES);
#pragma unroll
    for (int j = 0; j < FJ; ++j) fr[j] = g[tid + 256 * j];
  };
  auto store_filt = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < FJ; ++j) *reinterpret_cast<u32x4*>(filt + buf * UNIT_HALVES + fdst[j]) = fr[j];
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fg = lane >> 4;
  int acol[KW], b_rd[NT];                                      // A operand: half offset of (column pcol + kw, channel group fg) inside a patch row
  const int pc = pcol(fi);
#pragma unroll
  for (int kw = 0; kw < KW; ++kw) acol[kw] = (pc + kw) * 32 + 8 * (fg ^ sx(pc + kw));
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    const int row = ni * 16 + fi;
    b_rd[ni] = row * 32 + 8 * (fg ^ f4(row));
  }

  int kh_n = KH, kw_n = KW;                                    // taps of the phase (MP)
#pragma unroll 1
  for (int phase = 0; phase < (MP ? a.nphase : 1); ++phase) {
  if (MP) {
    kh_n = a.pth[phase];
    kw_n = a.ptw[phase];
    nunits = nch * kh_n * kw_n;
    wbase = a.pwp[phase] + (size_t)pn * nunits * UNIT_HALVES;
    patch_geometry(a.pbh[phase], a.pbw[phase]);
    if (phase > 0) __syncthreads();                            // every wave has read the last tap of the previous phase (patch and filter buffers)
  }
  load_patch(0, Set0{});
  if (DEPTH == 2 && nch > 1) load_patch(1, Set1{});
  load_filt(0);
  if (!MP || phase == 0) x3_range_scale(a.xr1, a.xr2, xs, ixs);
  if (UP) {
    park_lr();
    __syncthreads();
  }
  store_patch(Set0{});
  store_filt(0);
  // one chunk: `cur` = the register set chunk c was loaded into (already split into LDS: free for chunk c + DEPTH), `nxt` = the set of chunk c + 1
  auto chunk = [&](int c, auto cur, auto nxt) __attribute__((always_inline)) {
    if (c + DEPTH < nch) load_patch(c + DEPTH, cur);           // lands under the taps of this chunk (and, for the 1 x 1 layers, of the next)
#pragma unroll 1
    for (int kh = 0; kh < (MP ? kh_n : KH); ++kh) {
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {                        // unrolled: acol[kw] stays a register
        if (MP && kw >= kw_n) continue;                        // (block-uniform)
        const int u = MP ? (c * kh_n + kh) * kw_n + kw : c * NTAPS + kh * KW + kw;
        __syncthreads();                                       // unit u (and, at the first tap, the patch) is in LDS; buffer (u + 1) & 1 is free
        if (u + 1 < nunits) load_filt(u + 1);
        const _Float16* F = filt + (u & 1) * UNIT_HALVES;
        const int rowh = (wave * MT + kh) * PWID * 32;          // scalar: first patch row of this wave and tap
        // A-operand reads run one pixel row ahead of the MFMAs that use them
        f16x8 wh[NT], ws[NT], wl[NT], ph[2], pl[2];
        auto read_a = [&](int mi, int slot) __attribute__((always_inline)) {
          const int o = rowh + mi * PWID * 32 + acol[kw];
          ph[slot] = *reinterpret_cast<const f16x8*>(patch_h + o);
          pl[slot] = *reinterpret_cast<const f16x8*>(patch_l + o);
        };
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          wh[ni] = *reinterpret_cast<const f16x8*>(F + b_rd[ni]);
          wl[ni] = *reinterpret_cast<const f16x8*>(F + IMG_HALVES + b_rd[ni]);
        }
        read_a(0, 0);
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) ws[ni] = wh[ni] * (_Float16)(1.0f / 2048.0f);      // exact (power of two; gradual underflow below |wh| = 2^-3 as in fp16 itself)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          if (mi + 1 < MT) read_a(mi + 1, (mi + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], ph[mi & 1], acc[mi][ni], 0, 0, 0);
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ws[ni], pl[mi & 1], acc[mi][ni], 0, 0, 0);
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ni], ph[mi & 1], acc[mi][ni], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (u + 1 < nunits) store_filt((u + 1) & 1);           // read after the next barrier
      }
    }
    if (c + 1 < nch) {
      if (UP) {
        park_lr();                                             // the scratch was last read before the tap loop of this chunk
        __syncthreads();                                       // ... and every wave has read the last tap of this chunk's patch
      } else {
        __syncthreads();                                       // every wave has read the last tap of this chunk's patch
      }
      store_patch(nxt);                                        // visible after the barrier at the top of the next tap loop
    }
  };
#pragma unroll 1
  for (int c = 0; c < nch; c += DEPTH) {
    chunk(c, Set0{}, Set1{});
    if (DEPTH == 2 && c + 1 < nch) chunk(c + 1, Set1{}, Set0{});
  }
  }                                                            // phase loop
#undef pr

  // ---- epilogue: y = act2(affine(act1(acc * 2^-S + bias))), 16-byte stores (lane = pixel x 4 consecutive couts); the 2 x 2 pooling
  //      window of the optional second output is two rows of this lane and of its neighbour lane -----------------------------------
  const float inv_s = a.hdr[1] * ixs;
  const int ow = ow0 + pc;                                     // the lane's tile column (pcol permutation)
  // bias / scale / shift of the lane's four cout quads are loaded ONCE, in front of the store loop: a load inside it is followed by
  // "s_waitcnt vmcnt(0)", which also sits out the round trip of the output store issued just before it -- the stores of a block went
  // out one at a time
  // (the offsets pass through an opaque asm so that the loads are not hoisted above the tap loop, where their registers would be live
  //  for the whole kernel)
  const unsigned yr_seen = (a.yr && a.final) ? __hip_atomic_load(a.yr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  int ep0 = n0;
  asm volatile("" : "+s"(ep0));
  f32x4 bias_r[NT], scale_r[NT], shift_r[NT];
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    const int cl = ni * 16 + 4 * fg;
    const bool cv = ep0 + cl < a.Cout;                           // (false only in the zero half of a 32-cout layer's slice: nothing loaded, nothing stored)
    bias_r[ni] = (a.bias && a.final && cv) ? *reinterpret_cast<const f32x4*>(a.bias + ep0 + cl) : (f32x4){0.f, 0.f, 0.f, 0.f};
    scale_r[ni] = (a.scale && a.final && cv) ? *reinterpret_cast<const f32x4*>(a.scale + ep0 + cl) : (f32x4){1.f, 1.f, 1.f, 1.f};
    shift_r[ni] = (a.scale && a.final && cv) ? *reinterpret_cast<const f32x4*>(a.shift + ep0 + cl) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // the partial sums of the earlier phases of a stride-2 layer (yin), batched in front of the stores for the same reason
  // (the residual of a final launch travels in the same registers: a layer has partial sums OR a residual)
  f32x4 yin_r[MT / 2][NT][2];
  const float* addend = a.yin ? a.yin : (a.final ? a.res : nullptr);
  if (addend) {
    const int add_cs = a.yin ? a.Cout : a.res_cs;
#pragma unroll
    for (int mp = 0; mp < MT / 2; ++mp)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int oh = oh0 + wave * MT + 2 * mp + r;
          yin_r[mp][ni][r] = (oh < a.H && ow < a.W && ep0 + ni * 16 + 4 * fg < a.Cout)
                                 ? *reinterpret_cast<const f32x4*>(addend + ((size_t)(img * a.H + oh) * a.W + ow) * add_cs + ep0 + ni * 16 + 4 * fg)
                                 : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
  }
  float ym = 0.0f;                                             // max |y| over this lane's stored values (a.yr)
#pragma unroll
  for (int mp = 0; mp < MT / 2; ++mp) {
    const int oh = oh0 + wave * MT + 2 * mp;                   // even row of the pair (H even whenever yp is given)
    if (oh >= a.H) continue;                                   // wave-uniform
    float pj[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};      // a.proj: this lane's share (16 of the 64 couts) of the projected pixel, rows oh, oh + 1
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int cl = ni * 16 + 4 * fg;
      f32x4 v[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        v[r] = acc[2 * mp + r][ni] * inv_s;
        const bool inside = oh + r < a.H && ow < a.W;
        if (a.yin) v[r] += yin_r[mp][ni][r];
        if (!a.final) {
          if (inside) *reinterpret_cast<f32x4*>(a.y + ((size_t)(img * a.H + oh + r) * a.W + ow) * a.Cout + n0 + cl) = v[r];
          continue;
        }
        v[r] += bias_r[ni];
        shdr::act_apply4<0>(v[r], a.act1);
        if (a.scale) v[r] = v[r] * scale_r[ni] + shift_r[ni];
        if (a.res) v[r] += yin_r[mp][ni][r];
        shdr::act_apply4<0>(v[r], a.act2);
        if (a.y && oh + r < a.H && ow < a.W && n0 + cl < a.Cout)
          *reinterpret_cast<f32x4*>(a.y + ((size_t)(img * a.H + oh + r) * a.W + ow) * a.Cout + n0 + cl) = v[r];
        if (a.proj) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(a.proj + j * 64 + cl);
            pj[r][j] += v[r][0] * q[0] + v[r][1] * q[1] + v[r][2] * q[2] + v[r][3] * q[3];
          }
        }
#ifndef SHDR_ABL_NO_YM
        if (a.yr && oh + r < a.H) ym = fmaxf(fmaxf(fmaxf(fmaxf(ym, fabsf(v[r][0])), fabsf(v[r][1])), fabsf(v[r][2])), fabsf(v[r][3]));      // two v_max3_f32
#endif
      }
      if (a.yp && a.final) {
        f32x4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (a.pool_avg) {                                    // (top-left + top-right) + (bottom-left + bottom-right), as pool.hip adds them
            const float t = v[0][e] + __shfl_xor(v[0][e], 1, 64), b = v[1][e] + __shfl_xor(v[1][e], 1, 64);
            m[e] = 0.25f * (t + b);
          } else {
            m[e] = fmaxf(v[0][e], v[1][e]);
            m[e] = fmaxf(m[e], __shfl_xor(m[e], 1, 64));
          }
        }
        if (!(pc & 1) && ow < a.W && n0 + cl < a.Cout)           // lanes fi and fi ^ 1 hold columns pc and pc ^ 1
          *reinterpret_cast<f32x4*>(a.yp + ((size_t)(img * (a.H >> 1) + (oh >> 1)) * (a.W >> 1) + (ow >> 1)) * a.Cout + n0 + cl) = m;
      }
    }
    if (a.proj && a.final) {                                   // the four lane groups fg hold 16 couts each of the same pixel
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float t = pj[r][j];
          t += __shfl_xor(t, 16, 64);
          t += __shfl_xor(t, 32, 64);
          pj[r][j] = t;
        }
        if (fg == 0 && oh + r < a.H && ow < a.W) {
          float* o = a.yproj + ((size_t)(img * a.H + oh + r) * a.W + ow) * 3;
          o[0] = pj[r][0]; o[1] = pj[r][1]; o[2] = pj[r][2];
        }
      }
    }
  }
#ifndef SHDR_ABL_NO_TAIL
  if (a.yr && a.final) x3_range_out(a.yr, ow < a.W ? ym : 0.0f, lane, wave, yr_seen, reinterpret_cast<unsigned*>(xsm));      // (a pooled output is bounded by the same maximum)
#endif
}

#undef patch_h
#undef patch_l
#undef filt
#undef lrs

// ---- filter preparation ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void x3_absmax_kernel(const float* __restrict__ w, long n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));          // non-negative floats order like their bit patterns
}
// max |x| of a large tensor into *out (block-level reduction, one atomicMax per block)
__global__ __launch_bounds__(256) void x3_absmax_big_kernel(const float* __restrict__ x, long n4, unsigned* __restrict__ out) {
  __shared__ float part[4];
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}
// packed[nb][u = chunk * taps + tap][image][co][k]: w * x2-scale * 2^S split into wh, wl
// taps of the packed filter = the sub-filter w[p0 + step * a][q0 + step * b], a < TH, b < TW, of a KWF-wide filter
__global__ __launch_bounds__(256) void x3_pack_kernel(const float* __restrict__ w, float* __restrict__ hdr, _Float16* __restrict__ out, int Ct,
                                                      int C1, int Cout, float x2_scale, int KWF, int TH, int TW, int p0, int q0, int step) {
  const float mx = fmaxf(__uint_as_float(reinterpret_cast<const unsigned*>(hdr)[0]) * fmaxf(1.0f, fabsf(x2_scale)), 1e-30f);
  int ex;
  frexpf(mx, &ex);                                             // mx = f * 2^ex, f in [0.5, 1)
  int S = 14 - ex;                                             // max |w'| in [2^13, 2^14)
  S = S < -100 ? -100 : (S > 100 ? 100 : S);
  const float s = ldexpf(1.0f, S);
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[1] = ldexpf(1.0f, -S);
  const int ntaps = TH * TW;
  const int nunits = (Ct >> 5) * ntaps;
  const long total = (long)((Cout + 63) / 64) * nunits * 64 * 32;      // (a 32-cout layer fills half of its one 64-cout slice with zeros)
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int k = (int)(e & 31), co = (int)((e >> 5) & 63);
    const long t = e >> 11;
    const int u = (int)(t % nunits), nb = (int)(t / nunits);
    const int chunk = u / ntaps, t2 = u - ntaps * chunk;
    const int ta = t2 / TW, tb = t2 - TW * ta;
    const int tap = (p0 + step * ta) * KWF + (q0 + step * tb);
    const int ch = chunk * 32 + k;
    float v = nb * 64 + co < Cout ? w[((size_t)tap * Ct + ch) * Cout + nb * 64 + co] * s : 0.0f;
    if (ch >= C1) v *= x2_scale;
    const _Float16 h = (_Float16)v;
    _Float16* o = out + ((size_t)(nb * nunits + u) * 2) * IMG_HALVES + co * 32 + k;
    o[0] = h;
    o[IMG_HALVES] = (_Float16)(v - (float)h);
  }
}

}  // namespace

namespace {

struct X3Phase { int th, tw, p0, q0, step, bh, bw; };
// 3 x 3 / stride 1: one "phase" (all nine taps).  7 x 7 / stride 2 (linearization_net.py:91): input pixels of one row / column parity
// meet the filter taps of one parity only, so the layer is the sum of four stride-1 correlations of the parity-subsampled input with
// the 4 x 4, 4 x 3, 3 x 4 and 3 x 3 sub-filters -- exactly the 49 taps, each phase one launch accumulating into y.
int x3_phases(const shdr_conv2d_desc* d, X3Phase ph[4]) {
  if (d->stride == 1 || d->KH == 1) {
    // 3 x 3 (pad 1) or 1 x 1 (pad 0): all taps in one launch; the 1 x 1 / stride-2 layer reads input pixel (2 oh, 2 ow)
    ph[0] = X3Phase{d->KH, d->KW, 0, 0, d->stride, -d->pad_t, -d->pad_l};
    return 1;
  }
  int n = 0;
  for (int p0 = 0; p0 < 2; ++p0)
    for (int q0 = 0; q0 < 2; ++q0) ph[n++] = X3Phase{(d->KH - p0 + 1) / 2, (d->KW - q0 + 1) / 2, p0, q0, 2, p0 - d->pad_t, q0 - d->pad_l};
  return n;
}
inline int64_t x3_phase_floats(const X3Phase& p, int Ct, int Cout) { return X3_HEADER_FLOATS + (int64_t)p.th * p.tw * Ct * ((Cout + 63) / 64 * 64); }    // two fp16 images

template <bool UP, int KH, int KW, bool MP = false>
int launch_x3(const X3Args& a, hipStream_t st) {
  constexpr int lds = UP ? X3G<KH, KW>::LDS_BYTES_UP : X3G<KH, KW>::LDS_BYTES;
  static bool attr_done[shdr::kMaxDevices] = {};
  const int dev_slot = shdr::device_slot();
  if (!attr_done[dev_slot]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3_kernel<UP, KH, KW, MP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done[dev_slot] = true;
  }
  const long nblk = (long)a.nblk_m * a.nblk_n;
  if (nblk > 0x7fffffffL) return shdr::fail(SHDR_E_SHAPE, "conv2d_x3: grid of %ld blocks", nblk);
  hipLaunchKernelGGL((conv_x3_kernel<UP, KH, KW, MP>), dim3((unsigned)nblk), dim3(256), lds, st, a);
  return shdr::check_launch("conv_x3_kernel");
}

}  // namespace

extern "C" int shdr_conv2d_x3_ok_f32(const shdr_conv2d_desc* d) {
  // Cout 32 (the 64 -> 32 and 32 + 32 -> 32 decoder layers of the U-Nets, dequantization_net.py:17-29): one 64-cout slice, half of it zero
  // filter columns that are neither biased nor stored -- twice the arithmetic of the layer and still 1.3x the exact fp32 kernel
  if (!d || d->C1 <= 0 || d->C1 % 32 || d->C2 % 32 || (d->Cout % 64 && (d->Cout != 32 || d->stride != 1 || SHDR_ENV("SHDR_NO_X3_COUT32")))) return 0;
  const int cv = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  if (cv != d->Cout || d->w_batch_stride != 0 || d->y_pix_stride > 1) return 0;
  if ((long)d->N * d->H * d->W * (d->C1 > d->C2 ? d->C1 : d->C2) >= (1L << 31)) return 0;
  if (SHDR_ENV("SHDR_NO_X3")) return 0;
  if (d->act1 == SHDR_ACT_TANH || d->act2 == SHDR_ACT_TANH) return 0;      // no wide layer of the networks has one: tanhf is not compiled in (act_apply4)
  if (d->stride == 1) {
    const bool k3 = d->KH == 3 && d->KW == 3 && d->pad_t == 1 && d->pad_l == 1;
    // 1 x 1 layers (the skip layers of hallucination_net.py:93-107 on tf.concat of two sources, the bottleneck convs of the ResNet
    // blocks): one tap per chunk, so the patch split is not amortised over nine taps -- still 2-3x the fp32-MFMA kernel from K = 256 on
    int k1_min = 64;                                           // (measured, tools/one_1x1.py: 64 -> 256 at 16 x 128^2 0.171 -> 0.129 ms, 128 -> 512 at 64^2 0.119 -> 0.073)
    if (const char* e = SHDR_ENV("SHDR_X3_1X1_MIN_K")) k1_min = atoi(e);
    const bool k1 = d->KH == 1 && d->KW == 1 && d->pad_t == 0 && d->pad_l == 0 && d->C1 + d->C2 >= k1_min && SHDR_ENV("SHDR_NO_X3_1X1") == nullptr;
    if (!(k3 || k1) || d->Ho != d->H || d->Wo != d->W) return 0;
  } else if (d->stride == 2 && d->KH == 1 && d->KW == 1) {
    // the 1 x 1 / stride-2 projections of the ResNet blocks (linearization_net.py:6-48, res4): the 1 x 1 kernel on every other input pixel
    if (d->pad_t != 0 || d->pad_l != 0 || d->C2 != 0 || d->prologue != SHDR_PROLOGUE_NONE || d->Ho != (d->H + 1) / 2 || d->Wo != (d->W + 1) / 2 ||
        d->C1 < 64 || SHDR_ENV("SHDR_NO_X3_1X1") || SHDR_ENV("SHDR_NO_X3_STRIDE2"))
      return 0;
  } else {
    // the 7 x 7 / stride-2 stem with TF SAME padding (one source, no prologue)
    int ho = 0, wo = 0, pt = 0, pl = 0;
    shdr_same_pad(d->H, 7, 2, &ho, &pt);
    shdr_same_pad(d->W, 7, 2, &wo, &pl);
    if (d->stride != 2 || d->KH != 7 || d->KW != 7 || d->C2 != 0 || d->prologue != SHDR_PROLOGUE_NONE || d->Ho != ho || d->Wo != wo || d->pad_t != pt ||
        d->pad_l != pl || SHDR_ENV("SHDR_NO_X3_STRIDE2"))
      return 0;
  }
  // enough blocks to fill the chip: the deepest, smallest maps stay on the fused Winograd kernel (8 x 16 tiles)
  const long blocks = (long)d->N * ((d->Ho + 15) / 16) * ((d->Wo + 15) / 16) * ((d->Cout + 63) / 64);
  long min_blocks = 192;          // measured (tools/dbg/x3_threshold.py): 256 blocks 1.24-1.28x the fused Winograd kernel, 128 blocks 0.75x
  if (const char* e = SHDR_ENV("SHDR_X3_MIN_BLOCKS")) min_blocks = atol(e);
  return blocks >= min_blocks ? 1 : 0;
}

extern "C" int64_t shdr_conv2d_x3_filter_elems_f32(const shdr_conv2d_desc* d) {
  if (!d || d->C1 <= 0 || (d->C1 + d->C2) % 32 || (d->Cout % 64 && d->Cout != 32) || (d->stride != 1 && d->stride != 2)) return -1;
  X3Phase ph[4];
  const int n = x3_phases(d, ph);
  int64_t total = 0;
  for (int i = 0; i < n; ++i) total += x3_phase_floats(ph[i], d->C1 + d->C2, d->Cout);
  return total;
}

// premax: header slot 0 of the (single) phase already holds max |w| (written by the kernel that produced w: the input gradient's filter
// transform) -- no memset, no absmax launch
extern "C" int shdr_conv2d_x3_prepare_filter_premax_f32(const shdr_conv2d_desc* d, const float* w, float* prepared, int premax, void* stream);
extern "C" int shdr_conv2d_x3_prepare_filter_f32(const shdr_conv2d_desc* d, const float* w, float* prepared, void* stream) {
  return shdr_conv2d_x3_prepare_filter_premax_f32(d, w, prepared, 0, stream);
}
extern "C" int shdr_conv2d_x3_prepare_filter_premax_f32(const shdr_conv2d_desc* d, const float* w, float* prepared, int premax, void* stream) {
  SHDR_REQUIRE(d && w && prepared, SHDR_E_NULL, "conv2d_x3_prepare_filter: null pointer");
  const int Ct = d->C1 + d->C2;
  SHDR_REQUIRE(Ct > 0 && Ct % 32 == 0 && d->C1 % 32 == 0 && d->Cout > 0 && (d->Cout % 64 == 0 || d->Cout == 32), SHDR_E_SHAPE,
               "conv2d_x3_prepare_filter: need C %% 32 == 0, Cout %% 64 == 0 (or Cout 32)");
  SHDR_REQUIRE(shdr::aligned16(prepared), SHDR_E_ALIGN, "conv2d_x3_prepare_filter: prepared must be 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float x2s = d->C2 > 0 ? d->x2_scale : 1.0f;
  X3Phase ph[4];
  const int n = x3_phases(d, ph);
  const long nw = (long)d->KH * d->KW * Ct * d->Cout;
  float* out = prepared;
  SHDR_REQUIRE(!premax || n == 1, SHDR_E_SHAPE, "conv2d_x3_prepare_filter: premax is for single-phase (stride-1) layers");
  for (int i = 0; i < n; ++i) {
    if (!premax) {
      if (hipMemsetAsync(out, 0, X3_HEADER_FLOATS * sizeof(float), st) != hipSuccess) return shdr::fail(SHDR_E_LAUNCH, "conv2d_x3_prepare_filter: memset");
      // <= 64 blocks: every wave ends with an atomicMax on ONE address (2048 blocks took 50 us on a 9 MB filter, this takes 6)
      const int gmax = shdr::stream_grid(nw) < 64 ? shdr::stream_grid(nw) : 64;
      hipLaunchKernelGGL(x3_absmax_kernel, dim3(gmax), dim3(256), 0, st, w, nw, reinterpret_cast<unsigned*>(out));
    }
    const long np = (long)ph[i].th * ph[i].tw * Ct * ((d->Cout + 63) / 64 * 64);
    hipLaunchKernelGGL(x3_pack_kernel, dim3(shdr::stream_grid(np)), dim3(256), 0, st, w, out, reinterpret_cast<_Float16*>(out + X3_HEADER_FLOATS), Ct,
                       d->C1, d->Cout, x2s, d->KW, ph[i].th, ph[i].tw, ph[i].p0, ph[i].q0, ph[i].step);
    out += x3_phase_floats(ph[i], Ct, d->Cout);
  }
  return shdr::check_launch("conv2d_x3_prepare_filter");
}

// max |x| of a tensor, atomicMax-ed into a RANGE SLOT (a device word holding the bit pattern of a non-negative float; the caller zeroes
// it, e.g. one memset over a slab of slots per step).  The slot feeds x1_range / x2_range of the ranged convolution calls.
extern "C" int shdr_absmax_f32(const float* x, int64_t n, float* range, void* stream) {
  SHDR_REQUIRE(x && range, SHDR_E_NULL, "absmax: null pointer");
  SHDR_REQUIRE(n > 0, SHDR_E_SHAPE, "absmax: n must be positive");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned* slot = reinterpret_cast<unsigned*>(range);
  const int64_t n4 = shdr::aligned16(x) ? n / 4 : 0;
  if (n4 > 0) {
    int grid = shdr::stream_grid(n4);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(x3_absmax_big_kernel, dim3(grid), dim3(256), 0, st, x, (long)n4, slot);
  }
  if (n - 4 * n4 > 0) hipLaunchKernelGGL(x3_absmax_kernel, dim3(n4 > 0 ? 1 : (shdr::stream_grid(n) < 64 ? shdr::stream_grid(n) : 64)), dim3(256), 0, st, x + 4 * n4, (long)(n - 4 * n4), slot);
  return shdr::check_launch("absmax");
}

// the range slot inside a prepared filter's header (slot 2): the low-level protocol of desc.prologue = SHDR_PROLOGUE_RANGE_SCALE
extern "C" int shdr_conv2d_x3_input_absmax_f32(const float* x, int64_t n, float* prepared, void* stream) {
  SHDR_REQUIRE(x && prepared, SHDR_E_NULL, "conv2d_x3_input_absmax: null pointer");
  SHDR_REQUIRE(n > 0 && n % 4 == 0 && shdr::aligned16(x), SHDR_E_SHAPE, "conv2d_x3_input_absmax: n must be a positive multiple of 4, x 16-byte aligned");
  return shdr_absmax_f32(x, n, prepared + 2, stream);
}

static int x3_forward(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                      const float* scale, const float* shift, const float* residual, float* y, float* y_pool, const float* proj, float* y_proj,
                      const float* x1_range, const float* x2_range, float* y_range, void* stream) {
  SHDR_REQUIRE(d && x1 && prepared && (y || y_pool || y_proj), SHDR_E_NULL, "conv2d_x3: null desc/x1/filter or no output");
  SHDR_REQUIRE((proj == nullptr) == (y_proj == nullptr), SHDR_E_NULL, "conv2d_x3: proj and y_proj come together");
  SHDR_REQUIRE(!proj || (d->Cout == 64 && d->stride == 1 && shdr::aligned16(proj)), SHDR_E_SHAPE,
               "conv2d_x3: the projected output takes a stride-1 layer with 64 output channels (one cout slice per block) and a 16-byte aligned map");
  SHDR_REQUIRE(!y_pool || (d->Ho % 2 == 0 && d->Wo % 2 == 0 && shdr::aligned16(y_pool)), SHDR_E_SHAPE, "conv2d_x3: the fused 2x2 max-pool needs even Ho, Wo");
  const bool up = d->prologue == SHDR_PROLOGUE_BILINEAR2X, rs = d->prologue == SHDR_PROLOGUE_RANGE_SCALE;
  SHDR_REQUIRE(d->prologue == SHDR_PROLOGUE_NONE || rs || (up && d->stride == 1 && d->C2 == 0 && d->H % 2 == 0 && d->W % 2 == 0), SHDR_E_SHAPE,
               "conv2d_x3: the bilinear 2x prologue takes a stride-1 layer, one source and even (up-sampled) H, W");
  SHDR_REQUIRE(shdr_conv2d_x3_ok_f32(d), SHDR_E_SHAPE, "conv2d_x3: layer shape not taken by this kernel");
  SHDR_REQUIRE((d->C2 == 0) == (x2 == nullptr), SHDR_E_NULL, "conv2d_x3: x2 must be given iff C2 > 0");
  SHDR_REQUIRE((scale == nullptr) == (shift == nullptr), SHDR_E_NULL, "conv2d_x3: scale and shift come together");
  SHDR_REQUIRE(shdr::aligned16(x1) && (!x2 || shdr::aligned16(x2)) && shdr::aligned16(prepared) && (!y || shdr::aligned16(y)) &&
                   (!bias || shdr::aligned16(bias)) && (!scale || (shdr::aligned16(scale) && shdr::aligned16(shift))),
               SHDR_E_ALIGN, "conv2d_x3: tensors must be 16-byte aligned");
  SHDR_REQUIRE(!x2 || ((x1_range == nullptr) == (x2_range == nullptr)), SHDR_E_NULL, "conv2d_x3: give the range of both sources or of neither");
  X3Args a{};
  a.x1 = x1; a.x2 = x2 ? x2 : x1;
  a.bias = bias; a.scale = scale; a.shift = shift; a.y = y; a.yp = y_pool;
  a.pool_avg = d->pool == SHDR_POOL_AVG;
  a.N = d->N; a.H = d->Ho; a.W = d->Wo; a.C1 = d->C1; a.C2 = d->C2; a.Cout = d->Cout;
  a.Hin = d->H; a.Win = d->W;
  a.Hl = d->H / 2; a.Wl = d->W / 2;
  a.tiles_x = (a.W + 15) / 16;
  a.tiles_y = (a.H + 15) / 16;
  a.nblk_m = a.N * a.tiles_x * a.tiles_y;
  a.nblk_n = (a.Cout + 63) / 64;
  a.act1 = d->act1; a.act2 = d->act2;
  a.xr1 = reinterpret_cast<const unsigned*>(x1_range);
  a.xr2 = reinterpret_cast<const unsigned*>(x2 ? x2_range : nullptr);
  a.yr = reinterpret_cast<unsigned*>(y_range);
  a.proj = proj; a.yproj = y_proj;
  SHDR_REQUIRE(!residual || (d->stride == 1 && d->res_cstride >= d->Cout && d->res_cstride % 4 == 0 && shdr::aligned16(residual) && !y_pool && !proj), SHDR_E_SHAPE,
               "conv2d_x3: the residual takes a stride-1 layer, res_cstride >= Cout (a multiple of 4), a 16-byte aligned tensor, no pooled / projected output");
  a.res = residual; a.res_cs = d->res_cstride;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  X3Phase ph[4];
  const int n = x3_phases(d, ph);
  SHDR_REQUIRE(n == 1 || y, SHDR_E_NULL, "conv2d_x3: the phases of a stride-2 layer accumulate in y");
  const int Ct = d->C1 + d->C2;
  const float* pk = prepared;
  if (n == 4 && SHDR_ENV("SHDR_X3_STEM_PHASE_LAUNCHES") == nullptr) {
    // the stride-2 stem in ONE launch: the phases' packed sub-filters share the scale 2^S (each header's maximum is taken over the whole
    // filter), the partial sums stay in registers
    a.hdr = pk;
    a.in_s = ph[0].step;
    a.nphase = 4;
    for (int i = 0; i < 4; ++i) {
      SHDR_REQUIRE(ph[i].th <= 4 && ph[i].tw <= 4 && ph[i].step == ph[0].step, SHDR_E_SHAPE, "conv2d_x3: unexpected phase geometry");
      a.pth[i] = ph[i].th; a.ptw[i] = ph[i].tw; a.pbh[i] = ph[i].bh; a.pbw[i] = ph[i].bw;
      a.pwp[i] = reinterpret_cast<const _Float16*>(pk + X3_HEADER_FLOATS);
      pk += x3_phase_floats(ph[i], Ct, d->Cout);
    }
    a.wp = a.pwp[0];
    a.yin = nullptr;
    a.final = 1;
    if (rs && !x1_range) a.xr1 = reinterpret_cast<const unsigned*>(prepared) + 2;
    return launch_x3<false, 4, 4, true>(a, st);
  }
  for (int i = 0; i < n; ++i) {
    a.hdr = pk;
    a.wp = reinterpret_cast<const _Float16*>(pk + X3_HEADER_FLOATS);
    if (rs && !x1_range) a.xr1 = reinterpret_cast<const unsigned*>(pk) + 2;      // the header-slot protocol (shdr_conv2d_x3_input_absmax_f32)
    a.in_s = ph[i].step; a.bh = ph[i].bh; a.bw = ph[i].bw;
    a.yin = i > 0 ? y : nullptr;
    a.final = i == n - 1;
    int rc;
    if (ph[i].th == 3 && ph[i].tw == 3) rc = up ? launch_x3<true, 3, 3>(a, st) : launch_x3<false, 3, 3>(a, st);
    else if (ph[i].th == 1 && ph[i].tw == 1) rc = launch_x3<false, 1, 1>(a, st);
    else if (ph[i].th == 4 && ph[i].tw == 4) rc = launch_x3<false, 4, 4>(a, st);
    else if (ph[i].th == 4 && ph[i].tw == 3) rc = launch_x3<false, 4, 3>(a, st);
    else if (ph[i].th == 3 && ph[i].tw == 4) rc = launch_x3<false, 3, 4>(a, st);
    else rc = shdr::fail(SHDR_E_SHAPE, "conv2d_x3: no kernel for a %d x %d phase", ph[i].th, ph[i].tw);
    if (rc) return rc;
    pk += x3_phase_floats(ph[i], Ct, d->Cout);
  }
  return SHDR_OK;
}

extern "C" int shdr_conv2d_fwd_x3_ranged_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                                             const float* scale, const float* shift, float* y, float* y_pool, const float* x1_range,
                                             const float* x2_range, float* y_range, void* stream) {
  SHDR_REQUIRE(y || y_pool, SHDR_E_NULL, "conv2d_x3: neither y nor y_pool");
  return x3_forward(d, x1, x2, prepared, bias, scale, shift, nullptr, y, y_pool, nullptr, nullptr, x1_range, x2_range, y_range, stream);
}

// The same with a residual: y = act2(affine(act1(conv + bias)) + residual), residual [N,Ho,Wo,res_cstride] -- the joins of the ResNet
// blocks (linearization_net.py:6-48: relu(norm(conv) + shortcut)) on the 1x1 layers this kernel takes
extern "C" int shdr_conv2d_fwd_x3_residual_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                                               const float* scale, const float* shift, const float* residual, float* y, const float* x1_range,
                                               const float* x2_range, float* y_range, void* stream) {
  SHDR_REQUIRE(y, SHDR_E_NULL, "conv2d_x3: null y");
  return x3_forward(d, x1, x2, prepared, bias, scale, shift, residual, y, nullptr, nullptr, nullptr, x1_range, x2_range, y_range, stream);
}

// The same launch with a PROJECTED output: y_proj[n,h,w,j] = sum_c proj[j][c] * y[n,h,w,c] (proj: [3][64] floats, j < 3), written from
// the epilogue that holds y in registers; y itself (and / or its pooled copy) is written only if asked for.  The tail of the
// Hallucination-Net (hallucination_net.py:179-185: the 1x1 skip layer s1 on concat[u1, d1 / 255] followed by the 1x1 conv2 -- two linear
// maps in a row) needs nothing but such a projection of u1's and of d1's 64 channels: their full-resolution tensors (1 GB each at
// 16 x 512^2) are then never written nor read back.
extern "C" int shdr_conv2d_fwd_x3_projected_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared,
                                                const float* bias, const float* scale, const float* shift, const float* proj, float* y_proj,
                                                float* y, float* y_pool, const float* x1_range, const float* x2_range, float* y_range,
                                                void* stream) {
  SHDR_REQUIRE(proj && y_proj, SHDR_E_NULL, "conv2d_x3_projected: null proj / y_proj");
  return x3_forward(d, x1, x2, prepared, bias, scale, shift, nullptr, y, y_pool, proj, y_proj, x1_range, x2_range, y_range, stream);
}

// The low-level entry point without range slots: the input is split as it stands (|x| must stay inside the fp16 range) unless
// desc.prologue = SHDR_PROLOGUE_RANGE_SCALE names the header slot written by shdr_conv2d_x3_input_absmax_f32.  Hosts go through
// shdr_conv2d_fwd_prepared_f32 / _ranged_f32, which never run a split-operand launch without a range.
extern "C" int shdr_conv2d_fwd_x3_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                                      const float* scale, const float* shift, float* y, float* y_pool, void* stream) {
  return shdr_conv2d_fwd_x3_ranged_f32(d, x1, x2, prepared, bias, scale, shift, y, y_pool, nullptr, nullptr, nullptr, stream);
}
