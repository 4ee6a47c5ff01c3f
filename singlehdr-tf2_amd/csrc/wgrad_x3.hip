// Weight gradient of the fp32 convolution path on the fp16 matrix pipe by OPERAND SPLITTING (the arithmetic of conv_x3.hip applied to
// GradientTape.gradient w.r.t. the Conv2D kernels, joint_training.py:185-186, train.py:175-244, finetune_real_dataset.py:177):
//
//   dW[tap][ci_off + ci][co] += x_scale * sum_p X[p + tap][ci] * dZ[p][co]
//
// Both operands are activations, so both are split:  X 2^Tx = Xh + Xl 2^-11,  dZ 2^Tz = Zh + Zl 2^-11  (Xh = fp16(X 2^Tx),
// Xl = fp16((X 2^Tx - Xh) 2^11); Tx, Tz bring the tensors' range slots -- device words holding max |X|, max |dZ| -- to [2^10, 2^11):
// output gradients sit far below the fp16 range).  Then
//   X dZ 2^(Tx+Tz) = Xh Zh + 2^-11 (Xl Zh + Xh Zl) + (dropped: Xl Zl 2^-22)
// The two cross terms share the factor 2^-11: TWO fp32 accumulator sets, hi += Xh Zh and lo += Xl Zh + Xh Zl, three MFMAs per operand
// pair and no scaling instruction in the loop; the epilogue adds x_scale 2^-(Tx+Tz) (hi + 2^-11 lo) to dW with fp32 atomics.  Per-product
// error <= 3 * 2^-22, as in the forward kernel.
// The four fp16 planes come from ONE split pass per tensor (shdr_x3_split_planes_f32: 4 B read, 4 B written per element); the kernel is
// the per-tap kernel of wgrad_f16.hip with the planes staged side by side: pixel-major LDS images filled by LDS-DMA (bank swizzle on the
// source side), fragments by the transposing read ds_read_b64_tr_b16, one block = one filter tap x one (CI_T x CO_T) tile x one pixel
// slice, grid cut to one round of the chip's block slots.  Per 32-pixel chunk a wave issues 3 MT NT MFMAs for 2 (MT + NT) operand
// reads (the fp16 kernel: MT NT for MT + NT) and the block streams twice the bytes for three times the FLOPs: 96 flop/B at 128 x 128.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short sv4;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) sv4* lsv4_t;

__device__ __attribute__((aligned(16))) unsigned g_wx_zero_page[4] = {0u, 0u, 0u, 0u};

struct WgradXArgs {
  const _Float16* xh;  // [N,H,W,Cx] high plane of X 2^Tx
  const _Float16* xl;  // low plane, scaled by 2^11
  const _Float16* zh;  // [N,Ho,Wo,Cz]
  const _Float16* zl;
  const unsigned* xr;  // range slots the planes were split with
  const unsigned* zr;
  float* dw;           // [KH*KW][Ct][Cout]
  int N, H, W, Cx, Cz, Ct, ci_off, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo;
  int npix, slice, tiles_m, tiles_n;
  int ci_valid, co_valid;
  float x_scale;
};

constexpr int PK = 32;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
template <int T>
__host__ __device__ constexpr int win_swz(int p) {
  return T >= 128 ? ((p & 3) | (((p >> 3) & 1) << 2)) : T == 64 ? (((p >> 1) & 1) | (((p >> 3) & 1) << 1)) : T == 32 ? ((p >> 3) & 1) : 0;
}
// exponent T of the power of two that brings the bound in the slot to [2^10, 2^11) (0 for an empty, zero or non-finite slot) -- the ONE
// place the split pass and the kernel take it from
__device__ __forceinline__ int range_exponent(const unsigned* slot) {
  const unsigned b = slot ? *slot : 0u;
  if (b == 0u || b >= 0x7f800000u) return 0;
  int ex;
  frexpf(__uint_as_float(b), &ex);
  const int T = 11 - ex;
  return T < -126 ? -126 : (T > 126 ? 126 : T);
}

// x -> (fp16(x 2^T), fp16((x 2^T - high) 2^11)), 8 elements (two 16-byte stores) per thread and iteration
__global__ __launch_bounds__(256) void x3_split_planes_kernel(const float* __restrict__ x, long n8, const unsigned* __restrict__ slot,
                                                              _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const float s = ldexpf(1.0f, range_exponent(slot)), k2048 = 2048.0f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
    unsigned h[4], l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float v0 = p < 2 ? a[2 * p] : b[2 * p - 4], v1 = p < 2 ? a[2 * p + 1] : b[2 * p - 3];
      float t0, t1;
      asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[p]) : "v"(v0), "s"(s));
      asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[p]) : "v"(v1), "s"(s));
      asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(t0) : "v"(v0), "s"(s), "v"(h[p]));
      asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(t1) : "v"(v1), "s"(s), "v"(h[p]));
      asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(l[p]) : "v"(t0), "s"(k2048));
      asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(l[p]) : "v"(t1), "s"(k2048));
    }
    reinterpret_cast<uint4*>(hi)[i] = make_uint4(h[0], h[1], h[2], h[3]);
    reinterpret_cast<uint4*>(lo)[i] = make_uint4(l[0], l[1], l[2], l[3]);
  }
}

template <int CI_T, int CO_T>
__global__ __launch_bounds__(256, 2) void wgrad_x3_kernel(const WgradXArgs a) {
  constexpr int TM = CI_T / 16, TN = CO_T / 16;
  constexpr int WM = 2, WN = 2, MT = TM / WM, NT = TN / WN;
  static_assert(MT * WM == TM && NT * WN == TN, "wave tiles must cover the block tile");
  constexpr int XI = CI_T / 16 / 4, ZI = CO_T / 16 / 4;              // 1 KiB DMA instructions per wave, plane and chunk
  static_assert(XI >= 1 && ZI >= 1, "tiles of at least 64 channels");
  constexpr int XP = CI_T / 8, ZP = CO_T / 8;                        // 16-byte pieces per pixel row
  constexpr int XH = 0, XL = PK * CI_T, ZH = 2 * PK * CI_T, ZL = 2 * PK * CI_T + PK * CO_T;     // half offsets of the four plane images
  constexpr int CHUNK_HALVES = 2 * PK * (CI_T + CO_T);

  extern __shared__ __attribute__((aligned(16))) _Float16 wsm[];     // [2][ Xh | Xl | Zh | Zl ]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntaps = a.KH * a.KW;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tap = t % ntaps; t /= ntaps;
  const int tn = t % a.tiles_n; t /= a.tiles_n;
  const int tm = t % a.tiles_m; t /= a.tiles_m;
  const int slice_id = t;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int ci0 = tm * CI_T, co0 = tn * CO_T;
  const int p_begin = slice_id * a.slice;
  const int p_end = min(p_begin + a.slice, a.npix);
  const int nchunks = (p_end - p_begin + PK - 1) / PK;
  const _Float16* zero = reinterpret_cast<const _Float16*>(g_wx_zero_page);

  int xp[XI], xc[XI], zp[ZI], zc[ZI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int Q = (wave * XI + i) * 64 + lane;
    xp[i] = Q / XP;
    const int pq = Q % XP;
    xc[i] = 8 * ((((pq >> 1) ^ win_swz<CI_T>(xp[i])) << 1) | (pq & 1));
  }
#pragma unroll
  for (int i = 0; i < ZI; ++i) {
    const int Q = (wave * ZI + i) * 64 + lane;
    zp[i] = Q / ZP;
    const int pq = Q % ZP;
    zc[i] = 8 * ((((pq >> 1) ^ win_swz<CO_T>(zp[i])) << 1) | (pq & 1));
  }
  int s_ow[XI], s_oh[XI], s_n[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int p = p_begin + xp[i];
    s_ow[i] = p % a.Wo;
    const int q = p / a.Wo;
    s_oh[i] = q % a.Ho;
    s_n[i] = q / a.Ho;
  }
  int nx_chunk = 0;
  auto dma_chunk = [&](_Float16* B) {
    const int p0 = p_begin + nx_chunk * PK;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int p = p0 + xp[i];
      const int ih = s_oh[i] * a.stride - a.pad_t + kh, iw = s_ow[i] * a.stride - a.pad_l + kw;
      const bool ok = p < p_end && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W && ci0 + xc[i] < a.Cx;
      const size_t off = (size_t)((s_n[i] * a.H + ih) * a.W + iw) * a.Cx + ci0 + xc[i];
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? a.xh + off : zero), (lptr_t)(B + XH + (wave * XI + i) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? a.xl + off : zero), (lptr_t)(B + XL + (wave * XI + i) * 512), 16, 0, 0);
      s_ow[i] += PK;
      while (s_ow[i] >= a.Wo) {
        s_ow[i] -= a.Wo;
        if (++s_oh[i] == a.Ho) { s_oh[i] = 0; ++s_n[i]; }
      }
    }
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
      const int p = p0 + zp[i];
      const bool ok = p < p_end && co0 + zc[i] < a.Cz;
      const size_t off = (size_t)p * a.Cz + co0 + zc[i];
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? a.zh + off : zero), (lptr_t)(B + ZH + (wave * ZI + i) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? a.zl + off : zero), (lptr_t)(B + ZL + (wave * ZI + i) * 512), 16, 0, 0);
    }
    ++nx_chunk;
  };

  f32x4 hi[MT][NT], lo[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      hi[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
      lo[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  int x_rd[2][MT], z_rd[2][NT];                                // half offsets inside a plane image
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * g + 4 * h + q;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) x_rd[h][mi] = row * CI_T + 16 * ((wm * MT + mi) ^ win_swz<CI_T>(row)) + 4 * pp;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) z_rd[h][ni] = row * CO_T + 16 * ((wn * NT + ni) ^ win_swz<CO_T>(row)) + 4 * pp;
  }
  union Frag {
    sv4 h[2];
    f16x8 v;
  };
  auto rd = [&](const _Float16* img, int o0, int o1) __attribute__((always_inline)) {
    Frag f;
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lsv4_t)(img + o0));
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lsv4_t)(img + o1));
    return f.v;
  };

  dma_chunk(wsm);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    const _Float16* B = wsm + (c & 1) * CHUNK_HALVES;
    f16x8 xah[MT], xal[MT], zbh[NT], zbl[NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      xah[mi] = rd(B + XH, x_rd[0][mi], x_rd[1][mi]);
      xal[mi] = rd(B + XL, x_rd[0][mi], x_rd[1][mi]);
    }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      zbh[ni] = rd(B + ZH, z_rd[0][ni], z_rd[1][ni]);
      zbl[ni] = rd(B + ZL, z_rd[0][ni], z_rd[1][ni]);
    }
    if (c + 1 < nchunks) dma_chunk(wsm + ((c + 1) & 1) * CHUNK_HALVES);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        hi[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xah[mi], zbh[ni], hi[mi][ni], 0, 0, 0);
        lo[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xal[mi], zbh[ni], lo[mi][ni], 0, 0, 0);
      }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) lo[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xah[mi], zbl[ni], lo[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  // D[ci][co]: lane holds column co = lane & 15 of rows ci = 4 * (lane >> 4) + e; dW += x_scale 2^-(Tx+Tz) (hi + 2^-11 lo)
  const float sc = a.x_scale * ldexpf(1.0f, -range_exponent(a.xr)) * ldexpf(1.0f, -range_exponent(a.zr));
  const int fi = lane & 15, fg = lane >> 4;
  float* out = a.dw + ((size_t)tap * a.Ct + a.ci_off) * a.Cout;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int co = co0 + (wn * NT + ni) * 16 + fi;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ci = ci0 + (wm * MT + mi) * 16 + 4 * fg + e;
        if (ci < a.ci_valid && co < a.co_valid) atomicAdd(out + (size_t)ci * a.Cout + co, (hi[mi][ni][e] + lo[mi][ni][e] * (1.0f / 2048.0f)) * sc);
      }
    }
}

template <int CI_T, int CO_T>
int launch_wgrad_x3(WgradXArgs& a, hipStream_t st) {
  constexpr int lds = 2 * 2 * PK * (CI_T + CO_T) * 2;
  a.tiles_m = (a.Cx + CI_T - 1) / CI_T;
  a.tiles_n = (a.Cz + CO_T - 1) / CO_T;
  const long tiles = (long)a.KH * a.KW * a.tiles_m * a.tiles_n;
  static long slots_of[shdr::kMaxDevices] = {};
  const int dev_slot = shdr::device_slot();
  if (slots_of[dev_slot] == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_x3_kernel<CI_T, CO_T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    slots_of[dev_slot] = shdr::block_slots(wgrad_x3_kernel<CI_T, CO_T>, 256, lds);
    if (slots_of[dev_slot] < 1) return shdr::fail(SHDR_E_ARCH, "wgrad_x3: occupancy query failed");
  }
  // one round of the chip's block slots (wgrad_f16.hip: every block pays a DMA prologue and a tile of atomics)
  long slice = shdr::slice_for_rounds(slots_of[dev_slot], tiles, a.npix, 1024);
  slice = (slice + PK - 1) / PK * PK;
  a.slice = (int)slice;
  const long nslices = (a.npix + slice - 1) / slice;
  if (tiles * nslices > 0x7fffffffL) return shdr::fail(SHDR_E_SHAPE, "wgrad_x3: grid too large");
  hipLaunchKernelGGL((wgrad_x3_kernel<CI_T, CO_T>), dim3((unsigned)(tiles * nslices)), dim3(256), lds, st, a);
  return shdr::check_launch("wgrad_x3_kernel");
}

}  // namespace

// x [n] fp32 -> the two fp16 planes of x 2^T (T from the range slot: conv_x3.hip "Range"); n % 8 == 0, 16-byte aligned pointers
extern "C" int shdr_x3_split_planes_f32(const float* x, int64_t n, const float* range, void* hi, void* lo, void* stream) {
  SHDR_REQUIRE(x && range && hi && lo, SHDR_E_NULL, "x3_split_planes: null pointer");
  SHDR_REQUIRE(n > 0 && n % 8 == 0 && shdr::aligned16(x) && shdr::aligned16(hi) && shdr::aligned16(lo), SHDR_E_ALIGN,
               "x3_split_planes: n must be a positive multiple of 8, pointers 16-byte aligned");
  hipLaunchKernelGGL(x3_split_planes_kernel, dim3(shdr::stream_grid(n / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, (long)(n / 8),
                     reinterpret_cast<const unsigned*>(range), reinterpret_cast<_Float16*>(hi), reinterpret_cast<_Float16*>(lo));
  return shdr::check_launch("x3_split_planes");
}

// 1 if the split-operand weight gradient takes source `which` of the layer: whole 64-channel tiles on both sides
extern "C" int shdr_conv2d_wgrad_x3_ok_f32(const shdr_conv2d_desc* d, int which) {
  if (!d || (which != 0 && !(which == 1 && d->C2 > 0))) return 0;
  const int Cx = which ? d->C2 : d->C1, cout = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  // (a source whose channel count is not a multiple of the 64-channel tile -- the 96-channel front end of the Linearization-Net stem --
  //  has its last tile loaded with zeros beyond Cx and stored up to ci_valid)
  if (Cx % 32 || Cx < 64 || cout % 64 || cout != d->Cout || SHDR_ENV("SHDR_NO_X3") || SHDR_ENV("SHDR_NO_WGRAD_X3")) return 0;
  return (long)d->N * d->Ho * d->Wo < (1L << 31) && (long)d->N * d->H * d->W * Cx < (1L << 32) ? 1 : 0;
}

extern "C" int shdr_conv2d_wgrad_x3_f32(const shdr_conv2d_desc* d, const void* xh, const void* xl, int which, const void* zh, const void* zl,
                                        const float* x_range, const float* z_range, float* dw, void* stream) {
  SHDR_REQUIRE(d && xh && xl && zh && zl && x_range && z_range && dw, SHDR_E_NULL, "wgrad_x3: null pointer");
  SHDR_REQUIRE(shdr_conv2d_wgrad_x3_ok_f32(d, which), SHDR_E_SHAPE, "wgrad_x3: layer not taken (64-channel tiles on both sides)");
  SHDR_REQUIRE(shdr::aligned16(xh) && shdr::aligned16(xl) && shdr::aligned16(zh) && shdr::aligned16(zl), SHDR_E_ALIGN, "wgrad_x3: planes must be 16-byte aligned");
  WgradXArgs a{};
  a.xh = reinterpret_cast<const _Float16*>(xh); a.xl = reinterpret_cast<const _Float16*>(xl);
  a.zh = reinterpret_cast<const _Float16*>(zh); a.zl = reinterpret_cast<const _Float16*>(zl);
  a.xr = reinterpret_cast<const unsigned*>(x_range); a.zr = reinterpret_cast<const unsigned*>(z_range);
  a.dw = dw;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cx = which ? d->C2 : d->C1; a.Cz = d->Cout;
  a.Ct = d->C1 + d->C2;
  a.ci_off = which ? d->C1 : 0;
  a.ci_valid = a.Cx;
  a.Cout = d->Cout; a.co_valid = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.Ho = d->Ho; a.Wo = d->Wo;
  a.npix = d->N * d->Ho * d->Wo;
  a.x_scale = which ? d->x2_scale : 1.0f;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if ((a.Cx % 128 == 0 || a.Cx % 128 > 64) && a.Cz % 128 == 0) return launch_wgrad_x3<128, 128>(a, st);
  // (96 channels -- the stem of the Linearization-Net: ONE 128-row tile with a quarter of its rows zero, 1.93 ms at 32 x 256^2 against 2.81
  //  on two 64-row tiles and 3.0 on the exact fp32 kernel, split passes included)
  if (a.Cx % 128 == 0 || a.Cx % 128 > 64) return launch_wgrad_x3<128, 64>(a, st);
  if (a.Cz % 128 == 0) return launch_wgrad_x3<64, 128>(a, st);
  return launch_wgrad_x3<64, 64>(a, st);
}
