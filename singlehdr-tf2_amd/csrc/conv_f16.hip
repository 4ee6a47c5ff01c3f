// Native-fp16 convolution for BASELINE configs[4] ("finetune_real_dataset.py with Refinement-Net, 1024x1024 tiles, fp16 MFMA
// conv path") on gfx950: activations are fp16 in HBM (NHWC, half the bytes of the fp32 path everywhere), the filter is a
// packed fp16 copy of the fp32 master weights, accumulation is fp32 on v_mfma_f32_16x16x32_f16.
//
//   GEMM view   D[cout][pixel] = sum_k Wp[k][cout] * X[pixel][k],  k = (tap, cin), 32 k values ("chunk") per MFMA.
//   LDS images  A [BM pixels][32 halves]  and  B [BN couts][32 halves]: 64-byte rows, so ONE ds_read_b128 is a lane's whole
//               MFMA operand (8 consecutive k of one pixel / one cout).  Both images are filled by global_load_lds_dwordx4
//               (lane-linear 1 KiB pieces = 16 rows); the bank swizzle -- physical 16-byte slot = k-group ^ ((-(row >> 2)) & 3),
//               conflict-free for the four 16-lane groups of ds_read_b128 -- is applied on the SOURCE address.
//   filter      packed by conv_pack_filter_f16_kernel as [chunk][Cout][32]: the B rows of a chunk are contiguous 64-byte
//               lines, x2_scale folded into the rows of the second source, K tail zero-filled.
//   chunks      FAST (Ct % 32 == 0, sources split on a 32-channel boundary): channel-chunk outer / tap inner, all scalar;
//               otherwise natural k order with a per-lane (tap, channel) walk in units of 8 channels (C1 % 8 == C2 % 8 == 0:
//               image inputs are zero-padded to 8 channels).
//   pipeline    KC chunks per stage, two stages in LDS: fragments of the whole stage -> registers, DMA of the next stage,
//               MFMAs, one barrier per stage.
//   epilogue    bias + activation in fp32 -> fp16 through LDS -> 16-byte row-contiguous stores; or fp32 output with
//               cout_valid < Cout for the 3-channel heads (image-like tensors stay fp32).
// The same kernel is the input gradient (dgrad): it runs on dZ with the flipped / transposed filter.
// Replaces tf.keras.layers.Conv2D forward and GradientTape.gradient w.r.t. its input in finetune_real_dataset.py:144-178.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include <type_traits>

#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) unsigned g_h_zero_page[4] = {0u, 0u, 0u, 0u};

struct ConvHArgs {
  const _Float16* x1;
  const _Float16* x2;
  const _Float16* wp;      // packed [nchunks][Cout][32]
  const float* bias;
  _Float16* y16;           // fp16 output [N,Ho,Wo,cout_valid] (or null)
  float* y32;              // fp32 output (heads)
  int N, H, W, C1, C2, Ct, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo;
  int K, ntaps, nchunks, fast;
  int tiles_x, tiles_y, nblk_m, nblk_n;
  int act1, cout_valid;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__host__ __device__ inline int swz(int row) { return (-(row >> 2)) & 3; }

// ---- filter packing: fp32 HWIO [KH*KW][Ct][Cout] -> fp16 [chunk][Cout][32] in the chunk order of the conv kernel --------------
__global__ __launch_bounds__(256) void conv_pack_filter_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int ntaps,
                                                                   int Ct, int C1, int Cout, int nchunks, int fast, float x2_scale) {
  const long total = (long)nchunks * Cout * 4;           // one thread = 8 k values (16 bytes)
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int g = (int)(e & 3);
    const long r = e >> 2;
    const int co = (int)(r % Cout);
    const int kc = (int)(r / Cout);
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int tap, c;
      if (fast) {
        tap = kc % ntaps;
        c = (kc / ntaps) * 32 + 8 * g + j;
      } else {
        const int k = kc * 32 + 8 * g + j;
        tap = k / Ct;
        c = k - tap * Ct;
      }
      float f = 0.0f;
      if (tap < ntaps) f = w[((long)tap * Ct + c) * Cout + co] * (c >= C1 ? x2_scale : 1.0f);
      v[j] = (_Float16)f;
    }
    *reinterpret_cast<f16x8*>(wp + ((long)kc * Cout + co) * 32 + 8 * g) = v;
  }
}

template <int BM, int BN, int WM, int WN, bool FAST, int KC>
__global__ __launch_bounds__(256, 2) void conv_f16_kernel(const ConvHArgs a) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int TH = BM / 16;                      // pixel tile = TH rows x 16 columns
  constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
  constexpr int AI = BM / 16 / 4;                  // A DMA instructions per wave and chunk (16 rows each)
  constexpr int BI_TOTAL = BN / 16;                // B DMA instructions per chunk
  constexpr int BI = (BI_TOTAL + 3) / 4;
  constexpr int B_WAVES = BI_TOTAL >= 4 ? 4 : BI_TOTAL;
  constexpr int STAGE_HALVES = KC * (BM + BN) * 32;
  static_assert(AI >= 1, "BM >= 64");

  extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];   // [2][KC][BM + BN][32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int L = xcd_remap(blockIdx.x, a.nblk_m * a.nblk_n);
  const int pn = L % a.nblk_n;
  int pm = L / a.nblk_n;
  const int tx = pm % a.tiles_x;
  pm /= a.tiles_x;
  const int ty = pm % a.tiles_y;
  const int img = pm / a.tiles_y;
  const int n0 = pn * BN, oh0 = ty * TH, ow0 = tx * 16;
  const _Float16* zero = reinterpret_cast<const _Float16*>(g_h_zero_page);

  // ---- A geometry: instruction i of this wave fills tile rows (wave*AI + i)*16 .. +15; lane = (row, physical slot) -------
  int ihb[AI], iwb[AI], akg[AI];
  unsigned rowoff1[AI], rowoff2[AI];
  const unsigned img_base = (unsigned)img * (unsigned)(a.H * a.W);
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int r = (wave * AI + i) * 16 + (lane >> 2);
    akg[i] = (lane & 3) ^ swz(r);                  // logical k-group fetched into physical slot lane & 3
    const int oh = oh0 + (r >> 4), ow = ow0 + (r & 15);
    const bool ok = (oh < a.Ho) && (ow < a.Wo);
    ihb[i] = ok ? oh * a.stride - a.pad_t : -(1 << 28);
    iwb[i] = ow * a.stride - a.pad_l;
    const unsigned pix = ok ? img_base + (unsigned)(ihb[i] * a.W + iwb[i]) : 0u;
    rowoff1[i] = pix * (unsigned)a.C1 + (FAST ? 8u * (unsigned)akg[i] : 0u);
    rowoff2[i] = pix * (unsigned)a.C2 + (FAST ? 8u * (unsigned)akg[i] : 0u);
  }
  // ---- B geometry: packed rows are 64 contiguous bytes per cout -------------------------------------------------------------
  unsigned woff[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int r = (wave * BI + j) * 16 + (lane >> 2);           // cout row of the tile
    woff[j] = (unsigned)(n0 + r) * 32u + 8u * (unsigned)((lane & 3) ^ swz(r));
  }

  // chunk walk: scalars when FAST (tap inner, channel chunk outer); per instruction and lane otherwise
  int nx_tap = 0, nx_kh = 0, nx_kw = 0, nx_c = 0, nx_kc = 0;
  int t_kh[AI], t_kw[AI], t_c[AI];
  if (!FAST) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      t_kh[i] = 0; t_kw[i] = 0; t_c[i] = 8 * akg[i];
      while (t_c[i] >= a.Ct && t_kh[i] < a.KH) {
        t_c[i] -= a.Ct;
        if (++t_kw[i] == a.KW) { t_kw[i] = 0; ++t_kh[i]; }
      }
    }
  }

  auto dma_chunk = [&](_Float16* Ab, _Float16* Bb) {           // one chunk -> LDS images (wave-uniform bases)
    const bool live = nx_kc < a.nchunks;                       // stages are padded with empty chunks
    if (FAST) {
      const int kh = nx_kh, kw = nx_kw;
      const bool second = nx_c >= a.C1;
      const _Float16* src = second ? a.x2 : a.x1;
      const unsigned delta = (unsigned)((kh * a.W + kw) * (second ? a.C2 : a.C1) + (second ? nx_c - a.C1 : nx_c));
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const bool ok = live && (unsigned)(ihb[i] + kh) < (unsigned)a.H && (unsigned)(iwb[i] + kw) < (unsigned)a.W;
        const _Float16* p = ok ? src + (size_t)((second ? rowoff2[i] : rowoff1[i]) + delta) : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(Ab + (wave * AI + i) * 512), 16, 0, 0);
      }
      if (++nx_kw == a.KW) {
        nx_kw = 0;
        if (++nx_kh == a.KH) { nx_kh = 0; nx_c += 32; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int kh = t_kh[i], kw = t_kw[i];
        const bool kvalid = live && kh < a.KH;
        const bool second = kvalid && (t_c[i] >= a.C1);
        const _Float16* src = second ? a.x2 : a.x1;
        const bool ok = kvalid && (unsigned)(ihb[i] + kh) < (unsigned)a.H && (unsigned)(iwb[i] + kw) < (unsigned)a.W;
        const unsigned off = (second ? rowoff2[i] : rowoff1[i]) +
                             (unsigned)((kh * a.W + kw) * (second ? a.C2 : a.C1) + (second ? t_c[i] - a.C1 : t_c[i]));
        const _Float16* p = ok ? src + (size_t)off : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(Ab + (wave * AI + i) * 512), 16, 0, 0);
        t_c[i] += 32;
        while (t_c[i] >= a.Ct && t_kh[i] < a.KH) {
          t_c[i] -= a.Ct;
          if (++t_kw[i] == a.KW) { t_kw[i] = 0; ++t_kh[i]; }
        }
      }
    }
    if (wave < B_WAVES) {
      const _Float16* wb = a.wp + (size_t)nx_kc * a.Cout * 32;
#pragma unroll
      for (int j = 0; j < BI; ++j) {
        if (wave * BI + j < BI_TOTAL) {
          const _Float16* p = live ? wb + woff[j] : zero;
          __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(Bb + (wave * BI + j) * 512), 16, 0, 0);
        }
      }
    }
    ++nx_kc;
    (void)nx_tap;
  };
  auto dma_stage = [&](int buf) {
    _Float16* base = hsm + buf * STAGE_HALVES;
#pragma unroll
    for (int c = 0; c < KC; ++c) dma_chunk(base + c * (BM + BN) * 32, base + c * (BM + BN) * 32 + BM * 32);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fg = lane >> 4;
  int a_rd[MT], b_rd[NT];                                      // half offsets of this lane's operand inside a chunk image
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int row = wm * MT * 16 + mi * 16 + fi;
    a_rd[mi] = row * 32 + 8 * (fg ^ swz(row));
  }
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    const int row = wn * NT * 16 + ni * 16 + fi;
    b_rd[ni] = BM * 32 + row * 32 + 8 * (fg ^ swz(row));
  }

  const int nstages = (a.nchunks + KC - 1) / KC;
  dma_stage(0);
  __syncthreads();
#pragma unroll 1
  for (int st = 0; st < nstages; ++st) {
    const _Float16* base = hsm + (st & 1) * STAGE_HALVES;
    f16x8 pa[KC][MT], wb[KC][NT];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) pa[c][mi] = *reinterpret_cast<const f16x8*>(base + c * (BM + BN) * 32 + a_rd[mi]);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) wb[c][ni] = *reinterpret_cast<const f16x8*>(base + c * (BM + BN) * 32 + b_rd[ni]);
    }
    if (st + 1 < nstages) dma_stage((st + 1) & 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[c][ni], pa[c][mi], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  // ---- epilogue: lane (fi, fg) holds, per 16x16 tile, couts 4fg..4fg+3 of pixel fi ------------------------------------------
  if (a.y32) {                                                 // fp32 heads (cout_valid channels, scalar stores)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int r = wm * MT * 16 + mi * 16 + fi;
      const int oh = oh0 + (r >> 4), ow = ow0 + (r & 15);
      if (oh >= a.Ho || ow >= a.Wo) continue;
      const size_t pix = ((size_t)img * a.Ho + oh) * a.Wo + ow;
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = n0 + wn * NT * 16 + ni * 16 + 4 * fg + e;
          if (co < a.cout_valid) {
            float t = acc[mi][ni][e];
            if (a.bias) t += a.bias[co];
            a.y32[pix * a.cout_valid + co] = shdr::act_apply(t, a.act1);
          }
        }
    }
    return;
  }
  constexpr int RS = BN + 8;                                   // staged row stride in halves (16-byte aligned)
  _Float16* stage = hsm;                                       // the pipeline buffers are dead (last barrier passed)
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int r = wm * MT * 16 + mi * 16 + fi;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int cl = wn * NT * 16 + ni * 16 + 4 * fg;
      f32x4 v = acc[mi][ni];
      if (a.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n0 + cl);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
      f16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (_Float16)shdr::act_apply(v[e], a.act1);
      *reinterpret_cast<f16x4*>(stage + r * RS + cl) = h;
    }
  }
  __syncthreads();
  constexpr int QR = BN / 8;                                   // 16-byte pieces per tile row
#pragma unroll 2
  for (int e = tid; e < BM * QR; e += 256) {
    const int r = e / QR, q = e - r * QR;
    const int oh = oh0 + (r >> 4), ow = ow0 + (r & 15);
    if (oh >= a.Ho || ow >= a.Wo) continue;
    const size_t pix = ((size_t)img * a.Ho + oh) * a.Wo + ow;
    *reinterpret_cast<f16x8*>(a.y16 + pix * a.Cout + n0 + 8 * q) = *reinterpret_cast<const f16x8*>(stage + r * RS + 8 * q);
  }
}

template <int BM, int BN, int WM, int WN, bool FAST, int KC>
int launch_f16(ConvHArgs& a, hipStream_t st) {
  constexpr int TH = BM / 16;
  a.tiles_x = (a.Wo + 15) / 16;
  a.tiles_y = (a.Ho + TH - 1) / TH;
  a.nblk_m = a.N * a.tiles_y * a.tiles_x;
  a.nblk_n = a.Cout / BN;
  constexpr int pipe = 2 * KC * (BM + BN) * 32 * 2, stage = BM * (BN + 8) * 2;
  constexpr int lds = pipe > stage ? pipe : stage;
  static bool attr_done[shdr::kMaxDevices] = {};
  const int dev_slot = shdr::device_slot();
  if (!attr_done[dev_slot]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16_kernel<BM, BN, WM, WN, FAST, KC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done[dev_slot] = true;
  }
  const long nblk = (long)a.nblk_m * a.nblk_n;
  if (nblk <= 0 || nblk > 0x7fffffffL) return shdr::fail(SHDR_E_SHAPE, "conv2d_f16: grid of %ld blocks", nblk);
  hipLaunchKernelGGL((conv_f16_kernel<BM, BN, WM, WN, FAST, KC>), dim3((unsigned)nblk), dim3(256), lds, st, a);
  return shdr::check_launch("conv_f16_kernel");
}

template <int BM, int BN, int WM, int WN>
int launch_f16_k(ConvHArgs& a, hipStream_t st) {
  // two chunks per stage unless the layer has a single chunk per tap-walk that short (1x1 layers with 32 channels)
  if (a.fast) return a.nchunks >= 2 ? launch_f16<BM, BN, WM, WN, true, 2>(a, st) : launch_f16<BM, BN, WM, WN, true, 1>(a, st);
  return a.nchunks >= 2 ? launch_f16<BM, BN, WM, WN, false, 2>(a, st) : launch_f16<BM, BN, WM, WN, false, 1>(a, st);
}

extern "C" int shdr_conv2d_patch_ok_f16(const shdr_conv2d_desc* d);
extern "C" int shdr_conv2d_fwd_patch_f16(const shdr_conv2d_desc* d, const void* x1, const void* x2, const void* wp, const float* bias,
                                         void* y, int y_is_f32, void* stream);

extern "C" int shdr_conv2d_w3_ok_f16(const shdr_conv2d_desc* d);
extern "C" int shdr_conv2d_fwd_w3_f16(const shdr_conv2d_desc* d, const void* x1, const void* x2, const void* wp, const float* bias, void* y,
                                      void* stream);

inline bool f16_fast(int C1, int C2) { return ((C1 + C2) % 32 == 0) && (C2 == 0 || C1 % 32 == 0); }
inline int f16_nchunks(int ntaps, int C1, int C2) {
  const int Ct = C1 + C2;
  return f16_fast(C1, C2) ? ntaps * (Ct / 32) : (ntaps * Ct + 31) / 32;
}

}  // namespace

extern "C" int64_t shdr_conv2d_packed_filter_elems_f16(int KH, int KW, int C1, int C2, int Cout) {
  if (KH <= 0 || KW <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0) return -1;
  return (int64_t)f16_nchunks(KH * KW, C1, C2) * Cout * 32;
}

extern "C" int shdr_conv2d_pack_filter_f16(const float* w, void* wp, int KH, int KW, int C1, int C2, int Cout, float x2_scale,
                                           void* stream) {
  SHDR_REQUIRE(w && wp, SHDR_E_NULL, "pack_filter_f16: null pointer");
  SHDR_REQUIRE(KH > 0 && KW > 0 && C1 > 0 && C2 >= 0 && Cout > 0, SHDR_E_SHAPE, "pack_filter_f16: non-positive dimension");
  SHDR_REQUIRE(shdr::aligned16(wp), SHDR_E_ALIGN, "pack_filter_f16: wp must be 16-byte aligned");
  const int nchunks = f16_nchunks(KH * KW, C1, C2);
  hipLaunchKernelGGL(conv_pack_filter_f16_kernel, dim3(shdr::stream_grid((long)nchunks * Cout * 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), w, reinterpret_cast<_Float16*>(wp), KH * KW, C1 + C2, C1, Cout, nchunks,
                     f16_fast(C1, C2) ? 1 : 0, C2 > 0 ? x2_scale : 1.0f);
  return shdr::check_launch("conv_pack_filter_f16");
}

extern "C" int shdr_conv2d_fwd_f16(const shdr_conv2d_desc* d, const void* x1, const void* x2, const void* wp, const float* bias,
                                   void* y, int y_is_f32, void* stream) {
  SHDR_REQUIRE(d && x1 && wp && y, SHDR_E_NULL, "conv2d_f16: null desc/x1/wp/y");
  SHDR_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C1 > 0 && d->C2 >= 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 &&
                   d->stride > 0 && d->Ho > 0 && d->Wo > 0,
               SHDR_E_SHAPE, "conv2d_f16: non-positive dimension");
  SHDR_REQUIRE((d->C2 == 0) == (x2 == nullptr), SHDR_E_NULL, "conv2d_f16: x2 must be given iff C2 > 0");
  SHDR_REQUIRE(d->C1 % 8 == 0 && d->C2 % 8 == 0 && d->Cout % 16 == 0, SHDR_E_SHAPE,
               "conv2d_f16: need C1 %% 8 == 0, C2 %% 8 == 0 (16-byte channel groups) and Cout %% 16 == 0, got %d+%d -> %d", d->C1,
               d->C2, d->Cout);
  SHDR_REQUIRE(d->pad_t >= 0 && d->pad_l >= 0 && d->pad_t < d->KH && d->pad_l < d->KW, SHDR_E_SHAPE,
               "conv2d_f16: pad (%d,%d) outside kernel %dx%d", d->pad_t, d->pad_l, d->KH, d->KW);
  SHDR_REQUIRE((long)(d->Ho - 1) * d->stride - d->pad_t < d->H && (long)(d->Wo - 1) * d->stride - d->pad_l < d->W, SHDR_E_SHAPE,
               "conv2d_f16: output %dx%d too large for input %dx%d", d->Ho, d->Wo, d->H, d->W);
  const int cout_valid = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  SHDR_REQUIRE(cout_valid <= d->Cout, SHDR_E_SHAPE, "conv2d_f16: cout_valid > Cout");
  SHDR_REQUIRE(y_is_f32 || cout_valid == d->Cout, SHDR_E_SHAPE, "conv2d_f16: an fp16 output stores every channel (cout_valid == Cout)");
  SHDR_REQUIRE((long)d->N * d->H * d->W * (d->C1 > d->C2 ? d->C1 : d->C2) < (1L << 32) && (long)d->N * d->Ho * d->Wo < (1L << 31),
               SHDR_E_SHAPE, "conv2d_f16: tensor with more than 2^32 elements");
  SHDR_REQUIRE(shdr::aligned16(x1) && (!x2 || shdr::aligned16(x2)) && shdr::aligned16(wp) && shdr::aligned16(y) &&
                   (!bias || shdr::aligned16(bias)),
               SHDR_E_ALIGN, "conv2d_f16: tensors must be 16-byte aligned");
  ConvHArgs a{};
  a.x1 = reinterpret_cast<const _Float16*>(x1);
  a.x2 = reinterpret_cast<const _Float16*>(x2);
  a.wp = reinterpret_cast<const _Float16*>(wp);
  a.bias = bias;
  a.y16 = y_is_f32 ? nullptr : reinterpret_cast<_Float16*>(y);
  a.y32 = y_is_f32 ? reinterpret_cast<float*>(y) : nullptr;
  a.N = d->N; a.H = d->H; a.W = d->W; a.C1 = d->C1; a.C2 = d->C2; a.Ct = d->C1 + d->C2;
  a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride;
  a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.Ho = d->Ho; a.Wo = d->Wo;
  a.ntaps = a.KH * a.KW;
  a.K = a.ntaps * a.Ct;
  a.fast = f16_fast(a.C1, a.C2) ? 1 : 0;
  a.nchunks = f16_nchunks(a.ntaps, a.C1, a.C2);
  a.act1 = d->act1;
  a.cout_valid = cout_valid;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // narrow full-resolution layers (<= 32 channels per tap, 16 / 32 couts): raw patch + resident filter in LDS (conv_f16_patch.hip)
  if (shdr_conv2d_patch_ok_f16(d) && (y_is_f32 || d->act1 != SHDR_ACT_TANH) && SHDR_ENV("SHDR_NO_PATCH") == nullptr)
    return shdr_conv2d_fwd_patch_f16(d, x1, x2, wp, bias, y, y_is_f32, stream);
  // wide 3x3 layers: raw patch per 32-channel chunk instead of nine im2col stagings (conv_f16_w3.hip)
  if (!y_is_f32 && shdr_conv2d_w3_ok_f16(d) && SHDR_ENV("SHDR_NO_W3") == nullptr) return shdr_conv2d_fwd_w3_f16(d, x1, x2, wp, bias, y, stream);
  if (a.Cout % 128 == 0) return launch_f16_k<128, 128, 2, 2>(a, st);
  if (a.Cout % 64 == 0) return launch_f16_k<256, 64, 4, 1>(a, st);
  if (a.Cout % 32 == 0) return launch_f16_k<256, 32, 4, 1>(a, st);
  return launch_f16_k<256, 16, 4, 1>(a, st);
}
