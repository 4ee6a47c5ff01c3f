// Weight gradient of the native-fp16 convolution path (BASELINE configs[4]) on gfx950:
//
//   dW[tap][ci_off + ci][co] += x_scale * sum_p X[p + tap][ci] * dZ[p][co]          (fp32 accumulation and fp32 atomics)
//
// with X and dZ NHWC fp16.  The contraction runs over PIXELS, but both operands are channel-contiguous in memory, and an operand
// of v_mfma_f32_16x16x32_f16 wants 8 consecutive k (= pixels) of ONE channel per lane: the LDS images stay pixel-major
// ([32 pixels][T channels], filled by global_load_lds_dwordx4) and the fragments are read with the transposing LDS read
// ds_read_b64_tr_b16 (a 16-lane group reads a 4-pixel x 16-channel block and every lane receives one channel's four pixels):
// two of them are one MFMA operand.  Bank swizzle on the SOURCE side: the 32-byte channel windows of pixel row p are XOR-ed
// with s(p) so that the 8 rows {q, 8 + q} one half-wave reads hit 8 different bank groups (T = 128: s = (p & 3) | ((p >> 3) & 1) << 2,
// T = 64: ((p >> 1) & 1) | ((p >> 3) & 1) << 1, T = 32: (p >> 3) & 1; T = 16 is 2-way).
// One block = one filter tap x one (CI_T x CO_T) tile x one slice of pixels, KC chunks of 32 pixels per pipeline stage.
// Replaces GradientTape.gradient w.r.t. the Conv2D kernels in finetune_real_dataset.py:177.
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

__device__ __attribute__((aligned(16))) unsigned g_wh_zero_page[4] = {0u, 0u, 0u, 0u};

struct WgradHArgs {
  const _Float16* x;   // [N,H,W,Cx]
  const _Float16* dz;  // [N,Ho,Wo,Cz]
  float* dw;           // [KH*KW][Ct][Cout]
  int N, H, W, Cx, Cz, Ct, ci_off, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo;
  int npix, slice, tiles_m, tiles_n;
  int ci_valid, co_valid;     // rows / columns of dW that exist (<= Cx / Cz: zero-padded channels carry no gradient)
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

constexpr int pick_wm(int tm, int tn) { return (tm >= 2 && (tn >= 2 || tm >= 4)) ? (tn >= 2 ? 2 : 4) : (tm >= 2 && tn == 1 ? (tm >= 4 ? 4 : 2) : 1); }

// NWV = 4 waves, two blocks per CU (tiles up to 128 x 128), or NWV = 8 waves and ONE block per CU for the 256-wide tiles: the
// per-tap kernel streams (CI_T + CO_T) * 64 bytes per 32-pixel chunk from L2 for CI_T * CO_T * 64 flop, i.e. 64 flop/B at 128 x 128
// -- 530-650 TFLOP/s measured is what ~9 TB/s of L2 -> LDS traffic gives -- and 128 flop/B at 256 x 256.
template <int CI_T, int CO_T, int KC, int NWV = 4, int WM_ = 0>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) void wgrad_f16_kernel(const WgradHArgs a) {
  constexpr int TM = CI_T / 16, TN = CO_T / 16;
  constexpr int WM = WM_ ? WM_ : pick_wm(TM, TN);
  constexpr int WN = (NWV / WM) < TN ? (NWV / WM) : TN;
  constexpr int MT = TM / WM, NT = TN / WN;
  static_assert(MT * WM == TM && NT * WN == TN, "wave tiles must cover the block tile");
  constexpr bool PER_CHUNK = (MT + NT) * KC > 16;                    // fragments of one chunk at a time: the accumulators need the registers
  static_assert(!PER_CHUNK || WM * WN == NWV, "the per-chunk loop has every wave computing");
  constexpr int XI_TOTAL = CI_T / 16, ZI_TOTAL = CO_T / 16;          // wave DMA instructions per chunk (1 KiB each)
  constexpr int XI = (XI_TOTAL + NWV - 1) / NWV, ZI = (ZI_TOTAL + NWV - 1) / NWV;
  constexpr int XP = CI_T / 8, ZP = CO_T / 8;                        // 16-byte pieces per pixel row
  constexpr int CHUNK_HALVES = PK * (CI_T + CO_T);
  constexpr int STAGE_HALVES = KC * CHUNK_HALVES;

  extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];     // [2][KC][ X: 32 x CI_T | Z: 32 x CO_T ]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool computing = wave < WM * WN;
  const int wm = (wave / WN) % WM, wn = wave % WN;
  // 1-D grid, XCD-aware: consecutive logical ids share an XCD (and its L2), and the taps of one (tile, pixel slice) are consecutive --
  // the nine blocks that stream the same X / dZ slice fetch it into ONE L2 instead of eight
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
  const int nstages = (nchunks + KC - 1) / KC;
  const _Float16* zero = reinterpret_cast<const _Float16*>(g_wh_zero_page);

  // lane -> (pixel of the chunk, logical 16-byte piece) of each DMA instruction
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
  auto dma_chunk = [&](_Float16* Xb, _Float16* Zb) {
    const int p0 = p_begin + nx_chunk * PK;
    if (wave * XI < XI_TOTAL) {
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        if (wave * XI + i < XI_TOTAL) {
          const int p = p0 + xp[i];
          const int ih = s_oh[i] * a.stride - a.pad_t + kh, iw = s_ow[i] * a.stride - a.pad_l + kw;
          const bool ok = p < p_end && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W && ci0 + xc[i] < a.Cx;
          const _Float16* src = ok ? a.x + ((size_t)((s_n[i] * a.H + ih) * a.W + iw) * a.Cx + ci0 + xc[i]) : zero;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Xb + (wave * XI + i) * 512), 16, 0, 0);
          s_ow[i] += PK;
          while (s_ow[i] >= a.Wo) {
            s_ow[i] -= a.Wo;
            if (++s_oh[i] == a.Ho) { s_oh[i] = 0; ++s_n[i]; }
          }
        }
      }
    }
    if (wave * ZI < ZI_TOTAL) {
#pragma unroll
      for (int i = 0; i < ZI; ++i) {
        if (wave * ZI + i < ZI_TOTAL) {
          const int p = p0 + zp[i];
          const bool ok = p < p_end && co0 + zc[i] < a.Cz;
          const _Float16* src = ok ? a.dz + ((size_t)p * a.Cz + co0 + zc[i]) : zero;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Zb + (wave * ZI + i) * 512), 16, 0, 0);
        }
      }
    }
    ++nx_chunk;
  };
  auto dma_stage = [&](int buf) {
    _Float16* base = hsm + buf * STAGE_HALVES;
#pragma unroll
    for (int c = 0; c < KC; ++c) dma_chunk(base + c * CHUNK_HALVES, base + c * CHUNK_HALVES + PK * CI_T);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed reads: lane = (k-group g = lane >> 4, row q = (lane & 15) >> 2, piece pp = lane & 3): pixel row 8g + 4h + q,
  // 8 bytes at [16-channel window w][4 pp]; the lane receives channel (lane & 15) of the window for pixels 8g + 4h .. + 3
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  int x_rd[2][MT], z_rd[2][NT];                                // half offsets inside a chunk image
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * g + 4 * h + q;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int w = wm * MT + mi;
      x_rd[h][mi] = row * CI_T + 16 * (w ^ win_swz<CI_T>(row)) + 4 * pp;
    }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int w = wn * NT + ni;
      z_rd[h][ni] = PK * CI_T + row * CO_T + 16 * (w ^ win_swz<CO_T>(row)) + 4 * pp;
    }
  }
  union Frag {
    sv4 h[2];
    f16x8 v;
  };

  dma_stage(0);
  __syncthreads();
#pragma unroll 1
  for (int st = 0; st < nstages; ++st) {
    _Float16* base = hsm + (st & 1) * STAGE_HALVES;
    if constexpr (PER_CHUNK) {
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        Frag xa[MT], zb[NT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          xa[mi].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lsv4_t)(base + c * CHUNK_HALVES + x_rd[0][mi]));
          xa[mi].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lsv4_t)(base + c * CHUNK_HALVES + x_rd[1][mi]));
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          zb[ni].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lsv4_t)(base + c * CHUNK_HALVES + z_rd[0][ni]));
          zb[ni].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lsv4_t)(base + c * CHUNK_HALVES + z_rd[1][ni]));
        }
        if (c == 0 && st + 1 < nstages) dma_stage((st + 1) & 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[mi].v, zb[ni].v, acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      __syncthreads();
      continue;
    }
    Frag xa[KC][MT], zb[KC][NT];
    if (computing) {
#pragma unroll
      for (int c = 0; c < KC; ++c) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          xa[c][mi].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lsv4_t)(base + c * CHUNK_HALVES + x_rd[0][mi]));
          xa[c][mi].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lsv4_t)(base + c * CHUNK_HALVES + x_rd[1][mi]));
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          zb[c][ni].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lsv4_t)(base + c * CHUNK_HALVES + z_rd[0][ni]));
          zb[c][ni].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lsv4_t)(base + c * CHUNK_HALVES + z_rd[1][ni]));
        }
      }
    }
    if (st + 1 < nstages) dma_stage((st + 1) & 1);
    if (computing) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[c][mi].v, zb[c][ni].v, acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
  }

  // D[ci][co]: lane holds column co = lane & 15 of rows ci = 4 * (lane >> 4) + e; one wave instruction adds 4 rows x 64 bytes
  if (computing) {
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
          if (ci < a.ci_valid && co < a.co_valid) atomicAdd(out + (size_t)ci * a.Cout + co, acc[mi][ni][e] * a.x_scale);
        }
      }
  }
}

template <int CI_T, int CO_T, int NWV = 4, int WM_ = 0>
int launch_wgrad_f16(WgradHArgs& a, hipStream_t st) {
  constexpr int KC = (CI_T + CO_T) <= 128 ? 4 : 2;                   // >= 16 KiB per stage
  constexpr int lds = 2 * KC * PK * (CI_T + CO_T) * 2;
  a.tiles_m = (a.Cx + CI_T - 1) / CI_T;
  a.tiles_n = (a.Cz + CO_T - 1) / CO_T;
  const long tiles = (long)a.KH * a.KW * a.tiles_m * a.tiles_n;
  // Pixel slices.  Every block pays a DMA prologue and CI_T x CO_T atomics, so the grid is cut to fill whole ROUNDS of the chip's
  // block slots (CUs x the kernel's occupancy) -- ONE round unless SHDR_WGRAD_ROUNDS says otherwise:
  // one round of 252 blocks ran the 256 x 256 tile at 821-950 TFLOP/s where four rounds of 1008 ran at 676-843, and a grid of
  // 1029 blocks on 512 slots (the 7x7 stem) spent a third of its time in a round of 5 blocks.
  static long slots_of[shdr::kMaxDevices] = {};
  const int dev_slot = shdr::device_slot();
  if (slots_of[dev_slot] == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f16_kernel<CI_T, CO_T, KC, NWV, WM_>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    int dev = 0, cus = 0, occ = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, wgrad_f16_kernel<CI_T, CO_T, KC, NWV, WM_>, NWV * 64, lds) != hipSuccess ||
        cus < 1 || occ < 1)
      return shdr::fail(SHDR_E_ARCH, "wgrad_f16: occupancy query failed");
    slots_of[dev_slot] = (long)cus * occ;
  }
  const long slots = slots_of[dev_slot];
  int rounds = 1;
  if (const char* e = SHDR_ENV("SHDR_WGRAD_ROUNDS")) rounds = atoi(e);
  long slice = a.npix;
  bool chosen = false;
  for (int r = 1; r <= 64 && !(chosen && r > rounds); ++r) {
    long ns = slots * r / tiles;
    if (ns < 1) continue;
    if (ns > a.npix / 1024) ns = a.npix >= 2048 ? a.npix / 1024 : 1;   // slices of >= 1024 pixels
    const long sl = (a.npix + ns - 1) / ns;
    if (sl >= 1536 || !chosen) slice = sl;
    chosen = true;
  }
  slice = (slice + KC * PK - 1) / (KC * PK) * (KC * PK);
  a.slice = (int)slice;
  const long nslices = (a.npix + slice - 1) / slice;
  if (tiles * nslices > 0x7fffffffL) return shdr::fail(SHDR_E_SHAPE, "wgrad_f16: grid too large");
  hipLaunchKernelGGL((wgrad_f16_kernel<CI_T, CO_T, KC, NWV, WM_>), dim3((unsigned)(tiles * nslices)), dim3(NWV * 64), lds, st, a);
  return shdr::check_launch("wgrad_f16_kernel");
}

template <int CI_T>
int dispatch_co(WgradHArgs& a, hipStream_t st) {
  if (a.Cz % 128 == 0) return launch_wgrad_f16<CI_T, 128>(a, st);
  if (a.Cz % 64 == 0) return launch_wgrad_f16<CI_T, 64>(a, st);
  if (a.Cz % 32 == 0) return launch_wgrad_f16<CI_T, 32>(a, st);
  return launch_wgrad_f16<CI_T, 16>(a, st);
}

}  // namespace

extern "C" int shdr_conv2d_wgrad_alltaps_ok_f16(const shdr_conv2d_desc* d, int which, int dz_channels);
extern "C" int shdr_conv2d_wgrad_alltaps_f16(const shdr_conv2d_desc* d, const void* x, int which, const void* dz, int dz_channels,
                                             int c1_rows, int c2_rows, float* dw, void* stream);

extern "C" int shdr_conv2d_wgrad_f16(const shdr_conv2d_desc* d, const void* x, int which, const void* dz, int dz_channels, int c1_rows,
                                     int c2_rows, float* dw, void* stream) {
  SHDR_REQUIRE(d && x && dz && dw, SHDR_E_NULL, "wgrad_f16: null pointer");
  SHDR_REQUIRE(which == 0 || (which == 1 && d->C2 > 0), SHDR_E_SHAPE, "wgrad_f16: `which` selects x1 (0) or x2 (1)");
  SHDR_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C1 > 0 && d->C2 >= 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0 &&
                   d->Ho > 0 && d->Wo > 0,
               SHDR_E_SHAPE, "wgrad_f16: non-positive dimension");
  const int Cx = which ? d->C2 : d->C1;                       // channels per pixel of x (possibly zero-padded)
  const int cout = d->cout_valid > 0 ? d->cout_valid : d->Cout;   // columns of dw that are written; d->Cout = its row length
  SHDR_REQUIRE(c1_rows > 0 && c1_rows <= d->C1 && c2_rows >= 0 && c2_rows <= d->C2 && (d->C2 == 0 || c2_rows > 0), SHDR_E_SHAPE,
               "wgrad_f16: dw rows per source (%d, %d) must not exceed the tensor widths (%d, %d)", c1_rows, c2_rows, d->C1, d->C2);
  SHDR_REQUIRE(Cx % 8 == 0 && dz_channels % 8 == 0 && dz_channels >= cout && cout <= d->Cout, SHDR_E_SHAPE,
               "wgrad_f16: channels per pixel must be multiples of 8 (x %d, dz %d >= %d)", Cx, dz_channels, cout);
  SHDR_REQUIRE((long)d->N * d->Ho * d->Wo < (1L << 31) && (long)d->N * d->H * d->W * Cx < (1L << 32), SHDR_E_SHAPE,
               "wgrad_f16: tensor too large");
  SHDR_REQUIRE(shdr::aligned16(x) && shdr::aligned16(dz), SHDR_E_ALIGN, "wgrad_f16: tensors must be 16-byte aligned");
  // stride-1 layers with many pixels: every tap from one staged strip (wgrad_f16_alltaps.hip)
  if (shdr_conv2d_wgrad_alltaps_ok_f16(d, which, dz_channels) && SHDR_ENV("SHDR_NO_ALLTAPS") == nullptr)
    return shdr_conv2d_wgrad_alltaps_f16(d, x, which, dz, dz_channels, c1_rows, c2_rows, dw, stream);
  WgradHArgs a{};
  a.x = reinterpret_cast<const _Float16*>(x);
  a.dz = reinterpret_cast<const _Float16*>(dz);
  a.dw = dw;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cx = Cx; a.Cz = dz_channels;
  a.Ct = c1_rows + c2_rows;
  a.ci_off = which ? c1_rows : 0;
  a.ci_valid = which ? c2_rows : c1_rows;
  a.Cout = d->Cout; a.co_valid = cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.Ho = d->Ho; a.Wo = d->Wo;
  a.npix = d->N * d->Ho * d->Wo;
  a.x_scale = which ? d->x2_scale : 1.0f;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // 256-wide tiles (8 waves, one block per CU) halve the L2 -> LDS traffic per flop of the widest layers
  long min256 = 4096;
  if (const char* e = SHDR_ENV("SHDR_WGRAD_256_MIN_PIXELS")) min256 = atol(e);
  if (SHDR_ENV("SHDR_NO_WGRAD_256") == nullptr && a.npix >= min256) {
    const bool one = a.KH * a.KW == 1;                           // 1x1: few pixels per block, only the widest pay (0.128 -> 0.115 ms at 1024 -> 512)
    if (Cx % 256 == 0 && a.Cz % 256 == 0) return launch_wgrad_f16<256, 256, 8, 4>(a, st);
    if (Cx % 256 == 0 && a.Cz % 128 == 0 && !one) return launch_wgrad_f16<256, 128, 8, 4>(a, st);
    if (Cx % 128 == 0 && a.Cz % 256 == 0 && !one) return launch_wgrad_f16<128, 256, 8, 2>(a, st);
  }
  if (Cx % 128 == 0) return dispatch_co<128>(a, st);
  // 96 input channels (the Linearization-Net stem, 7x7 / 2): ONE ci tile, so dZ is streamed once per tap instead of three times
  // (the 134 MB gradient of the 4 x 1024^2 step: 2.97 -> ms with three 32-channel tiles)
  if (Cx == 96 && a.Cz % 64 == 0) return launch_wgrad_f16<96, 64>(a, st);
  if (Cx % 64 == 0) return dispatch_co<64>(a, st);
  if (Cx % 32 == 0) return dispatch_co<32>(a, st);
  return dispatch_co<16>(a, st);
}
