// All-taps weight gradient of the native-fp16 path (BASELINE configs[4]) for stride-1 SAME layers with many pixels:
//
//   dW[tap][ci_off + ci][co] += x_scale * sum_p X[p + tap][ci] * dZ[p][co]        for ALL taps of a KK x KK filter in one block.
//
// wgrad_f16_kernel gives every tap its own block, so X and dZ are streamed from HBM / L2 once per tap: 9 times for the 3x3 layers
// at 1024 x 1024 (1 GB tensors: 211-280 TFLOP/s, HBM-bound), 49 times for the 7x7 layers of the U-Nets (32-64 TFLOP/s).  Here
// a block walks row segments of 32 output pixels; per segment it stages the input STRIP (KK rows x (32 + KK - 1) pixels x CI_T
// channels) and the 32 x CO_T gradient pixels ONCE (LDS-DMA) and forms every tap's operand from the strip with transposing LDS
// reads (ds_read_b64_tr_b16) at the tap's pixel offset; the gradient operand is read once and reused by all taps.
//   MODE 0: the four waves split the TAPS (tap = wave, wave + 4, ...), each wave covers the whole CI_T x CO_T tile  (narrow layers);
//   MODE 1: the four waves split the tile 2 x 2, each wave holds all taps of its quarter                            (64 x 64 tiles, 3x3).
// Same swizzle on the source side as wgrad_f16.hip (32-byte channel windows XOR-ed by the pixel index), fp32 atomics into dW.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short sv4;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) sv4* lsv4_t;

__device__ __attribute__((aligned(16))) unsigned g_wa_zero_page[4] = {0u, 0u, 0u, 0u};

struct WgradAArgs {
  const _Float16* x;   // [N,H,W,Cx]
  const _Float16* dz;  // [N,H,W,Cz]
  float* dw;           // [KK*KK][Ct][Cout]
  int N, H, W, Cx, Cz, Ct, ci_off, Cout;
  int segs_x, nsegs, slice, tiles_n;
  int ci_valid, co_valid;
  float x_scale;
};

template <int T>
__host__ __device__ constexpr int wswz(int p) {
  return T >= 128 ? ((p & 3) | (((p >> 3) & 1) << 2)) : T == 64 ? (((p >> 1) & 1) | (((p >> 3) & 1) << 1)) : T == 32 ? ((p >> 3) & 1) : 0;
}

template <int KK, int CI_T, int CO_T, int MODE>
struct AG {
  static constexpr int NTAPS = KK * KK;
  static constexpr int TM = CI_T / 16, TN = CO_T / 16;
  static constexpr int MT = MODE ? TM / 2 : TM, NT = MODE ? TN / 2 : TN;
  static constexpr int TAPS_W = MODE ? NTAPS : (NTAPS + 3) / 4;       // taps per wave
  static constexpr int SW = 32 + KK - 1;                              // strip pixels per row
  static constexpr int XP = CI_T / 8, ZP = CO_T / 8;                  // 16-byte pieces per pixel
  static constexpr int X_PIECES = KK * SW * XP, Z_PIECES = 32 * ZP;
  static constexpr int XJ = ((X_PIECES + 63) / 64 + 3) / 4, ZJ = ((Z_PIECES + 63) / 64 + 3) / 4;   // DMA instructions per wave
  static constexpr int X_HALVES = XJ * 4 * 512, Z_HALVES = ZJ * 4 * 512;
  static constexpr int STAGE_HALVES = X_HALVES + Z_HALVES;
  static constexpr int STAGE_BYTES = STAGE_HALVES * 2;
  static constexpr int NST = 4 * STAGE_BYTES < 80 * 1024 ? 4 : 3;     // ring of segment stages: two blocks per CU keep their 160 KB
  static constexpr int DPS = XJ + ZJ;                                 // DMA instructions per wave and stage
  static constexpr int LDS_BYTES = NST * STAGE_BYTES;
};

// s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "no wait")
constexpr int vmcnt_only(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }
constexpr int lgkmcnt_only(int n) { return 0xC07F | (n << 8); }

// ds_read_b64_tr_b16 with an immediate byte offset, outside the compiler's view of LDS (see the kernel)
template <int OFF>
__device__ __forceinline__ sv4 ld_tr(unsigned addr) {
  sv4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int T0, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (T0 < N) {
    f(std::integral_constant<int, T0>{});
    static_for<T0 + 1, N>(f);
  }
}

template <int KK, int CI_T, int CO_T, int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_f16_alltaps_kernel(const WgradAArgs a) {
  using G = AG<KK, CI_T, CO_T, MODE>;
  constexpr int MT = G::MT, NT = G::NT, PAD = (KK - 1) / 2;
  extern __shared__ __attribute__((aligned(16))) _Float16 asm_[];     // [2][ X strip | Z ]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tn = blockIdx.x % a.tiles_n, tm = blockIdx.x / a.tiles_n;
  const int ci0 = tm * CI_T, co0 = tn * CO_T;
  const int s_begin = blockIdx.y * a.slice;
  const int s_end = min(s_begin + a.slice, a.nsegs);
  const _Float16* zero = reinterpret_cast<const _Float16*>(g_wa_zero_page);

  // ---- DMA geometry: piece -> (strip row, strip pixel, logical 16-byte piece) -------------------------------------------------
  // Segments run DOWN a 32-pixel column (consecutive segments of a block share KK - 1 of their KK strip rows, which the XCD's L2
  // still holds; walking along a row the re-read came 1024 pixels later, from the Infinity Cache or HBM), so from one segment
  // to the next every piece's source moves by exactly one image row: the element offsets are kept per lane and advanced by a
  // constant, and rebuilt only where the column ends (the per-segment address arithmetic -- two divisions, 64-bit multiplies
  // and bounds tests per piece -- was a third of the loop's vector instructions, and the loop is bound by vector issue).
  int xrp[G::XJ], xsp[G::XJ], xc[G::XJ], zp[G::ZJ], zc[G::ZJ];      // strip row / pixel relative to the segment's first output pixel
  bool xok[G::XJ], zok[G::ZJ];
#pragma unroll
  for (int j = 0; j < G::XJ; ++j) {
    const int piece = (wave * G::XJ + j) * 64 + lane;
    xok[j] = piece < G::X_PIECES;
    const int pix = piece / G::XP, pq = piece - pix * G::XP;
    const int r = pix / G::SW, sp = pix - r * G::SW;
    xrp[j] = r - PAD;
    xsp[j] = sp - PAD;
    xc[j] = 8 * ((((pq >> 1) ^ wswz<CI_T>(sp)) << 1) | (pq & 1));
    xok[j] = xok[j] && ci0 + xc[j] < a.Cx;
  }
#pragma unroll
  for (int j = 0; j < G::ZJ; ++j) {
    const int piece = (wave * G::ZJ + j) * 64 + lane;
    zok[j] = piece < G::Z_PIECES;
    zp[j] = piece / G::ZP;
    const int pq = piece - zp[j] * G::ZP;
    zc[j] = 8 * ((((pq >> 1) ^ wswz<CO_T>(zp[j])) << 1) | (pq & 1));
    zok[j] = zok[j] && co0 + zc[j] < a.Cz;
  }
  // element offsets modulo 2^32 (the host checks N*H*W*C < 2^32; rows above the image wrap and are never dereferenced)
  unsigned xoff[G::XJ], zoff[G::ZJ];
  bool xcol[G::XJ], zcol[G::ZJ];
  int d_seg = s_begin, d_oh = 0;                                      // the next segment to fetch and its image row
  const unsigned rowx = (unsigned)(a.W * a.Cx), rowz = (unsigned)(a.W * a.Cz);
  auto dma_column = [&]() {
    d_oh = d_seg % a.H;
    const int t = d_seg / a.H;
    const int sx = t % a.segs_x, n = t / a.segs_x;
    const int ow0 = sx * 32;
#pragma unroll
    for (int j = 0; j < G::XJ; ++j) {
      const int iw = ow0 + xsp[j];
      xcol[j] = xok[j] && (unsigned)iw < (unsigned)a.W;
      xoff[j] = ((unsigned)(n * a.H + d_oh + xrp[j]) * (unsigned)a.W + (unsigned)iw) * (unsigned)a.Cx + (unsigned)(ci0 + xc[j]);
    }
#pragma unroll
    for (int j = 0; j < G::ZJ; ++j) {
      const int ow = ow0 + zp[j];
      zcol[j] = zok[j] && ow < a.W;
      zoff[j] = ((unsigned)(n * a.H + d_oh) * (unsigned)a.W + (unsigned)ow) * (unsigned)a.Cz + (unsigned)(co0 + zc[j]);
    }
  };
  // fetch segment d_seg into stage `buf` and step to the next one (past the slice's end the last segment is fetched again: the
  // DMA count per wave and iteration stays uniform for the counted waits)
  auto dma_next = [&](int buf) {
    _Float16* Xb = asm_ + buf * G::STAGE_HALVES;
    _Float16* Zb = Xb + G::X_HALVES;
#pragma unroll
    for (int j = 0; j < G::XJ; ++j) {
      const bool ok = xcol[j] && (unsigned)(d_oh + xrp[j]) < (unsigned)a.H;
      const _Float16* src = ok ? a.x + (size_t)xoff[j] : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Xb + (wave * G::XJ + j) * 512), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < G::ZJ; ++j) {
      const _Float16* src = zcol[j] ? a.dz + (size_t)zoff[j] : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Zb + (wave * G::ZJ + j) * 512), 16, 0, 0);
    }
    if (d_seg + 1 < s_end) {
      ++d_seg;
      if (++d_oh == a.H) {
        dma_column();
      } else {
#pragma unroll
        for (int j = 0; j < G::XJ; ++j) xoff[j] += rowx;
#pragma unroll
        for (int j = 0; j < G::ZJ; ++j) zoff[j] += rowz;
      }
    }
  };

  f32x4 acc[G::TAPS_W][MT][NT];
#pragma unroll
  for (int t = 0; t < G::TAPS_W; ++t)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) acc[t][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed reads (wgrad_f16.hip): lane = (k-group g, row q, piece pp) -> pixel 8g + 4h + q of the segment
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int wm = MODE ? (wave >> 1) : 0, wn = MODE ? (wave & 1) : 0;
  union Frag {
    sv4 h[2];
    f16x8 v;
  };
  // LDS byte addresses inside a stage.  The reads are inline asm: the compiler orders every LDS read it knows of behind ALL
  // LDS-DMA writes in flight (s_waitcnt vmcnt(0) -- it cannot tell the ring's buffers apart), which would drain the ring once per
  // segment; hidden in asm their completion is the counted "s_waitcnt lgkmcnt" + register fence in front of each MFMA group.
  // The stage (and in MODE 1 the tap's kh row) is the instruction's immediate offset: the ring is unrolled over its NST stages.
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)asm_;
  constexpr int XR = MODE ? KK : G::TAPS_W;                           // MODE 1: per kw (+ immediate kh row); MODE 0: per tap of this wave
  constexpr int ROW_BYTES = G::SW * CI_T * 2;
  unsigned z_rd[2][NT], x_rd[XR][2][MT];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * g + 4 * h + q;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
      z_rd[h][ni] = lds0 + 2u * (unsigned)(G::X_HALVES + row * CO_T + 16 * ((wn * NT + ni) ^ wswz<CO_T>(row)) + 4 * pp);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int tap = MODE ? i : min(4 * i + wave, G::NTAPS - 1);
      const int kh = MODE ? 0 : tap / KK, kw = MODE ? i : tap - kh * KK;
      const int sp = row + kw;                                        // strip pixel of this lane's row
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        x_rd[i][h][mi] = lds0 + 2u * (unsigned)((kh * G::SW + sp) * CI_T + 16 * ((wm * MT + mi) ^ wswz<CI_T>(sp)) + 4 * pp);
    }
  }

  if (s_begin >= s_end) return;
  // Ring of NST stages: the DMA of segment s + NST - 1 is issued at the top of iteration s, so NST - 1 segments are in flight
  // per block (a segment's MFMAs take ~0.3 us, an L2 / HBM round trip 1-2 us: with the two-stage ring the kernel waited for
  // every segment).  "s_waitcnt vmcnt((NST - 2) * DPS)" at the bottom = this wave's pieces of segment s + 1 have landed, the
  // barrier after it = everybody's.
  dma_column();
#pragma unroll
  for (int i = 0; i < G::NST - 1; ++i) dma_next(i);
  __builtin_amdgcn_s_waitcnt(vmcnt_only((G::NST - 2) * G::DPS));
  __builtin_amdgcn_s_barrier();

  // one segment from stage RB; operand reads run two taps ahead of the MFMAs (xa is a ring of three)
  auto segment = [&](auto rbc) {
    constexpr int RB = decltype(rbc)::value;
    constexpr int SOFF = RB * G::STAGE_BYTES;
    dma_next((RB + G::NST - 1) % G::NST);
    Frag zb[NT], xa[3][MT];
    auto read_x = [&](Frag* dst, auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int i = MODE ? t % KK : t;
      constexpr int off = SOFF + (MODE ? (t / KK) * ROW_BYTES : 0);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int h = 0; h < 2; ++h) dst[mi].h[h] = ld_tr<off>(x_rd[i][h][mi]);
    };
    auto valid = [&](int t) { return MODE || 4 * t + wave < G::NTAPS; };       // wave-uniform, monotone in t
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int h = 0; h < 2; ++h) zb[ni].h[h] = ld_tr<SOFF>(z_rd[h][ni]);
    if (valid(0)) read_x(xa[0], std::integral_constant<int, 0>{});
    if (G::TAPS_W > 1 && valid(1)) read_x(xa[1], std::integral_constant<int, (G::TAPS_W > 1 ? 1 : 0)>{});
    auto tap_step = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if (!valid(t)) return;
      if (t + 2 < G::TAPS_W && valid(t + 2)) {
        read_x(xa[(t + 2) % 3], std::integral_constant<int, (t + 2 < G::TAPS_W ? t + 2 : 0)>{});
        __builtin_amdgcn_s_waitcnt(lgkmcnt_only(4 * MT));
      } else if (t + 1 < G::TAPS_W && valid(t + 1)) {
        __builtin_amdgcn_s_waitcnt(lgkmcnt_only(2 * MT));
      } else {
        __builtin_amdgcn_s_waitcnt(lgkmcnt_only(0));
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) asm volatile("" : "+v"(xa[t % 3][mi].v));
      if (t == 0) {
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) asm volatile("" : "+v"(zb[ni].v));
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[t][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[t % 3][mi].v, zb[ni].v, acc[t][mi][ni], 0, 0, 0);
    };
    static_for<0, G::TAPS_W>(tap_step);
    __builtin_amdgcn_s_waitcnt(vmcnt_only((G::NST - 2) * G::DPS) & lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };
  for (int seg = s_begin;;) {
    segment(std::integral_constant<int, 0>{});
    if (++seg >= s_end) break;
    segment(std::integral_constant<int, 1>{});
    if (++seg >= s_end) break;
    segment(std::integral_constant<int, 2>{});
    if (++seg >= s_end) break;
    if constexpr (G::NST == 4) {
      segment(std::integral_constant<int, 3>{});
      if (++seg >= s_end) break;
    }
  }
  __builtin_amdgcn_s_waitcnt(vmcnt_only(0));                          // the re-fetched tail segments land before the block's LDS is released

  // D[ci][co]: lane holds column co = lane & 15 of rows ci = 4 * (lane >> 4) + e
  const int fi = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int t = 0; t < G::TAPS_W; ++t) {
    const int tap = MODE ? t : 4 * t + wave;
    if (!MODE && tap >= G::NTAPS) continue;
    float* out = a.dw + ((size_t)tap * a.Ct + a.ci_off) * a.Cout;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        const int co = co0 + (wn * NT + ni) * 16 + fi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ci = ci0 + (wm * MT + mi) * 16 + 4 * fg + e;
          if (ci < a.ci_valid && co < a.co_valid) atomicAdd(out + (size_t)ci * a.Cout + co, acc[t][mi][ni][e] * a.x_scale);
        }
      }
  }
}

template <int KK, int CI_T, int CO_T, int MODE>
int launch_alltaps(WgradAArgs& a, hipStream_t st) {
  using G = AG<KK, CI_T, CO_T, MODE>;
  const int tiles_m = (a.Cx + CI_T - 1) / CI_T;
  a.tiles_n = (a.Cz + CO_T - 1) / CO_T;
  const long tiles = (long)tiles_m * a.tiles_n;
  a.segs_x = (a.W + 31) / 32;
  a.nsegs = a.N * a.H * a.segs_x;
  // segment slices: ONE round of the chip's block slots (CUs x occupancy; wgrad_f16.hip has the measurements), at least 32
  // segments (1024 pixels) per block so that the KK^2 x CI_T x CO_T atomics of a block stay cheap
  static long slots_of[shdr::kMaxDevices] = {};
  const int dev_slot = shdr::device_slot();
  if (slots_of[dev_slot] == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f16_alltaps_kernel<KK, CI_T, CO_T, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    int dev = 0, cus = 0, occ = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, wgrad_f16_alltaps_kernel<KK, CI_T, CO_T, MODE>, 256, G::LDS_BYTES) != hipSuccess ||
        cus < 1 || occ < 1)
      return shdr::fail(SHDR_E_ARCH, "wgrad_f16_alltaps: occupancy query failed");
    slots_of[dev_slot] = (long)cus * occ;
  }
  long rounds = 1;
  if (const char* e = SHDR_ENV("SHDR_WGRAD_ROUNDS")) rounds = atol(e) > 0 ? atol(e) : 1;
  long want = slots_of[dev_slot] * rounds / tiles;
  if (want < 1) want = 1;
  long slice = (a.nsegs + want - 1) / want;
  if (slice < 32) slice = 32;
  a.slice = (int)slice;
  const long nslices = (a.nsegs + slice - 1) / slice;
  if (nslices > 65535) return shdr::fail(SHDR_E_SHAPE, "wgrad_f16_alltaps: grid too large");
  hipLaunchKernelGGL((wgrad_f16_alltaps_kernel<KK, CI_T, CO_T, MODE>), dim3((unsigned)tiles, (unsigned)nslices), dim3(256), G::LDS_BYTES, st, a);
  return shdr::check_launch("wgrad_f16_alltaps_kernel");
}

}  // namespace

// 1 if the all-taps kernel takes the weight gradient of source `which` of the layer
extern "C" int shdr_conv2d_wgrad_alltaps_ok_f16(const shdr_conv2d_desc* d, int which, int dz_channels) {
  if (!d || d->stride != 1 || d->KH != d->KW || !(d->KH == 3 || d->KH == 5 || d->KH == 7)) return 0;
  if (d->pad_t != (d->KH - 1) / 2 || d->pad_l != (d->KW - 1) / 2 || d->Ho != d->H || d->Wo != d->W) return 0;
  const int cx = which ? d->C2 : d->C1;
  long min_pixels = 65536;                                            // few pixels: the per-tap kernel with its larger tiles
  if (const char* e = SHDR_ENV("SHDR_ALLTAPS_MIN_PIXELS")) min_pixels = atol(e);
  if ((long)d->N * d->H * d->W < min_pixels) return 0;
  if (d->KH == 3) return ((cx == 8 || cx == 16 || cx == 32 || cx % 64 == 0) && cx <= 128 && (dz_channels == 8 || dz_channels == 16 || dz_channels == 32 || dz_channels == 64)) ? 1 : 0;
  if (d->KH == 5) return ((cx == 16 || cx == 32) && (dz_channels == 16 || dz_channels == 32)) ? 1 : 0;
  return ((cx == 8 || cx == 16) && dz_channels == 16) ? 1 : 0;
}

extern "C" int shdr_conv2d_wgrad_alltaps_f16(const shdr_conv2d_desc* d, const void* x, int which, const void* dz, int dz_channels,
                                             int c1_rows, int c2_rows, float* dw, void* stream) {
  SHDR_REQUIRE(d && x && dz && dw, SHDR_E_NULL, "wgrad_f16_alltaps: null pointer");
  SHDR_REQUIRE(shdr_conv2d_wgrad_alltaps_ok_f16(d, which, dz_channels), SHDR_E_SHAPE, "wgrad_f16_alltaps: layer not taken by this kernel");
  const int Cx = which ? d->C2 : d->C1;
  const int cout = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  SHDR_REQUIRE(c1_rows > 0 && c1_rows <= d->C1 && c2_rows >= 0 && c2_rows <= d->C2 && cout <= d->Cout && cout <= dz_channels, SHDR_E_SHAPE,
               "wgrad_f16_alltaps: bad row / column counts");
  SHDR_REQUIRE((long)d->N * d->H * d->W * (Cx > dz_channels ? Cx : dz_channels) < (1L << 32), SHDR_E_SHAPE, "wgrad_f16_alltaps: tensor too large");
  SHDR_REQUIRE(shdr::aligned16(x) && shdr::aligned16(dz), SHDR_E_ALIGN, "wgrad_f16_alltaps: tensors must be 16-byte aligned");
  WgradAArgs a{};
  a.x = reinterpret_cast<const _Float16*>(x);
  a.dz = reinterpret_cast<const _Float16*>(dz);
  a.dw = dw;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cx = Cx; a.Cz = dz_channels;
  a.Ct = c1_rows + c2_rows;
  a.ci_off = which ? c1_rows : 0;
  a.ci_valid = which ? c2_rows : c1_rows;
  a.Cout = d->Cout; a.co_valid = cout;
  a.x_scale = which ? d->x2_scale : 1.0f;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int cz = dz_channels;
  if (d->KH == 7) return launch_alltaps<7, 16, 16, 0>(a, st);
  if (d->KH == 5) {
    if (Cx == 16) return cz == 16 ? launch_alltaps<5, 16, 16, 0>(a, st) : launch_alltaps<5, 16, 32, 0>(a, st);
    return cz == 16 ? launch_alltaps<5, 32, 16, 0>(a, st) : launch_alltaps<5, 32, 32, 0>(a, st);
  }
  // 3x3
  if (Cx % 64 == 0) {
    if (cz == 64) return launch_alltaps<3, 64, 64, 1>(a, st);
    if (cz == 32) return launch_alltaps<3, 64, 32, 0>(a, st);
    return launch_alltaps<3, 64, 16, 0>(a, st);                       // dz 8 / 16 channels
  }
  if (Cx == 32) {
    if (cz == 64) return launch_alltaps<3, 32, 64, 0>(a, st);
    if (cz == 32) return launch_alltaps<3, 32, 32, 0>(a, st);
    return launch_alltaps<3, 32, 16, 0>(a, st);
  }
  if (cz == 64) return launch_alltaps<3, 16, 64, 0>(a, st);
  if (cz == 32) return launch_alltaps<3, 16, 32, 0>(a, st);
  return launch_alltaps<3, 16, 16, 0>(a, st);
}
