// Fused Winograd F(2x2, 3x3) convolution for gfx950: input transform, the 16 batched GEMMs and the output transform in
// ONE kernel, so the 4x-expanded V / M planes of the three-kernel form (winograd.hip) never touch HBM.
//
//   block  = 8 x 16 output pixels (32 Winograd tiles) x 64 output channels x all 16 transform positions xi,
//            512 threads = 8 wavefronts; wave w owns xi = {2w, 2w+1}: 2 x (32 tiles x 64 couts) fp32 accumulators.
//   K loop = 8 input channels per chunk.  Per chunk and block:
//            raw patch   10 x 18 pixels x 8 ch   HBM/L2 -> LDS by global_load_lds (zero page outside the image)
//            U slice     16 x 8 x 64             L2 -> LDS by global_load_lds (column swizzle on the source side)
//            V = B^T d B 16 x 32 x 8             LDS -> VALU -> LDS, one (tile, channel, row-pair) per thread
//            256 x v_mfma_f32_16x16x4_f32        A = V (row = tile), B = U (col = cout)
//            Raw, V and U are double buffered: the DMA of chunk c+1 / c+2 and the transform of chunk c+1 run in the
//            shadow of chunk c's MFMAs; one barrier per chunk.
//   epilogue = accumulators -> LDS (M[xi][tile][cout], overlaying the pipeline buffers) -> A^T M A per (tile, cout)
//            -> bias / activation / folded BN / activation -> 256-byte coalesced stores.
// Algorithmic intensity vs HBM is that of the direct kernel (input read once per cout slice, output written once);
// the MFMA count is 2.25x lower.  Replaces the same tf.keras.layers.Conv2D call sites as shdr_conv2d_fwd_f32
// (hallucination_net.py:43-75,115-144, vgg16.py:72-83, dequantization_net.py:35-46 for the 3x3 stride-1 layers).
#include <stdlib.h>

#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) float g_wf_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

struct WinoFusedArgs {
  const float* x;      // [N,H,W,Cin]
  const float* u;      // [16][Cin][Cout]  (winograd_filter_kernel)
  const float* bias;
  const float* scale;
  const float* shift;
  float* y;            // [N,H,W,Cout]
  int N, H, W, Cin, Cout;
  int tiles_x, tiles_y, nblk_m, nblk_n;
  int act1, act2;
};

constexpr int PW = 18;                       // raw patch: 10 rows x 18 columns
constexpr int NPIX_RAW = 10 * PW;            // 180
constexpr int RAW_FLOATS = 6 * 256;          // 6 wave DMA instructions >= 180 pixels x 8 channels
constexpr int V_FLOATS = 16 * 32 * 8;
constexpr int U_FLOATS = 16 * 8 * 64;
constexpr int M_STRIDE = 68;                 // 64 couts + 4: the 4 row groups of an accumulator tile hit disjoint banks
constexpr int PIPE_FLOATS = 2 * (RAW_FLOATS + V_FLOATS + U_FLOATS);
constexpr int EPI_FLOATS = 16 * 32 * M_STRIDE;
constexpr int LDS_BYTES = (EPI_FLOATS > PIPE_FLOATS ? EPI_FLOATS : PIPE_FLOATS) * 4;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__global__ __launch_bounds__(512) void winograd_fused_kernel(const WinoFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw = smem;                         // [2][RAW_FLOATS]   pixel-major, 8 channels per pixel
  float* Vs = smem + 2 * RAW_FLOATS;         // [2][16][32][8]
  float* Us = Vs + 2 * V_FLOATS;             // [2][16][8][64]    column swizzled: col ^ 16*((ch>>1)&3)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = xcd_remap(blockIdx.x, a.nblk_m * a.nblk_n);
  const int pn = L / a.nblk_m;               // cout slice is the SLOW index: the blocks of an XCD share their U slice in L2
  int pm = L - pn * a.nblk_m;
  const int tx = pm % a.tiles_x;
  pm /= a.tiles_x;
  const int ty = pm % a.tiles_y;
  const int img = pm / a.tiles_y;
  const int oh0 = ty * 8, ow0 = tx * 16, n0 = pn * 64;
  const float* zero = g_wf_zero_page;
  const int nch = a.Cin >> 3;

  // ---- DMA geometry ---------------------------------------------------------------------------------------------------
  // raw patch: wave w < 6 fills pixels [32w, 32w+32): lane -> (pixel, channel quad)
  bool raw_ok = false;
  unsigned raw_off = 0;
  {
    const int lr = wave * 64 + lane, pix = lr >> 1, quad = lr & 1;
    const int py = pix / PW, px = pix - py * PW;
    const int ih = oh0 - 1 + py, iw = ow0 - 1 + px;
    raw_ok = wave < 6 && pix < NPIX_RAW && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    if (raw_ok) raw_off = ((unsigned)(img * a.H + ih) * (unsigned)a.W + (unsigned)iw) * (unsigned)a.Cin + 4u * quad;
  }
  // U slice: wave w issues instructions 4w .. 4w+3; quad index Q -> (xi, ch, physical quad)
  unsigned u_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int Q = (wave * 4 + j) * 64 + lane;
    const int xi = Q >> 7, ch = (Q >> 4) & 7, pq = Q & 15;
    const int lq = pq ^ (4 * ((ch >> 1) & 3));
    u_off[j] = ((unsigned)(xi * a.Cin + ch) * (unsigned)a.Cout) + (unsigned)(n0 + 4 * lq);
  }
  auto dma_raw = [&](int c, int buf) {
    if (wave < 6) {
      const float* p = raw_ok ? a.x + (size_t)(raw_off + 8u * (unsigned)c) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(raw + buf * RAW_FLOATS + wave * 256), 16, 0, 0);
    }
  };
  auto dma_u = [&](int c, int buf) {
    const unsigned base = 8u * (unsigned)c * (unsigned)a.Cout;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(a.u + (size_t)(u_off[j] + base)),
                                       (lptr_t)(Us + buf * U_FLOATS + (wave * 4 + j) * 256), 16, 0, 0);
  };

  // ---- input transform: thread = (tile, channel, half); half 0 produces rows xi 0..7, half 1 rows 8..15 --------------
  const int t_pair = tid & 255, t_half = tid >> 8;
  const int t_tile = t_pair >> 3, t_ch = t_pair & 7;
  const int t_src = ((2 * (t_tile >> 3) + t_half) * PW + 2 * (t_tile & 7)) * 8 + t_ch;   // first patch row this half reads
  const int t_dst = t_tile * 8 + t_ch;
  auto transform = [&](int rbuf, int vbuf) {
    const float* rp = raw + rbuf * RAW_FLOATS + t_src;
    float d[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j] = rp[(i * PW + j) * 8];
    float r0[4], r1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (t_half == 0) {                      // patch rows 0,1,2:  B^T rows 0, 1
        r0[j] = d[0][j] - d[2][j];
        r1[j] = d[1][j] + d[2][j];
      } else {                                // patch rows 1,2,3:  B^T rows 2, 3
        r0[j] = d[1][j] - d[0][j];
        r1[j] = d[0][j] - d[2][j];
      }
    }
    float* vp = Vs + vbuf * V_FLOATS + t_half * (8 * 256) + t_dst;   // xi = 8*half + 4*row + col, plane stride 32*8
    vp[0 * 256] = r0[0] - r0[2];
    vp[1 * 256] = r0[1] + r0[2];
    vp[2 * 256] = r0[2] - r0[1];
    vp[3 * 256] = r0[1] - r0[3];
    vp[4 * 256] = r1[0] - r1[2];
    vp[5 * 256] = r1[1] + r1[2];
    vp[6 * 256] = r1[2] - r1[1];
    vp[7 * 256] = r1[1] - r1[3];
  };

  // ---- MFMA geometry --------------------------------------------------------------------------------------------------
  const int fi = lane & 15, fg = lane >> 4;
  const int xi0 = 2 * wave;
  const int a_off = (xi0 * 32 + fi) * 8 + 2 * fg;                    // + x2*256 + mt*128
  int b_off[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) b_off[nt] = (xi0 * 8 + 2 * fg) * 64 + ((nt * 16 + fi) ^ (16 * fg));   // + x2*512 + s*64

  f32x4 acc[2][2][4];
#pragma unroll
  for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[x2][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  struct Frags {
    float2 a0, a1;
    float b[2][4];
  };
  auto load_frags = [&](int buf, int x2, Frags& f) {
    const float* vb = Vs + buf * V_FLOATS + a_off + x2 * 256;
    const float* ub = Us + buf * U_FLOATS + x2 * 512;
    f.a0 = *reinterpret_cast<const float2*>(vb);
    f.a1 = *reinterpret_cast<const float2*>(vb + 128);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) f.b[s][nt] = ub[b_off[nt] + s * 64];
  };
  auto mfma_half = [&](int x2, const Frags& f) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      acc[x2][0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a0.x, f.b[0][nt], acc[x2][0][nt], 0, 0, 0);
      acc[x2][1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1.x, f.b[0][nt], acc[x2][1][nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      acc[x2][0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a0.y, f.b[1][nt], acc[x2][0][nt], 0, 0, 0);
      acc[x2][1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1.y, f.b[1][nt], acc[x2][1][nt], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- pipeline -------------------------------------------------------------------------------------------------------
  dma_raw(0, 0);
  dma_u(0, 0);
  if (nch > 1) dma_raw(1, 1);
  __syncthreads();
  transform(0, 0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    const int b = c & 1;
    // every LDS read of V(c) / U(c) is issued BEFORE this chunk's DMAs: the compiler orders an LDS read behind all
    // LDS-DMA writes in flight (s_waitcnt vmcnt(0)) when it cannot tell the buffers apart
    Frags f0, f1;
    load_frags(b, 0, f0);
    load_frags(b, 1, f1);
    if (c + 1 < nch) dma_u(c + 1, b ^ 1);
    if (c + 2 < nch) dma_raw(c + 2, b);
    mfma_half(0, f0);
    if (c + 1 < nch) transform(b ^ 1, b ^ 1);           // raw(c+1) lives in raw[(c+1)&1]
    mfma_half(1, f1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> M[xi][tile][cout] in LDS (the pipeline buffers are dead) ----------------------------
  float* Ms = smem;
#pragma unroll
  for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Ms[((xi0 + x2) * 32 + mt * 16 + 4 * fg + r) * M_STRIDE + nt * 16 + fi] = acc[x2][mt][nt][r];
  __syncthreads();

  const int co = tid & 63, txx = tid >> 6;               // thread = (cout, tile column); loops over the 4 tile rows
  const float bv = a.bias ? a.bias[n0 + co] : 0.0f;
  const float sc = a.scale ? a.scale[n0 + co] : 1.0f;
  const float sh = a.scale ? a.shift[n0 + co] : 0.0f;
#pragma unroll 1
  for (int tyy = 0; tyy < 4; ++tyy) {
    const float* mp = Ms + (tyy * 8 + txx) * M_STRIDE + co;
    float s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float m0 = mp[(0 * 4 + j) * 32 * M_STRIDE], m1 = mp[(1 * 4 + j) * 32 * M_STRIDE];
      const float m2 = mp[(2 * 4 + j) * 32 * M_STRIDE], m3 = mp[(3 * 4 + j) * 32 * M_STRIDE];
      s[0][j] = m0 + m1 + m2;
      s[1][j] = m1 - m2 - m3;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oh = oh0 + 2 * tyy + i;
      if (oh >= a.H) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ow = ow0 + 2 * txx + j;
        if (ow >= a.W) continue;
        float v = (j == 0) ? s[i][0] + s[i][1] + s[i][2] : s[i][1] - s[i][2] - s[i][3];
        v = shdr::act_apply(v + bv, a.act1);
        if (a.scale) v = v * sc + sh;
        v = shdr::act_apply(v, a.act2);
        a.y[((size_t)(img * a.H + oh) * a.W + ow) * a.Cout + n0 + co] = v;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// v2: no V staging.  The A operand of v_mfma_f32_16x16x4_f32 wants, per lane, V[xi][tile = lane & 15][channel = lane >> 4 ...]
// -- exactly one lane of the one wave that owns xi.  So every lane builds its own operand values straight from the raw
// patch in LDS (6 ds_read_b64 + 10 VALU per 16-tile group and chunk: xi = (row combination i, column combination j) needs
// 2 patch rows x 3 patch columns), and the U slice of a wave's two xi is private to that wave.  What is left to share
// is the raw patch: 78 KB of LDS per block instead of 139 KB -> TWO blocks (16 wavefronts) per CU, one barrier per chunk
// that only orders the raw-patch DMA.  The epilogue runs in two passes over 32-cout halves to stay inside that LDS.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int M2_STRIDE = 36;
// raw patch image of v2: 16-byte slots [channel quad][patch row (pitch 20)][patch column + (tile-row parity)].  A lane's
// ds_read_b64 then takes one slot per (tile column, tile-row parity) = 16 different slots of 16 bytes per half-wave:
// conflict-free (the pixel-major image of v1 lands 4 lanes on every bank pair: measured LDS-bound)
constexpr int RP2 = 20, QS2 = 10 * RP2;          // slots per patch row / per channel quad
constexpr int RAW2_FLOATS = 7 * 256;             // 7 wave DMA instructions >= 2 * 200 slots
constexpr int PIPE2_FLOATS = 2 * (RAW2_FLOATS + U_FLOATS);
constexpr int EPI2_FLOATS = 16 * 32 * M2_STRIDE;
constexpr int LDS2_BYTES = (EPI2_FLOATS > PIPE2_FLOATS ? EPI2_FLOATS : PIPE2_FLOATS) * 4;

__global__ __launch_bounds__(512, 4) void winograd_fused2_kernel(const WinoFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw = smem;                         // [2][RAW2_FLOATS]
  float* Us = smem + 2 * RAW2_FLOATS;        // [2][8 waves][2 xi][8 ch][64 co], column swizzled

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = xcd_remap(blockIdx.x, a.nblk_m * a.nblk_n);
  const int pn = L / a.nblk_m;
  int pm = L - pn * a.nblk_m;
  const int tx = pm % a.tiles_x;
  pm /= a.tiles_x;
  const int ty = pm % a.tiles_y;
  const int img = pm / a.tiles_y;
  const int oh0 = ty * 8, ow0 = tx * 16, n0 = pn * 64;
  const float* zero = g_wf_zero_page;
  const int nch = a.Cin >> 3;
  const int xi0 = 2 * wave;

  // ---- DMA geometry ---------------------------------------------------------------------------------------------------
  bool raw_ok = false;
  unsigned raw_off = 0;
  {
    const int slot = wave * 64 + lane;       // waves 0..6
    const int quad = slot / QS2, rem = slot - quad * QS2;
    const int py = rem / RP2, px = rem - py * RP2 - ((py >> 1) & 1);
    const int ih = oh0 - 1 + py, iw = ow0 - 1 + px;
    raw_ok = wave < 7 && quad < 2 && px >= 0 && px < PW && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    if (raw_ok) raw_off = ((unsigned)(img * a.H + ih) * (unsigned)a.W + (unsigned)iw) * (unsigned)a.Cin + 4u * quad;
  }
  unsigned u_off[4];                          // this wave's own two xi planes
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int Q = j * 64 + lane;
    const int x2 = Q >> 7, ch = (Q >> 4) & 7, pq = Q & 15;
    const int lq = pq ^ (4 * ((ch >> 1) & 3));
    u_off[j] = ((unsigned)((xi0 + x2) * a.Cin + ch) * (unsigned)a.Cout) + (unsigned)(n0 + 4 * lq);
  }
  auto dma_chunk = [&](int c, int buf) {
    const unsigned base = 8u * (unsigned)c * (unsigned)a.Cout;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(a.u + (size_t)(u_off[j] + base)),
                                       (lptr_t)(Us + buf * U_FLOATS + wave * 1024 + j * 256), 16, 0, 0);
    if (wave < 7) {
      const float* p = raw_ok ? a.x + (size_t)(raw_off + 8u * (unsigned)c) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(raw + buf * RAW2_FLOATS + wave * 256), 16, 0, 0);
    }
  };

  // ---- operand geometry -----------------------------------------------------------------------------------------------
  const int fi = lane & 15, fg = lane >> 4;
  const int wi = wave >> 1, jp = wave & 1;    // B^T row combination i; column pair: jp = 0 -> j in {0,1}, 1 -> {2,3}
  // V row i = d[ra] + sr * d[rb]:  i=0: d0-d2, 1: d1+d2, 2: d2-d1, 3: d1-d3
  const int ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1);
  const int rb = wi == 3 ? 3 : (wi == 2 ? 1 : 2);
  const float sr = wi == 1 ? 1.0f : -1.0f;
  int a_addr[2][2];                           // [mt][row a / row b]: float offset of patch column 2*txx + jp, channels 2fg..2fg+1
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int tyy = 2 * mt + (fi >> 3), txx = fi & 7;
    const int pa = 2 * tyy + ra, pb = 2 * tyy + rb;
    a_addr[mt][0] = ((fg >> 1) * QS2 + pa * RP2 + 2 * txx + jp + ((pa >> 1) & 1)) * 4 + 2 * (fg & 1);
    a_addr[mt][1] = ((fg >> 1) * QS2 + pb * RP2 + 2 * txx + jp + ((pb >> 1) & 1)) * 4 + 2 * (fg & 1);
  }
  int b_off[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) b_off[nt] = wave * 1024 + (2 * fg) * 64 + ((nt * 16 + fi) ^ (16 * fg));   // + x2*512 + s*64

  f32x4 acc[2][2][4];
#pragma unroll
  for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[x2][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  dma_chunk(0, 0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    const int b = c & 1;
    // every LDS read of this chunk is issued before the next chunk's DMAs (see v1)
    const float* rp = raw + b * RAW2_FLOATS;
    const float* ub = Us + b * U_FLOATS;
    float2 d[2][2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int k = 0; k < 3; ++k) d[mt][rr][k] = *reinterpret_cast<const float2*>(rp + a_addr[mt][rr] + k * 4);
    float bq[2][2][4];
#pragma unroll
    for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bq[x2][s][nt] = ub[b_off[nt] + x2 * 512 + s * 64];
    if (c + 1 < nch) dma_chunk(c + 1, b ^ 1);

    float2 v[2][2];                           // [x2][mt]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float2 r[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        r[k].x = d[mt][0][k].x + sr * d[mt][1][k].x;
        r[k].y = d[mt][0][k].y + sr * d[mt][1][k].y;
      }
      if (jp == 0) {                          // columns 0,1,2: j=0: c0-c2, j=1: c1+c2
        v[0][mt] = make_float2(r[0].x - r[2].x, r[0].y - r[2].y);
        v[1][mt] = make_float2(r[1].x + r[2].x, r[1].y + r[2].y);
      } else {                                // columns 1,2,3: j=2: c2-c1, j=3: c1-c3
        v[0][mt] = make_float2(r[1].x - r[0].x, r[1].y - r[0].y);
        v[1][mt] = make_float2(r[0].x - r[2].x, r[0].y - r[2].y);
      }
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int x2 = 0; x2 < 2; ++x2) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        acc[x2][0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[x2][0].x, bq[x2][0][nt], acc[x2][0][nt], 0, 0, 0);
        acc[x2][1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[x2][1].x, bq[x2][0][nt], acc[x2][1][nt], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        acc[x2][0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[x2][0].y, bq[x2][1][nt], acc[x2][0][nt], 0, 0, 0);
        acc[x2][1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[x2][1].y, bq[x2][1][nt], acc[x2][1][nt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  // ---- epilogue in two 32-cout passes ---------------------------------------------------------------------------------
  float* Ms = smem;
  const int co = tid & 31, tg = tid >> 5;    // thread = (cout of the half, tile group); tiles tg and tg + 16
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 t = h == 0 ? acc[x2][mt][n2] : acc[x2][mt][2 + n2];
            Ms[((xi0 + x2) * 32 + mt * 16 + 4 * fg + r) * M2_STRIDE + n2 * 16 + fi] = t[r];
          }
    __syncthreads();
    const int cg = n0 + 32 * h + co;
    const float bv = a.bias ? a.bias[cg] : 0.0f;
    const float sc = a.scale ? a.scale[cg] : 1.0f;
    const float sh = a.scale ? a.shift[cg] : 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int tile = tg + 16 * k, tyy = tile >> 3, txx = tile & 7;
      const float* mp = Ms + tile * M2_STRIDE + co;
      float s[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m0 = mp[(0 * 4 + j) * 32 * M2_STRIDE], m1 = mp[(1 * 4 + j) * 32 * M2_STRIDE];
        const float m2 = mp[(2 * 4 + j) * 32 * M2_STRIDE], m3 = mp[(3 * 4 + j) * 32 * M2_STRIDE];
        s[0][j] = m0 + m1 + m2;
        s[1][j] = m1 - m2 - m3;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int oh = oh0 + 2 * tyy + i;
        if (oh >= a.H) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ow = ow0 + 2 * txx + j;
          if (ow >= a.W) continue;
          float v = (j == 0) ? s[i][0] + s[i][1] + s[i][2] : s[i][1] - s[i][2] - s[i][3];
          v = shdr::act_apply(v + bv, a.act1);
          if (a.scale) v = v * sc + sh;
          v = shdr::act_apply(v, a.act2);
          a.y[((size_t)(img * a.H + oh) * a.W + ow) * a.Cout + cg] = v;
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int shdr_conv2d_winograd_fused_f32(const float* x, const float* u, const float* bias, const float* scale,
                                              const float* shift, float* y, int N, int H, int W, int Cin, int Cout,
                                              int act1, int act2, void* stream) {
  SHDR_REQUIRE(x && u && y, SHDR_E_NULL, "winograd_fused: null x/u/y");
  SHDR_REQUIRE((scale == nullptr) == (shift == nullptr), SHDR_E_NULL, "winograd_fused: scale and shift come together");
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, SHDR_E_SHAPE, "winograd_fused: non-positive dimension");
  SHDR_REQUIRE(Cin % 8 == 0 && Cout % 64 == 0, SHDR_E_SHAPE, "winograd_fused: need Cin %% 8 == 0 and Cout %% 64 == 0");
  SHDR_REQUIRE(shdr::aligned16(x) && shdr::aligned16(u), SHDR_E_ALIGN, "winograd_fused: x and u must be 16-byte aligned");
  SHDR_REQUIRE((long)N * H * W * Cin < (1L << 32) && 16L * Cin * Cout < (1L << 32), SHDR_E_SHAPE,
               "winograd_fused: tensor with more than 2^32 elements");
  WinoFusedArgs a{};
  a.x = x; a.u = u; a.bias = bias; a.scale = scale; a.shift = shift; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.tiles_x = (W + 15) / 16;
  a.tiles_y = (H + 7) / 8;
  a.nblk_m = N * a.tiles_y * a.tiles_x;
  a.nblk_n = Cout / 64;
  a.act1 = act1; a.act2 = act2;
  const long nblk = (long)a.nblk_m * a.nblk_n;
  SHDR_REQUIRE(nblk > 0 && nblk <= 0x7fffffffL, SHDR_E_SHAPE, "winograd_fused: grid of %ld blocks", nblk);
  static bool attr_done = false;
  static int variant = 2;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd_fused2_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    const char* v = getenv("SHDR_WINOGRAD_FUSED_VARIANT");     // 1: V staged through LDS, 2 (default): operands built per lane
    if (v && v[0] == '1') variant = 1;
    attr_done = true;
  }
  if (variant == 1)
    hipLaunchKernelGGL(winograd_fused_kernel, dim3((unsigned)nblk), dim3(512), LDS_BYTES, reinterpret_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(winograd_fused2_kernel, dim3((unsigned)nblk), dim3(512), LDS2_BYTES, reinterpret_cast<hipStream_t>(stream), a);
  return shdr::check_launch("winograd_fused_kernel");
}
