// Fused Winograd F(2x2, 3x3) convolution for gfx950: input transform, the 16 batched GEMMs and the output transform in
// ONE kernel, so the 4x-expanded V / M planes of the three-kernel form (winograd.hip) never touch HBM.
//
//   block  = 8 x 16 output pixels (32 Winograd tiles) x 64 output channels x all 16 transform positions xi,
//            512 threads = 8 wavefronts, two blocks per CU; wave w owns xi = {2w, 2w+1}: 2 x (32 tiles x 64 couts) fp32
//            accumulators = 64 registers.
//   K loop = 8 input channels per chunk.  Per chunk and block:
//            raw patch  10 x 18 pixels x 8 ch   HBM/L2 -> LDS by global_load_lds (zero page outside the image), double buffered
//            filter     16 x 8 x 64             L2 -> VGPRs: the 16 B-operand values of a lane are four coalesced float4 loads
//                                               from the packed layout of winograd_filter_packed_kernel -- no LDS, no DMA; a
//                                               register group is refilled for chunk c+1 right after the MFMAs that read it
//            256 x v_mfma_f32_16x16x4_f32       A = V (row = tile), B = U (col = cout)
//   There is NO V staging: the A operand wants V[xi][tile = lane & 15][channel pair = lane >> 4] in exactly one lane of
//   the one wave that owns xi, so every lane builds its own operand values straight from the raw patch in LDS --
//   xi = (row combination i, column combination j) needs 2 patch rows x 3 patch columns: 6 ds_read_b64 + 10 VALU per
//   16-tile group and chunk.  The only data the waves share is the raw patch: one barrier per chunk orders its DMA.
//   All LDS reads of a chunk are issued before the DMAs of the next one: the compiler orders an LDS read behind every
//   LDS-DMA write in flight (s_waitcnt vmcnt(0)) when it cannot tell the buffers apart (measured: 2x).
//   Raw patch image: 16-byte slots [channel quad][patch row, pitch 20][patch column + tile-row parity], so the 16
//   (tile column, tile-row parity) lanes of a half-wave read 16 different slots: conflict-free (the pixel-major image
//   put 4 lanes on every bank pair and made the kernel LDS-bound: 0.50 -> 0.58..0.64 of the MFMA peak).
//   epilogue = two passes over 32-cout halves: accumulators -> LDS (M[xi][tile][cout], overlaying the pipeline buffers)
//            -> A^T M A per (tile, cout) -> bias / activation / folded BN / activation -> 128-byte coalesced stores.
// HBM traffic is that of the direct kernel (input read once per cout slice, output written once); the MFMA count is 2.25x
// lower.  Ablation on 16 x 64^2 x 512 -> 512 (1.38 ms, 0.64 of the fp32-MFMA peak in executed FLOPs, 224 TFLOP/s
// algorithmic): MFMA + VALU alone 1.02 ms, + LDS reads 1.17 ms, + DMA 1.45 ms, + barrier 1.51 ms before the image fix.
// Replaces the same tf.keras.layers.Conv2D call sites as shdr_conv2d_fwd_f32 (hallucination_net.py:43-75,115-144,
// vgg16.py:72-83, dequantization_net.py:35-46 for the 3x3 stride-1 layers).
#include <stdlib.h>

#include <type_traits>

#include "shdr_internal.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) float g_wf_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

struct WinoFusedArgs {
  const float* x;      // [N,H,W,C1]
  const float* x2;     // [N,H,W,C1] second source of a channel concatenation (or null); the channels of x come first
  const float* u;      // packed U (winograd_filter_packed_kernel)
  const float* bias;
  const float* scale;
  const float* shift;
  float* y;            // [N,H,W,Cout]
  float* yp;           // [N,H/2,W/2,Cout] 2x2 max-pool of y (or null): a Winograd tile is exactly one pooling window
  int N, H, W, Cin, Cout;   // Cin = all input channels (both sources)
  int C1;                   // channels per source (= Cin, or Cin / 2 with x2)
  int tiles_x, tiles_y, nblk_m, nblk_n;
  int act1, act2;
  int Hl, Wl;               // UP variant: x is the LOW-RES tensor [N,Hl,Wl,C1], H = 2 Hl, W = 2 Wl are the dims of the up-sampled conv input
};

constexpr int PW = 18;                           // raw patch columns (16 + 2)
constexpr int RP2 = 20;                          // slots per patch row
constexpr int M2_STRIDE = 36;                    // 32 couts + 4: the 4 row groups of an accumulator tile hit disjoint banks

// R = block height in units of 8 output rows: R = 1 -> 8 x 16 pixels, 32 tiles, 64 accumulator registers, two blocks per CU;
// R = 2 -> 16 x 16 pixels, 64 tiles, 128 accumulator registers, one block per CU and HALF the filter bytes and B-operand
// reads per MFMA (for layers with enough tiles to fill the chip that way).
template <int R>
struct WF {
  static constexpr int MT = 2 * R;                         // 16-tile groups
  static constexpr int PROWS = 8 * R + 2;                  // raw patch rows
  static constexpr int QS = PROWS * RP2;                   // slots per channel quad
  static constexpr int RAW_INSTR = ((2 * QS + 63) / 64 + 7) / 8 * 8;   // wave DMA instructions per raw patch, the same
                                                                       // number for every wave (uniform vmcnt)
  static constexpr int RAW_FLOATS = RAW_INSTR * 256;
  static constexpr int PIPE_FLOATS = 4 * RAW_FLOATS;       // four raw-patch buffers: chunk c+1 is read while c+3 is in flight
  static constexpr int EPI_FLOATS = 16 * 32 * R * M2_STRIDE;
  static constexpr int LDS_BYTES = (EPI_FLOATS > PIPE_FLOATS ? EPI_FLOATS : PIPE_FLOATS) * 4;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// UP = true (R = 1): tf.image.resize(x, 2x, BILINEAR) fused in front of the convolution (hallucination_net.py:86,
// dequantization_net.py:25): the block stages the LOW-RES patch (6 x 10 pixels per 8-channel chunk, a third of the bytes) by
// LDS-DMA into a ring of five buffers and expands it to the 10 x 18 raw patch in LDS itself -- same arithmetic, same rounding as
// resize2x_kernel (pool.hip: horizontal lerp, then vertical, clamped taps, zeros outside the up-sampled image) -- one chunk
// ahead of the transform that reads it.  The up-sampled tensor never exists in HBM.
template <int R, bool UP = false>
__global__ __launch_bounds__(512, (R == 1 ? 4 : 2)) void winograd_fused_kernel(const WinoFusedArgs a) {
  using G = WF<R>;
  constexpr int MT = G::MT;
  static_assert(!UP || R == 1, "the up-sampling prologue is built for the 8 x 16 tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw = smem;                         // [4][RAW_FLOATS]
  float* lrb = smem + G::PIPE_FLOATS;        // UP: [5][LR_FLOATS] low-res patches (inside the epilogue overlay's footprint)
  constexpr int LR_FLOATS = 8 * 256;         // one 1 KiB DMA instruction per wave (waves 0, 1 carry the 120 slots, the others a zero page)
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)raw;
  const unsigned lds_lr = (unsigned)(unsigned long)(lptr_t)lrb;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = xcd_remap(blockIdx.x, a.nblk_m * a.nblk_n);
  const int pn = L / a.nblk_m;
  int pm = L - pn * a.nblk_m;
  const int tx = pm % a.tiles_x;
  pm /= a.tiles_x;
  const int ty = pm % a.tiles_y;
  const int img = pm / a.tiles_y;
  const int oh0 = ty * 8 * R, ow0 = tx * 16, n0 = pn * 64;
  const float* zero = g_wf_zero_page;
  const int nch = a.Cin >> 3;
  const int xi0 = 2 * wave;

  // ---- DMA geometry ---------------------------------------------------------------------------------------------------
  constexpr int RJ = (G::RAW_INSTR + 7) / 8;  // raw DMA instructions per wave (instruction index = wave + 8*j)
  bool raw_ok[RJ];
  unsigned raw_off[RJ];
#pragma unroll
  for (int j = 0; j < RJ; ++j) {
    const int slot = (wave + 8 * j) * 64 + lane;
    const int quad = slot / G::QS, rem = slot - quad * G::QS;
    const int py = rem / RP2, px = rem - py * RP2 - ((py >> 1) & 1);
    const int ih = oh0 - 1 + py, iw = ow0 - 1 + px;
    raw_ok[j] = quad < 2 && px >= 0 && px < PW && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    raw_off[j] = raw_ok[j] ? ((unsigned)(img * a.H + ih) * (unsigned)a.W + (unsigned)iw) * (unsigned)a.C1 + 4u * quad : 0u;
  }
  // UP: low-res patch rows oh0/2 - 1 .. oh0/2 + 4, columns ow0/2 - 1 .. ow0/2 + 8 as [quad][6][10] slots of 16 bytes
  unsigned lr_off = 0xFFFFFFFFu;              // sentinel: this lane stages zeros
  if (UP) {
    const int s = wave * 64 + lane;
    const int quad = s / 60, rem = s - quad * 60;
    const int ly = rem / 10, lx = rem - ly * 10;
    const int r = (oh0 >> 1) - 1 + ly, c = (ow0 >> 1) - 1 + lx;
    const bool ok = wave < 2 && s < 120 && (unsigned)r < (unsigned)a.Hl && (unsigned)c < (unsigned)a.Wl;
    if (ok) lr_off = ((unsigned)(img * a.Hl + r) * (unsigned)a.Wl + (unsigned)c) * (unsigned)a.C1 + 4u * quad;
  }
  // UP: expansion geometry, thread t < 360 owns raw slot (quad, py, px).  One packed register (the kernel sits at the 128-VGPR
  // limit of two blocks per CU): bits 0..12 byte offset of the top-left low-res slot, 13 right tap one slot on, 14 bottom tap
  // one row on, 15 wx = 0.25 (else 0.75), 16 wy = 0.25, 17 pixel outside the image (zeros), 18 thread takes part,
  // 19..31 byte offset of the raw slot
  unsigned ex_pack = 0u;
  if (UP) {
    const bool on = tid < 360;
    const int t = on ? tid : 0;
    const int quad = t / 180, rem = t - quad * 180;
    const int py = rem / PW, px = rem - py * PW;
    const int rf = oh0 - 1 + py, cf = ow0 - 1 + px;            // pixel of the up-sampled image
    const bool outside = !((unsigned)rf < (unsigned)a.H && (unsigned)cf < (unsigned)a.W);
    const int rfc = outside ? oh0 : rf, cfc = outside ? ow0 : cf;
    const int mr = rfc >> 1, mc = cfc >> 1;
    const int ra = (rfc & 1) ? mr : max(mr - 1, 0), rb = (rfc & 1) ? min(mr + 1, a.Hl - 1) : mr;
    const int ca = (cfc & 1) ? mc : max(mc - 1, 0), cb = (cfc & 1) ? min(mc + 1, a.Wl - 1) : mc;
    const int r0 = (oh0 >> 1) - 1, c0 = (ow0 >> 1) - 1;
    const unsigned src = (unsigned)((quad * 60 + (ra - r0) * 10 + (ca - c0)) * 16);
    const unsigned dst = (unsigned)((quad * G::QS + py * RP2 + px + ((py >> 1) & 1)) * 16);
    ex_pack = src | ((unsigned)(cb != ca) << 13) | ((unsigned)(rb != ra) << 14) | ((unsigned)(cfc & 1) << 15) |
              ((unsigned)(rfc & 1) << 16) | ((unsigned)outside << 17) | ((unsigned)on << 18) | (dst << 19);
  }
  // B operands: this lane's 16 filter values of a chunk = four float4 at up[(c*4 + j)*256], j = 2*x2 + s, .xyzw = nt 0..3
  const float* ub = a.u + (size_t)(pn * 8 + wave) * nch * 1024;     // wave-uniform: SGPR base, the lane offset stays 32-bit
  // chunk c of the concatenated channel axis: both sources have C1 channels, so the pixel offsets are shared (wave-uniform
  // pointer select)
  const int nch1 = a.C1 >> 3;
  auto dma_lr = [&](int c, int buf) {           // UP: one instruction per wave, like the raw DMA of the 8 x 16 tile (same vmcnt counts)
    const float* p = lr_off != 0xFFFFFFFFu ? a.x + (size_t)(lr_off + 8u * (unsigned)c) : zero;
    __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lrb + buf * LR_FLOATS + wave * 256), 16, 0, 0);
  };
  // UP: low-res(buf_lr) -> raw(buf_raw).  LDS accesses in inline asm: the compiler would order its own LDS reads behind every
  // LDS-DMA write in flight (vmcnt(0)); the data was waited for and barrier-ordered an iteration ago.
  auto expand = [&](int buf_lr, int buf_raw) {
    unsigned pk = ex_pack;
    asm volatile("" : "+v"(pk));              // unpack HERE: hoisted out of the loop the unpacked values spill (and a spill reload is a vmcnt(0))
    if (pk & (1u << 18)) {
      const unsigned s00 = lds_lr + (unsigned)(buf_lr * LR_FLOATS * 4) + (pk & 0x1FFFu);
      const unsigned dx = (pk >> 9) & 16u, dy = ((pk >> 14) & 1u) * 160u;
      const float wx = (pk & (1u << 15)) ? 0.25f : 0.75f, wy = (pk & (1u << 16)) ? 0.25f : 0.75f;
      f32x4 q00, q01, q10, q11;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(q00), "=&v"(q01), "=&v"(q10), "=&v"(q11)
                   : "v"(s00), "v"(s00 + dx), "v"(s00 + dy), "v"(s00 + dx + dy));
      const bool outside = pk & (1u << 17);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = q00[e] + (q01[e] - q00[e]) * wx;        // horizontal first, then vertical: the order of resize2x_kernel
        const float u = q10[e] + (q11[e] - q10[e]) * wx;
        o[e] = outside ? 0.0f : t + (u - t) * wy;
      }
      const unsigned db = lds0 + (unsigned)(buf_raw * G::RAW_FLOATS * 4) + (pk >> 19);
      asm volatile("ds_write_b128 %0, %1" ::"v"(db), "v"(o) : "memory");
    }
  };
  auto dma_raw = [&](int c, int buf) {
    const float* src = c < nch1 ? a.x : a.x2;
    const unsigned cc = (unsigned)(c < nch1 ? c : c - nch1);
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
      const float* p = raw_ok[j] ? src + (size_t)(raw_off[j] + 8u * cc) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(raw + buf * G::RAW_FLOATS + (wave + 8 * j) * 256), 16, 0, 0);
    }
  };

  // ---- operand geometry -----------------------------------------------------------------------------------------------
  const int fi = lane & 15, fg = lane >> 4;
  const int wi = wave >> 1, jp = wave & 1;    // B^T row combination i; column pair: jp = 0 -> j in {0,1}, 1 -> {2,3}
  // V row i = d[ra] + sr * d[rb]:  i=0: d0-d2, 1: d1+d2, 2: d2-d1, 3: d1-d3
  const int ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1);
  const int rb = wi == 3 ? 3 : (wi == 2 ? 1 : 2);
  const float sr = wi == 1 ? 1.0f : -1.0f;
  int a_addr[MT][2];                          // [mt][row a / row b]: float offset of patch column 2*txx + jp, channels 2fg..2fg+1
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int tyy = 2 * mt + (fi >> 3), txx = fi & 7;
    const int pa = 2 * tyy + ra, pb = 2 * tyy + rb;
    a_addr[mt][0] = ((fg >> 1) * G::QS + pa * RP2 + 2 * txx + jp + ((pa >> 1) & 1)) * 4 + 2 * (fg & 1);
    a_addr[mt][1] = ((fg >> 1) * G::QS + pb * RP2 + 2 * txx + jp + ((pb >> 1) & 1)) * 4 + 2 * (fg & 1);
  }
  f32x4 acc[2][MT][4];
#pragma unroll
  for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[x2][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Software pipeline over the 8-channel chunks (raw patch in 4 LDS buffers, filter operands and V in registers):
  //   iteration c:  LDS reads of raw(c+1) -> issue DMA raw(c+3) -> MFMAs of chunk c (operands v(c), bq = U(c), both already in
  //                 registers; each bq half is refilled with U(c+1) right after the MFMAs that read it) -> V(c+1) from the
  //                 values read at the top (their LDS latency sat under the MFMAs) -> barrier.
  // Every LDS read precedes the DMA issue of its iteration (see the header).  Counted waits: "s_waitcnt vmcnt(RJ)" before the
  // MFMAs = everything but this iteration's RJ raw DMAs has arrived (the filter registers AND raw(c+2), so the barrier at the
  // end needs no wait of its own and the DMA of raw(c+3) plus the 4 filter loads stay in flight across it).
  // The filter loads are inline asm as well: the compiler's own wait for a register loaded in the PREVIOUS iteration is
  // vmcnt(0) (its loop-carried scoreboard is conservative), which would drain raw(c+3) in front of the first MFMA group.
  // Untracked, the only waits are the counted ones written below.
  f32x4 bq[4];
  const unsigned ulb = 16u * (unsigned)lane;
#define WF_LOAD_BQ(J, BASE) \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #J "*1024" : "=v"(bq[J]) : "v"(ulb), "s"(BASE))
  WF_LOAD_BQ(0, ub); WF_LOAD_BQ(1, ub); WF_LOAD_BQ(2, ub); WF_LOAD_BQ(3, ub);
  if (UP) {
    // low-res patches run TWO chunks ahead of the raw patches they expand into (a DMA issued in iteration k is landed and
    // barrier-ordered for every wave from iteration k + 2 on): lr(c + 4) is issued where raw(c + 3) used to be
    dma_lr(0, 0);
    dma_lr(nch > 1 ? 1 : 0, 1);
    dma_lr(nch > 2 ? 2 : nch - 1, 2);
    dma_lr(nch > 3 ? 3 : nch - 1, 3);
    __syncthreads();
    expand(0, 0);
    expand(1, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  } else {
    dma_raw(0, 0);
    dma_raw(nch > 1 ? 1 : 0, 1);
    dma_raw(nch > 2 ? 2 : nch - 1, 2);
    __syncthreads();
  }

  f32x2 d[MT][2][3];
  float2 v[2][MT];                            // [x2][mt]
  // The patch reads are inline asm: the compiler orders every LDS read it knows of behind ALL LDS-DMA writes in flight
  // (s_waitcnt vmcnt(0) at the top of the loop -- it cannot tell the four buffers apart), which drained the raw DMA and
  // the filter reloads once per chunk.  Hidden in asm the reads cost no vmcnt; their own completion is the explicit
  // "s_waitcnt lgkmcnt(0)" + register fence in patch_ready() before the transform.
  auto read_patch = [&](int buf) {
    const unsigned base = lds0 + (unsigned)(buf * G::RAW_FLOATS * 4);       // LDS byte address of the buffer
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const unsigned ad = base + (unsigned)a_addr[mt][rr] * 4u;
        asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:16\n\tds_read_b64 %2, %3 offset:32"
                     : "=&v"(d[mt][rr][0]), "=&v"(d[mt][rr][1]), "=&v"(d[mt][rr][2]) : "v"(ad));
      }
  };
  auto patch_ready = [&]() {
    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0) only
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
        asm volatile("" : "+v"(d[mt][rr][0]), "+v"(d[mt][rr][1]), "+v"(d[mt][rr][2]));
  };
  auto transform = [&]() {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
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
  };
  read_patch(0);
  patch_ready();
  transform();
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    read_patch((c + 1) & 3);                  // raw(c+1): landed and barrier-ordered one iteration ago (unused after the last chunk)
    __builtin_amdgcn_sched_barrier(0);
    if (UP) dma_lr(c + 4 < nch ? c + 4 : nch - 1, (c + 4) % 5);
    else dma_raw(c + 3 < nch ? c + 3 : nch - 1, (c + 3) & 3);      // unconditional: a uniform vmcnt
    __builtin_amdgcn_sched_barrier(0);
    const float* un = ub + (size_t)(c + 1 < nch ? c + 1 : c) * 1024;       // next chunk's operands (the last chunk reloads itself)
    // Four groups of 2*MT*... MFMAs, one per filter register quad j = 2*x2 + k-step.  In flight before group j, oldest first:
    //   [raw(c+2)] [bq j] [the three quads after j: reloads of this or the previous iteration] [raw(c+3)]
    // so "vmcnt(RJ + 3)" = quad j (and raw(c+2)) has landed, everything newer stays in flight.  The MFMA builtin is a pure
    // function of its registers and instruction selection floats it over every chained instruction (all 32 MFMAs ended up
    // above the waits and reloads: the reloads then sat right in front of the barrier and their L2 latency was exposed in
    // every chunk); the empty asm statements tie the group to its place: operands are "produced" after the wait, accumulators
    // are "consumed" before the reload of the quad the group has just read.
    auto group = [&](auto jc) {
      constexpr int j = decltype(jc)::value, x2 = j >> 1;
      __builtin_amdgcn_s_waitcnt(0x0F70 | (RJ + 3));
      asm volatile("" : "+v"(bq[j]));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[x2][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32((j & 1) ? v[x2][mt].y : v[x2][mt].x, bq[j][nt], acc[x2][mt][nt], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        asm volatile("" : "+v"(acc[x2][mt][0]), "+v"(acc[x2][mt][1]), "+v"(acc[x2][mt][2]), "+v"(acc[x2][mt][3]));
    };
    group(std::integral_constant<int, 0>{}); WF_LOAD_BQ(0, un);
    group(std::integral_constant<int, 1>{}); WF_LOAD_BQ(1, un);
    group(std::integral_constant<int, 2>{}); WF_LOAD_BQ(2, un);
    group(std::integral_constant<int, 3>{}); WF_LOAD_BQ(3, un);
    patch_ready();
    transform();                              // V(c+1)
    if (UP) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(v[0][mt].x), "+v"(v[0][mt].y), "+v"(v[1][mt].x), "+v"(v[1][mt].y));
      // raw(c+2) <- low-res(c+2) (issued two iterations ago: landed for every wave); read at the top of the next iteration
      expand((c + 2 < nch ? c + 2 : nch - 1) % 5, (c + 2) & 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);        // drain the tail DMA / loads before the pipeline buffers become the M overlay
  asm volatile("" :: "v"(bq[0]), "v"(bq[1]), "v"(bq[2]), "v"(bq[3]));      // the filter registers stay allocated until their last (unused) reload has landed
  __syncthreads();

  // ---- epilogue in two 32-cout passes ---------------------------------------------------------------------------------
  float* Ms = smem;
  constexpr int NTILES = 32 * R;
  const int co = tid & 31, tg = tid >> 5;    // thread = (cout of the half, tile group); tiles tg + 16*k
  // per-channel epilogue constants of both halves, requested before the first LDS pass so that their L2 latency sits under
  // the accumulator writes instead of in front of each half's output transform
  float bvh[2], sch[2], shh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int cg = n0 + 32 * h + co;
    bvh[h] = a.bias ? a.bias[cg] : 0.0f;
    sch[h] = a.scale ? a.scale[cg] : 1.0f;
    shh[h] = a.scale ? a.shift[cg] : 0.0f;
  }
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 t = h == 0 ? acc[x2][mt][n2] : acc[x2][mt][2 + n2];
            Ms[((xi0 + x2) * NTILES + mt * 16 + 4 * fg + r) * M2_STRIDE + n2 * 16 + fi] = t[r];
          }
    __syncthreads();
    const int cg = n0 + 32 * h + co;
    const float bv = h == 0 ? bvh[0] : bvh[1], sc = h == 0 ? sch[0] : sch[1], sh = h == 0 ? shh[0] : shh[1];
#pragma unroll 1
    for (int k = 0; k < 2 * R; ++k) {
      const int tile = tg + 16 * k, tyy = tile >> 3, txx = tile & 7;
      const float* mp = Ms + tile * M2_STRIDE + co;
      float s[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m0 = mp[(0 * 4 + j) * NTILES * M2_STRIDE], m1 = mp[(1 * 4 + j) * NTILES * M2_STRIDE];
        const float m2 = mp[(2 * 4 + j) * NTILES * M2_STRIDE], m3 = mp[(3 * 4 + j) * NTILES * M2_STRIDE];
        s[0][j] = m0 + m1 + m2;
        s[1][j] = m1 - m2 - m3;
      }
      float pm = -3.0e38f;
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
          if (a.y) a.y[((size_t)(img * a.H + oh) * a.W + ow) * a.Cout + cg] = v;
          pm = fmaxf(pm, v);
        }
      }
      // fused MaxPool2D(2): H and W are even, so a tile inside the image has all four outputs
      if (a.yp && oh0 + 2 * tyy < a.H && ow0 + 2 * txx < a.W)
        a.yp[((size_t)(img * (a.H >> 1) + (oh0 >> 1) + tyy) * (a.W >> 1) + (ow0 >> 1) + txx) * a.Cout + cg] = pm;
    }
    __syncthreads();
  }
}

template <int R, bool UP = false>
int launch_fused(WinoFusedArgs& a, hipStream_t st) {
  a.tiles_x = (a.W + 15) / 16;
  a.tiles_y = (a.H + 8 * R - 1) / (8 * R);
  a.nblk_m = a.N * a.tiles_y * a.tiles_x;
  a.nblk_n = a.Cout / 64;
  const long nblk = (long)a.nblk_m * a.nblk_n;
  if (nblk <= 0 || nblk > 0x7fffffffL) return shdr::fail(SHDR_E_SHAPE, "winograd_fused: grid of %ld blocks", nblk);
  static_assert(!UP || WF<R>::PIPE_FLOATS + 5 * 8 * 256 <= WF<R>::LDS_BYTES / 4, "the low-res ring lives inside the epilogue overlay");
  static bool attr_done[shdr::kMaxDevices] = {};
  const int dev_slot = shdr::device_slot();
  if (!attr_done[dev_slot]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd_fused_kernel<R, UP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WF<R>::LDS_BYTES);
    if (e != hipSuccess) return shdr::fail(SHDR_E_ARCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done[dev_slot] = true;
  }
  hipLaunchKernelGGL((winograd_fused_kernel<R, UP>), dim3((unsigned)nblk), dim3(512), WF<R>::LDS_BYTES, st, a);
  return shdr::check_launch("winograd_fused_kernel");
}

}  // namespace

extern "C" int shdr_conv2d_winograd_fused2_f32(const float* x, const float* x2, const float* u, const float* bias,
                                               const float* scale, const float* shift, float* y, float* y_pool, int N, int H,
                                               int W, int C1, int C2, int Cout, int act1, int act2, void* stream) {
  SHDR_REQUIRE(y_pool == nullptr || (H % 2 == 0 && W % 2 == 0), SHDR_E_SHAPE, "winograd_fused: the fused 2x2 max-pool needs even H, W");
  SHDR_REQUIRE(x && u && (y || y_pool), SHDR_E_NULL, "winograd_fused: null x/u or neither y nor y_pool");
  SHDR_REQUIRE((x2 == nullptr) == (C2 == 0), SHDR_E_NULL, "winograd_fused: x2 and C2 come together");
  SHDR_REQUIRE(C2 == 0 || (C2 == C1 && C1 % 8 == 0 && shdr::aligned16(x2)), SHDR_E_SHAPE,
               "winograd_fused: the two sources of a concatenation need the same channel count, a multiple of 8, 16-byte aligned");
  const int Cin = C1 + C2;
  SHDR_REQUIRE((scale == nullptr) == (shift == nullptr), SHDR_E_NULL, "winograd_fused: scale and shift come together");
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, SHDR_E_SHAPE, "winograd_fused: non-positive dimension");
  SHDR_REQUIRE(Cin % 8 == 0 && Cout % 64 == 0, SHDR_E_SHAPE, "winograd_fused: need Cin %% 8 == 0 and Cout %% 64 == 0");
  SHDR_REQUIRE(shdr::aligned16(x) && shdr::aligned16(u), SHDR_E_ALIGN, "winograd_fused: x and u must be 16-byte aligned");
  SHDR_REQUIRE((long)N * H * W * Cin < (1L << 32) && 16L * Cin * Cout < (1L << 32), SHDR_E_SHAPE,
               "winograd_fused: tensor with more than 2^32 elements");
  WinoFusedArgs a{};
  a.x = x; a.x2 = x2 ? x2 : x; a.u = u; a.bias = bias; a.scale = scale; a.shift = shift; a.y = y; a.yp = y_pool;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.C1 = C1; a.Cout = Cout;
  a.act1 = act1; a.act2 = act2;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // The 16 x 16 tile (one block per CU, half the filter bytes per MFMA) was measured equal on wide layers and 5..10 % slower on
  // narrow ones (the kernel is not bound by the filter stream): it stays selectable for experiments only.
  const char* force = SHDR_ENV("SHDR_WINOGRAD_TILE");        // "16": the tall tile
  return (force && force[0] == '1') ? launch_fused<2>(a, st) : launch_fused<1>(a, st);
}

extern "C" int shdr_conv2d_winograd_fused_f32(const float* x, const float* u, const float* bias, const float* scale,
                                              const float* shift, float* y, int N, int H, int W, int Cin, int Cout,
                                              int act1, int act2, void* stream) {
  return shdr_conv2d_winograd_fused2_f32(x, nullptr, u, bias, scale, shift, y, nullptr, N, H, W, Cin, 0, Cout, act1, act2, stream);
}

// Conv2D 3x3 SAME stride 1 of tf.image.resize(x, 2x, BILINEAR) (hallucination_net.py:86-88, dequantization_net.py:25-27) in one
// kernel: x is the LOW-RES tensor [N,H/2,W/2,Cin], H x W the (even) size of the up-sampled image = of y [N,H,W,Cout].
extern "C" int shdr_conv2d_winograd_fused_up2_f32(const float* x, const float* u, const float* bias, const float* scale,
                                                  const float* shift, float* y, int N, int H, int W, int Cin, int Cout,
                                                  int act1, int act2, void* stream) {
  SHDR_REQUIRE(x && u && y, SHDR_E_NULL, "winograd_fused_up2: null x/u/y");
  SHDR_REQUIRE((scale == nullptr) == (shift == nullptr), SHDR_E_NULL, "winograd_fused_up2: scale and shift come together");
  SHDR_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, SHDR_E_SHAPE, "winograd_fused_up2: non-positive dimension");
  SHDR_REQUIRE(H % 2 == 0 && W % 2 == 0, SHDR_E_SHAPE, "winograd_fused_up2: H, W are the up-sampled (even) dimensions");
  SHDR_REQUIRE(Cin % 8 == 0 && Cout % 64 == 0, SHDR_E_SHAPE, "winograd_fused_up2: need Cin %% 8 == 0 and Cout %% 64 == 0");
  SHDR_REQUIRE(shdr::aligned16(x) && shdr::aligned16(u), SHDR_E_ALIGN, "winograd_fused_up2: x and u must be 16-byte aligned");
  SHDR_REQUIRE((long)N * H * W * Cin < (1L << 32) && 16L * Cin * Cout < (1L << 32), SHDR_E_SHAPE,
               "winograd_fused_up2: tensor with more than 2^32 elements");
  WinoFusedArgs a{};
  a.x = x; a.x2 = x; a.u = u; a.bias = bias; a.scale = scale; a.shift = shift; a.y = y; a.yp = nullptr;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.C1 = Cin; a.Cout = Cout; a.Hl = H / 2; a.Wl = W / 2;
  a.act1 = act1; a.act2 = act2;
  return launch_fused<1, true>(a, reinterpret_cast<hipStream_t>(stream));
}
