// Dispatch policy of the convolution BELOW the C ABI: which kernel family a layer takes, the filter
// preparation that family needs, and the input gradient with all its decompositions.  A host that binds libshdr (TF custom op, C,
// ctypes) calls
//     shdr_conv2d_plan_f32 / shdr_conv2d_prepared_filter_elems_f32 / shdr_conv2d_prepare_filter_f32      once per filter version
//     shdr_conv2d_fwd_prepared_f32                                                                        per step
//     shdr_conv2d_dgrad_f32 (+ shdr_conv2d_dgrad_workspace_bytes_f32)                                     in the backward pass
// and gets the one-kernel Winograd F(2x2,3x3), the register-A, the LDS-DMA or the direct kernel exactly as the Python layer of
// this package does -- that layer holds no dispatch logic of its own any more.
// Replaces tf.keras.layers.Conv2D / tf.nn.conv2d forward and GradientTape.gradient w.r.t. the conv input (dequantization_net.py:8-46,
// linearization_net.py:12-101, hallucination_net.py:47-140, refinement_net.py:8-47, vgg16.py:33-35, joint_training.py:185).
#include <stdlib.h>

#include "shdr_internal.h"

extern "C" int shdr_conv2d_x3_prepare_filter_premax_f32(const shdr_conv2d_desc* d, const float* w, float* prepared, int premax, void* stream);
extern "C" int shdr_conv2d_x3n_prepare_filter_premax_f32(const shdr_conv2d_desc* d, const float* w, float* prepared, int premax, void* stream);

namespace {

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

// the Winograd form a 3x3 / stride-1 / SAME layer takes (measured on MI355X, tools/wino_bench.py): the one-kernel fused form beats
// the direct kernels on every shape it accepts; the three-kernel "planes" form pays from 128 -> 256 / 256 -> 128 channels up
constexpr int X3_HEADER_FLOATS_PUB = 16;       // header of a packed split-operand filter (conv_x3.hip)
inline bool is_auto(int algo) { return algo == SHDR_ALGO_AUTO || algo == SHDR_ALGO_AUTO_EXACT; }

int plan_of(const shdr_conv2d_desc* d, bool has_residual) {
  const int Ct = d->C1 + d->C2;
  const int cout_valid = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  const bool mfma_ok = d->C1 % 4 == 0 && d->C2 % 4 == 0 && d->Cout % 16 == 0;
  if (!is_auto(d->algo)) return d->algo == SHDR_ALGO_DIRECT ? SHDR_PLAN_DIRECT : SHDR_PLAN_MFMA;
  int pt = 0, pl = 0, ho = 0, wo = 0;
  shdr_same_pad(d->H, d->KH, d->stride, &ho, &pt);
  shdr_same_pad(d->W, d->KW, d->stride, &wo, &pl);
  const bool same = d->pad_t == pt && d->pad_l == pl && d->Ho == ho && d->Wo == wo;
  const bool wino_shape = d->KH == 3 && d->KW == 3 && d->stride == 1 && same && !has_residual && cout_valid == d->Cout &&
                          d->w_batch_stride == 0 && d->y_pix_stride <= 1 && SHDR_ENV("SHDR_NO_WINOGRAD") == nullptr;
  // the narrow layers of the U-Nets (Cout 16 / 32, <= 32 channels per tap): the split-operand arithmetic with the whole filter in LDS;
  // its epilogue takes a residual
  if (d->algo == SHDR_ALGO_AUTO && same && SHDR_ENV("SHDR_NO_WINOGRAD") == nullptr && shdr_conv2d_x3n_ok_f32(d)) return SHDR_PLAN_X3N;
  // the split-operand fp16 kernel first (1.4-1.5x the fused Winograd kernel's rate, same accuracy class; also the 7x7 / 2 stem);
  // SHDR_ALGO_AUTO_EXACT opts out
  // (with a residual: the stride-1 layers only -- the ResNet joins on the 1x1 layers; shdr_conv2d_fwd_x3_residual_f32)
  if (d->algo == SHDR_ALGO_AUTO && (!has_residual || (d->stride == 1 && d->res_cstride % 4 == 0 && SHDR_ENV("SHDR_NO_X3_RESIDUAL") == nullptr)) && same &&
      SHDR_ENV("SHDR_NO_WINOGRAD") == nullptr && shdr_conv2d_x3_ok_f32(d))
    return SHDR_PLAN_X3;
  if (wino_shape) {
    const bool two_ok = d->C2 == 0 || (d->C2 == d->C1 && d->C1 % 8 == 0 && d->x2_scale == 1.0f);
    if (two_ok && Ct % 8 == 0 && d->Cout % 64 == 0 && Ct >= 32 && (long)d->N * d->H * d->W * Ct < (1L << 32)) return SHDR_PLAN_WINOGRAD_FUSED;
    if (d->C2 == 0 && Ct % 32 == 0 && d->Cout % 16 == 0 && (Ct < d->Cout ? Ct : d->Cout) >= 128 && (long)Ct * d->Cout >= 32768)
      return SHDR_PLAN_WINOGRAD_PLANES;
  }
  return mfma_ok ? SHDR_PLAN_MFMA : SHDR_PLAN_DIRECT;
}

// prepared[k][co] = w[k][co] * (k's channel >= C1 ? x2_scale : 1): the plain HWIO filter with the skip scale folded in
__global__ __launch_bounds__(256) void fold_x2_scale_kernel(const float* __restrict__ w, float* __restrict__ out, long total, int Ct, int C1,
                                                            int Cout, float x2_scale) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)((e / Cout) % Ct);
    out[e] = w[e] * (c >= C1 ? x2_scale : 1.0f);
  }
}

// dgrad filter with slicing and zero padding in one pass:
//   wt[kh][kw][co][ci] = scale * w[KH-1-kh][KW-1-kw][c_begin + ci][co]   for co < cout_real, ci < c_count, zero for the padded rest
//   (co < CZ = channels per pixel of the dz tensor the conv will read, ci < CC = output channels of that conv)
//   maxslot (optional): receives max |wt| (bits, atomicMax; zeroed by the caller) -- the header slot of the split-operand filter that is
//   packed from wt next, which then needs no absmax launch of its own
__global__ __launch_bounds__(256) void dgrad_filter_kernel(const float* __restrict__ w, float* __restrict__ wt, int KH, int KW, int Ct, int Cout,
                                                           int cout_real, int c_begin, int c_count, int CZ, int CC, float scale,
                                                           unsigned* __restrict__ maxslot) {
  const long total = (long)KH * KW * CZ * CC;
  float m = 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int ci = (int)(e % CC);
    long t = e / CC;
    const int co = (int)(t % CZ);
    t /= CZ;
    const int kw = (int)(t % KW), kh = (int)(t / KW);
    float v = 0.0f;
    if (ci < c_count && co < cout_real) v = scale * w[(((long)(KH - 1 - kh) * KW + (KW - 1 - kw)) * Ct + c_begin + ci) * Cout + co];
    wt[e] = v;
    m = fmaxf(m, fabsf(v));
  }
  if (maxslot) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(maxslot, __float_as_uint(m));
  }
}

// sub[a][b][co][ci] = wt[a0 + 2a][b0 + 2b][co][ci]: the taps one input-pixel parity of a stride-2 convolution sees
__global__ __launch_bounds__(256) void subfilter_kernel(const float* __restrict__ wt, float* __restrict__ sub, int KW, int a0, int b0, int TH, int TW,
                                                        long cc) {
  const long total = (long)TH * TW * cc;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e % cc;
    const long t = e / cc;
    const int b = (int)(t % TW), a = (int)(t / TW);
    sub[e] = wt[((long)(a0 + 2 * a) * KW + (b0 + 2 * b)) * cc + r];
  }
}

struct DgradGeom {
  int c_begin, c_count, cout_real, CZ, CC;      // CZ: dz channels the conv reads (multiple of 4), CC: conv output channels (padded to 16 when narrow)
  bool pad_dz, wino, x3, x3n;
  size_t off_wt, off_u, off_dz, off_sub, total;
};

DgradGeom dgrad_geom(const shdr_conv2d_desc* d, int which) {
  DgradGeom g{};
  g.c_begin = which ? d->C1 : 0;
  g.c_count = which ? d->C2 : d->C1;
  g.cout_real = d->cout_valid > 0 ? d->cout_valid : d->Cout;
  g.CZ = (g.cout_real + 3) / 4 * 4;
  g.pad_dz = g.CZ != g.cout_real;
  g.CC = g.c_count % 16 == 0 ? g.c_count : (g.c_count + 15) / 16 * 16;
  // stride-1 3x3 layers take the fused Winograd kernel when the transposed shape qualifies (the rule of plan_of)
  g.wino = is_auto(d->algo) && d->stride == 1 && d->KH == 3 && d->KW == 3 && !g.pad_dz && g.CC == g.c_count && g.CZ % 8 == 0 && g.CC % 64 == 0 && g.CZ >= 32 &&
           (long)d->N * d->Ho * d->Wo * g.CZ < (1L << 32) && SHDR_ENV("SHDR_NO_WINOGRAD") == nullptr;
  // ... and the split-operand fp16 kernel before it, by the rule of plan_of on the transposed convolution
  g.x3 = false;
  if (g.wino && d->algo == SHDR_ALGO_AUTO && g.CZ % 32 == 0) {
    shdr_conv2d_desc t{};
    t.N = d->N; t.H = d->Ho; t.W = d->Wo; t.C1 = g.CZ; t.Cout = g.CC; t.KH = 3; t.KW = 3; t.stride = 1; t.pad_t = 2 - d->pad_t; t.pad_l = 2 - d->pad_l;
    t.Ho = d->H; t.Wo = d->W; t.cout_valid = g.CC;
    g.x3 = shdr_conv2d_x3_ok_f32(&t) != 0;
  }
  // the narrow stride-1 layers of the U-Nets: the split-operand kernel with the whole (transposed) filter in LDS
  g.x3n = false;
  shdr_conv2d_desc tn{};
  if (d->algo == SHDR_ALGO_AUTO && d->stride == 1 && d->KH == d->KW && !g.pad_dz && SHDR_ENV("SHDR_NO_WINOGRAD") == nullptr) {
    tn.N = d->N; tn.H = d->Ho; tn.W = d->Wo; tn.C1 = g.CZ; tn.Cout = g.CC; tn.KH = d->KH; tn.KW = d->KW; tn.stride = 1;
    tn.pad_t = (d->KH - 1) - d->pad_t; tn.pad_l = (d->KW - 1) - d->pad_l; tn.Ho = d->H; tn.Wo = d->W; tn.cout_valid = g.c_count;
    g.x3n = shdr_conv2d_x3n_ok_f32(&tn) != 0;
  }
  const size_t filt = (size_t)d->KH * d->KW * g.CZ * g.CC * sizeof(float);
  size_t o = 0;
  g.off_wt = o; o += up256(filt);
  g.off_u = o;
  if (g.x3n) o += up256((size_t)shdr_conv2d_x3n_filter_elems_f32(&tn) * sizeof(float));
  else if (g.x3) o += up256((size_t)(X3_HEADER_FLOATS_PUB + (int64_t)9 * g.CZ * g.CC) * sizeof(float));
  else if (g.wino) o += up256((size_t)16 * g.CZ * g.CC * sizeof(float));
  g.off_dz = o; if (g.pad_dz) o += up256((size_t)d->N * d->Ho * d->Wo * g.CZ * sizeof(float));
  g.off_sub = o; if (d->stride == 2 && !(d->KH == 1 && d->KW == 1)) o += up256(filt);
  g.total = o;
  return g;
}

}  // namespace

extern "C" int shdr_conv2d_plan_f32(const shdr_conv2d_desc* d, int has_residual) {
  if (!d) return -1;
  return plan_of(d, has_residual != 0);
}

// 1 if the prepared filter of this layer is byte-identical to the HWIO filter itself (no Winograd transform, no skip scale to fold):
// the host may then hand `w` to shdr_conv2d_fwd_prepared_f32 directly and skip the copy
extern "C" int shdr_conv2d_filter_is_plain_f32(const shdr_conv2d_desc* d, int has_residual) {
  if (!d) return -1;
  const int plan = plan_of(d, has_residual != 0);
  return (plan == SHDR_PLAN_MFMA || plan == SHDR_PLAN_DIRECT) && (d->C2 == 0 || d->x2_scale == 1.0f) ? 1 : 0;
}

extern "C" int64_t shdr_conv2d_prepared_filter_elems_f32(const shdr_conv2d_desc* d, int has_residual) {
  if (!d) return -1;
  const int plan = plan_of(d, has_residual != 0);
  const int64_t Ct = d->C1 + d->C2;
  if (plan == SHDR_PLAN_X3) return shdr_conv2d_x3_filter_elems_f32(d);
  if (plan == SHDR_PLAN_X3N) return shdr_conv2d_x3n_filter_elems_f32(d);
  if (plan == SHDR_PLAN_WINOGRAD_FUSED || plan == SHDR_PLAN_WINOGRAD_PLANES) return 16 * Ct * d->Cout;
  return (int64_t)d->KH * d->KW * Ct * d->Cout;
}

extern "C" int shdr_conv2d_prepare_filter_f32(const shdr_conv2d_desc* d, int has_residual, const float* w, float* prepared, void* stream) {
  SHDR_REQUIRE(d && w && prepared, SHDR_E_NULL, "prepare_filter: null pointer");
  const int plan = plan_of(d, has_residual != 0);
  const int Ct = d->C1 + d->C2;
  if (plan == SHDR_PLAN_X3) return shdr_conv2d_x3_prepare_filter_f32(d, w, prepared, stream);
  if (plan == SHDR_PLAN_X3N) return shdr_conv2d_x3n_prepare_filter_f32(d, w, prepared, stream);
  if (plan == SHDR_PLAN_WINOGRAD_FUSED) return shdr_winograd_filter_packed_f32(w, prepared, Ct, d->Cout, stream);
  if (plan == SHDR_PLAN_WINOGRAD_PLANES) return shdr_winograd_filter_f32(w, prepared, Ct, d->Cout, stream);
  const long total = (long)d->KH * d->KW * Ct * d->Cout;
  const float sc = d->C2 > 0 ? d->x2_scale : 1.0f;
  hipLaunchKernelGGL(fold_x2_scale_kernel, dim3(shdr::stream_grid(total)), dim3(256), 0, S(stream), w, prepared, total, Ct, d->C1, d->Cout, sc);
  return shdr::check_launch("prepare_filter");
}

// the bilinear 2x prologue runs inside the convolution kernel on the fused Winograd plan (single source, no pooled output)
// ... and on the split-operand plan where the up-sampling pass is dear next to the convolution: the in-kernel expansion is repeated by
// every 64-cout block of a tile, the resize2x pass costs 1 / Cout of the layer -- measured (tools/up2_bench.py, 16 x 512^2 step shapes):
// 128 -> 64 2.70 -> 2.03 ms, 256 -> 128 2.13 -> 1.88, 512 -> 256 1.86 -> 1.79, 512 -> 512 0.91 -> 0.96: fused up to 256 couts
inline bool up2_in_kernel(const shdr_conv2d_desc* d, int plan) {
  if (d->C2 != 0) return false;
  if (plan == SHDR_PLAN_X3) return d->KH == 3 && d->stride == 1 && (d->Cout <= 256 || SHDR_ENV("SHDR_X3_UP_ALWAYS") != nullptr);
  return plan == SHDR_PLAN_WINOGRAD_FUSED;
}
// bytes of the materialised up-sampled tensor in front of the plan's own workspace (0 when the prologue is fused or absent)
inline size_t up2_bytes(const shdr_conv2d_desc* d, int plan) {
  if (d->prologue != SHDR_PROLOGUE_BILINEAR2X || up2_in_kernel(d, plan)) return 0;
  return up256((size_t)d->N * d->H * d->W * d->C1 * sizeof(float));
}

// The split-operand plans scale their input by a power of two taken from a RANGE SLOT per source (conv_x3.hip "Range").  A caller that
// does not know the range of a source (x1_range / x2_range = NULL) gets it measured here: one absmax pass over that source into two
// scratch slots at the tail of the workspace.
constexpr size_t kRangeScratch = 256;
inline bool split_plan(int plan) { return plan == SHDR_PLAN_X3 || plan == SHDR_PLAN_X3N; }

extern "C" int64_t shdr_conv2d_workspace_bytes_f32(const shdr_conv2d_desc* d, int has_residual) {
  if (!d) return -1;
  const int plan = plan_of(d, has_residual != 0);
  const size_t up = up2_bytes(d, plan);
  if (split_plan(plan)) return (int64_t)(up + kRangeScratch);
  if (plan != SHDR_PLAN_WINOGRAD_PLANES) return (int64_t)up;
  const int64_t rows = shdr_winograd_tiles(d->N, d->H, d->W);          // rows of each of the 16 transform planes
  return (int64_t)(up + up256((size_t)16 * rows * (d->C1 + d->C2) * sizeof(float)) + up256((size_t)16 * rows * d->Cout * sizeof(float)));
}

// a split-operand launch never runs unscaled from the planned entry points: the range of a source that arrives without a slot is measured
// into the scratch slots at the tail of the workspace (x1 of the bilinear prologue is the low-res tensor; bilinear weights are convex,
// so its bound holds for the up-sampled one)
static int measure_missing_ranges(const shdr_conv2d_desc* d, int has_residual, const float* x1, const float* x2, void* workspace,
                                  const float*& x1_range, const float*& x2_range, void* stream) {
  if (x1_range && (!x2 || x2_range)) return SHDR_OK;
  SHDR_REQUIRE(workspace && shdr::aligned16(workspace), SHDR_E_NULL,
               "conv2d_fwd_prepared: a split-operand layer without x ranges needs shdr_conv2d_workspace_bytes_f32 bytes of workspace");
  float* slots = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + shdr_conv2d_workspace_bytes_f32(d, has_residual) - kRangeScratch);
  if (hipMemsetAsync(slots, 0, 2 * sizeof(float), S(stream)) != hipSuccess) return shdr::fail(SHDR_E_LAUNCH, "conv2d_fwd_prepared: memset failed");
  const bool lowres = d->prologue == SHDR_PROLOGUE_BILINEAR2X;
  const int64_t npix = (int64_t)d->N * (lowres ? d->H / 2 : d->H) * (lowres ? d->W / 2 : d->W);
  if (!x1_range) {
    if (int rcm = shdr_absmax_f32(x1, npix * d->C1, slots, stream)) return rcm;
    x1_range = slots;
  }
  if (x2 && !x2_range) {
    if (int rcm = shdr_absmax_f32(x2, npix * d->C2, slots + 1, stream)) return rcm;
    x2_range = slots + 1;
  }
  return SHDR_OK;
}

extern "C" int shdr_conv2d_fwd_prepared_ranged_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared,
                                                   const float* bias, const float* scale, const float* shift, const float* residual, float* y,
                                                   float* y_pool, void* workspace, const float* x1_range, const float* x2_range,
                                                   float* y_range, void* stream) {
  SHDR_REQUIRE(d && x1 && prepared && (y || y_pool), SHDR_E_NULL, "conv2d_fwd_prepared: null desc / x1 / prepared filter / output");
  SHDR_REQUIRE(!y_pool || (d->Ho % 2 == 0 && d->Wo % 2 == 0), SHDR_E_SHAPE, "conv2d_fwd_prepared: the fused 2x2 pooling needs even Ho, Wo");
  SHDR_REQUIRE(d->pool == SHDR_POOL_MAX || d->pool == SHDR_POOL_AVG, SHDR_E_SHAPE, "conv2d_fwd_prepared: unknown pool kind");
  const int plan = plan_of(d, residual != nullptr);
  const bool avg = y_pool && d->pool == SHDR_POOL_AVG;
  if (split_plan(plan)) {
    if (int rcm = measure_missing_ranges(d, residual != nullptr, x1, x2, workspace, x1_range, x2_range, stream)) return rcm;
  }
  const bool epilogue_range = split_plan(plan) || ((plan == SHDR_PLAN_MFMA || plan == SHDR_PLAN_DIRECT) && d->prologue == SHDR_PROLOGUE_NONE && y);
  if (y_range && !epilogue_range) {
    // plans whose epilogue does not track the output range: one pass over the output(s) after the convolution
    SHDR_REQUIRE(d->y_pix_stride <= 1 && (d->y_cstride == 0 || d->y_cstride == (d->cout_valid > 0 ? d->cout_valid : d->Cout)), SHDR_E_SHAPE,
                 "conv2d_fwd_prepared: y_range needs a dense output");
    if (int rcc = shdr_conv2d_fwd_prepared_ranged_f32(d, x1, x2, prepared, bias, scale, shift, residual, y, y_pool, workspace, x1_range, x2_range,
                                                      nullptr, stream))
      return rcc;
    const int64_t cv = d->cout_valid > 0 ? d->cout_valid : d->Cout;
    if (y) return shdr_absmax_f32(y, (int64_t)d->N * d->Ho * d->Wo * cv, y_range, stream);
    return shdr_absmax_f32(y_pool, (int64_t)d->N * (d->Ho / 2) * (d->Wo / 2) * cv, y_range, stream);
  }
  if (avg && (plan == SHDR_PLAN_WINOGRAD_FUSED || (d->prologue == SHDR_PROLOGUE_BILINEAR2X && plan != SHDR_PLAN_X3))) {
    // the Winograd kernel's epilogue pools by maximum only: convolution, then the pooling launch
    SHDR_REQUIRE(y, SHDR_E_NULL, "conv2d_fwd_prepared: this plan writes y before it pools");
    if (int rcw = shdr_conv2d_fwd_prepared_ranged_f32(d, x1, x2, prepared, bias, scale, shift, residual, y, nullptr, workspace, x1_range, x2_range,
                                                      y_range, stream))
      return rcw;
    return shdr_avgpool2_fwd_f32(y, y_pool, d->N, d->Ho, d->Wo, d->Cout, stream);
  }
  if (d->prologue != SHDR_PROLOGUE_NONE) {
    SHDR_REQUIRE(d->prologue == SHDR_PROLOGUE_BILINEAR2X, SHDR_E_SHAPE, "conv2d_fwd_prepared: unknown prologue");
    SHDR_REQUIRE(d->H % 2 == 0 && d->W % 2 == 0 && d->C2 == 0 && x2 == nullptr, SHDR_E_SHAPE,
                 "conv2d_fwd_prepared: the bilinear 2x prologue takes one source and even (up-sampled) H, W");
    if (plan == SHDR_PLAN_X3 && up2_in_kernel(d, plan))
      return shdr_conv2d_fwd_x3_ranged_f32(d, x1, nullptr, prepared, bias, scale, shift, y, y_pool, x1_range, nullptr, y_range, stream);
    if (up2_in_kernel(d, plan)) {
      SHDR_REQUIRE(y, SHDR_E_NULL, "conv2d_fwd_prepared: the bilinear 2x prologue writes y");
      int rcf = shdr_conv2d_winograd_fused_up2_f32(x1, prepared, bias, scale, shift, y, d->N, d->H, d->W, d->C1, d->Cout, d->act1, d->act2, stream);
      if (rcf || !y_pool) return rcf;
      return shdr_maxpool2_fwd_f32(y, y_pool, d->N, d->Ho, d->Wo, d->Cout, stream);
    }
    // every other plan: the up-sampled tensor goes through the head of the workspace
    SHDR_REQUIRE(workspace && shdr::aligned16(workspace), SHDR_E_NULL, "conv2d_fwd_prepared: this layer needs shdr_conv2d_workspace_bytes_f32 bytes of workspace");
    float* xu = reinterpret_cast<float*>(workspace);
    int rcu = shdr_resize2x_fwd_f32(x1, xu, d->N, d->H / 2, d->W / 2, d->C1, stream);
    if (rcu) return rcu;
    shdr_conv2d_desc g = *d;
    g.prologue = SHDR_PROLOGUE_NONE;
    return shdr_conv2d_fwd_prepared_ranged_f32(&g, xu, nullptr, prepared, bias, scale, shift, residual, y, y_pool,
                                               reinterpret_cast<char*>(workspace) + up256((size_t)d->N * d->H * d->W * d->C1 * sizeof(float)),
                                               x1_range, nullptr, y_range, stream);
  }
  if (plan == SHDR_PLAN_WINOGRAD_FUSED)
    return shdr_conv2d_winograd_fused2_f32(x1, x2, prepared, bias, scale, shift, y, y_pool, d->N, d->H, d->W, d->C1, d->C2, d->Cout, d->act1,
                                           d->act2, stream);
  if (plan == SHDR_PLAN_X3) {
    if (residual) {
      SHDR_REQUIRE(y && !y_pool, SHDR_E_SHAPE, "conv2d_fwd_prepared: a layer with a residual writes y and has no pooled output");
      return shdr_conv2d_fwd_x3_residual_f32(d, x1, x2, prepared, bias, scale, shift, residual, y, x1_range, x2_range, y_range, stream);
    }
    return shdr_conv2d_fwd_x3_ranged_f32(d, x1, x2, prepared, bias, scale, shift, y, y_pool, x1_range, x2_range, y_range, stream);
  }
  SHDR_REQUIRE(y, SHDR_E_NULL, "conv2d_fwd_prepared: y may be omitted only on the fused Winograd and split-operand paths");
  int rc;
  if (plan == SHDR_PLAN_X3N) {
    const bool in_kernel = y_pool && d->y_pix_stride <= 1 && (d->cout_valid == 0 || d->cout_valid == d->Cout);
    rc = shdr_conv2d_fwd_x3n_ranged_f32(d, x1, x2, prepared, bias, scale, shift, residual, y, in_kernel ? y_pool : nullptr, x1_range, x2_range, y_range,
                                        stream);
    if (rc || in_kernel) return rc;
  } else if (plan == SHDR_PLAN_WINOGRAD_PLANES) {
    SHDR_REQUIRE(workspace && shdr::aligned16(workspace), SHDR_E_NULL, "conv2d_fwd_prepared: this layer needs shdr_conv2d_workspace_bytes_f32 bytes of workspace");
    const int Cin = d->C1;
    const int64_t rows = shdr_winograd_tiles(d->N, d->H, d->W);
    float* V = reinterpret_cast<float*>(workspace);
    float* M = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + up256((size_t)16 * rows * Cin * sizeof(float)));
    if ((rc = shdr_winograd_input_f32(x1, V, d->N, d->H, d->W, Cin, stream))) return rc;
    // 16 GEMMs [rows / 16, Cin] @ [Cin, Cout] as ONE launch of the conv kernel on a "16-image" 1x1 problem with a per-image filter
    shdr_conv2d_desc g{};
    g.N = 16; g.H = (int)(rows / 16); g.W = 16; g.C1 = Cin; g.Cout = d->Cout; g.KH = 1; g.KW = 1; g.stride = 1;
    g.Ho = g.H; g.Wo = g.W; g.x2_scale = 1.0f; g.algo = SHDR_ALGO_AUTO; g.w_batch_stride = (int64_t)Cin * d->Cout;
    if ((rc = shdr_conv2d_fwd_f32(&g, V, nullptr, prepared, nullptr, nullptr, nullptr, nullptr, M, stream))) return rc;
    rc = shdr_winograd_output_f32(M, y, bias, scale, shift, d->N, d->H, d->W, d->Cout, d->act1, d->act2, stream);
  } else {
    shdr_conv2d_desc g = *d;
    g.x2_scale = 1.0f;                                   // folded into the prepared filter
    if (g.algo == SHDR_ALGO_AUTO_EXACT) g.algo = SHDR_ALGO_AUTO;
    rc = shdr_conv2d_fwd_yrange_f32(&g, x1, x2, prepared, bias, scale, shift, residual, y, y_range, stream);
  }
  if (rc) return rc;
  if (y_pool) {
    SHDR_REQUIRE(d->y_pix_stride <= 1, SHDR_E_SHAPE, "conv2d_fwd_prepared: no pooled output with a strided y");
    const int cv = d->cout_valid > 0 ? d->cout_valid : d->Cout;
    return avg ? shdr_avgpool2_fwd_f32(y, y_pool, d->N, d->Ho, d->Wo, cv, stream) : shdr_maxpool2_fwd_f32(y, y_pool, d->N, d->Ho, d->Wo, cv, stream);
  }
  return SHDR_OK;
}

// the same without range slots: every split-operand layer measures its input range (one pass over x per source)
// 1 if the planned kernel of the layer can write a projected output (shdr_conv2d_fwd_prepared_projected_f32)
extern "C" int shdr_conv2d_projected_ok_f32(const shdr_conv2d_desc* d) {
  if (!d || d->Cout != 64 || d->stride != 1 || plan_of(d, false) != SHDR_PLAN_X3) return 0;
  if (d->cout_valid != 0 && d->cout_valid != 64) return 0;
  return d->prologue == SHDR_PROLOGUE_NONE || (d->prologue == SHDR_PROLOGUE_BILINEAR2X && up2_in_kernel(d, SHDR_PLAN_X3));
}

// The planned forward call with a projected output (conv_x3.hip: shdr_conv2d_fwd_x3_projected_f32): y_proj[n,h,w,j] = sum_c proj[j][c] *
// y[n,h,w,c], j < 3, from the epilogue that holds y; y and y_pool are written only when given.  Layers for which
// shdr_conv2d_projected_ok_f32 is 0 are refused (the caller runs the convolution and the 1x1 map as two calls).
extern "C" int shdr_conv2d_fwd_prepared_projected_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared,
                                                      const float* bias, const float* scale, const float* shift, const float* proj,
                                                      float* y_proj, float* y, float* y_pool, void* workspace, const float* x1_range,
                                                      const float* x2_range, float* y_range, void* stream) {
  SHDR_REQUIRE(d && x1 && prepared && proj && y_proj, SHDR_E_NULL, "conv2d_fwd_prepared_projected: null desc / x1 / prepared filter / proj / y_proj");
  SHDR_REQUIRE(shdr_conv2d_projected_ok_f32(d), SHDR_E_SHAPE, "conv2d_fwd_prepared_projected: the planned kernel of this layer has no projected output");
  SHDR_REQUIRE(!y_pool || (d->Ho % 2 == 0 && d->Wo % 2 == 0), SHDR_E_SHAPE, "conv2d_fwd_prepared_projected: the fused 2x2 pooling needs even Ho, Wo");
  if (int rcm = measure_missing_ranges(d, 0, x1, x2, workspace, x1_range, x2_range, stream)) return rcm;
  return shdr_conv2d_fwd_x3_projected_f32(d, x1, x2, prepared, bias, scale, shift, proj, y_proj, y, y_pool, x1_range, x2_range, y_range, stream);
}

extern "C" int shdr_conv2d_fwd_prepared_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                                            const float* scale, const float* shift, const float* residual, float* y, float* y_pool,
                                            void* workspace, void* stream) {
  return shdr_conv2d_fwd_prepared_ranged_f32(d, x1, x2, prepared, bias, scale, shift, residual, y, y_pool, workspace, nullptr, nullptr, nullptr, stream);
}

extern "C" int64_t shdr_conv2d_dgrad_workspace_bytes_f32(const shdr_conv2d_desc* d, int which) {
  if (!d || (which != 0 && !(which == 1 && d->C2 > 0))) return -1;
  return (int64_t)dgrad_geom(d, which).total;
}

// 1 if the input-gradient launch of this layer writes the range slot of dx from its own epilogue (stride-1 layers on the split-operand
// and exact MFMA kernels): a host then hands a slot to shdr_conv2d_dgrad_ranged_f32 and passes it on to the consumer of dx
extern "C" int shdr_conv2d_dgrad_tracks_range_f32(const shdr_conv2d_desc* d, int which) {
  if (!d || (which != 0 && !(which == 1 && d->C2 > 0)) || d->stride != 1) return 0;
  const DgradGeom g = dgrad_geom(d, which);
  return (g.x3n || g.x3 || !g.wino) ? 1 : 0;
}

extern "C" int shdr_conv2d_dgrad_f32(const shdr_conv2d_desc* d, int which, const float* dz, const float* w, float* dx, void* workspace,
                                     void* stream) {
  return shdr_conv2d_dgrad_ranged_f32(d, which, dz, w, dx, workspace, nullptr, nullptr, stream);
}

// The same with range slots (conv_x3.hip "Range"): dz_range = upper bound of max |dz| (NULL: measured where a split-operand kernel needs
// it -- output gradients sit far below the fp16 range), dx_range (only where shdr_conv2d_dgrad_tracks_range_f32) = slot that receives max |dx|.
extern "C" int shdr_conv2d_dgrad_ranged_f32(const shdr_conv2d_desc* d, int which, const float* dz, const float* w, float* dx, void* workspace,
                                            const float* dz_range, float* dx_range, void* stream) {
  SHDR_REQUIRE(d && dz && w && dx && workspace, SHDR_E_NULL, "conv2d_dgrad: null pointer");
  SHDR_REQUIRE(which == 0 || (which == 1 && d->C2 > 0), SHDR_E_SHAPE, "conv2d_dgrad: `which` selects x1 (0) or x2 (1)");
  SHDR_REQUIRE(d->stride == 1 || d->stride == 2, SHDR_E_SHAPE, "conv2d_dgrad: stride %d is not built", d->stride);
  SHDR_REQUIRE(d->stride == 2 || (d->KH % 2 == 1 && d->KW % 2 == 1), SHDR_E_SHAPE, "conv2d_dgrad: stride-1 layers need odd filter sizes");
  SHDR_REQUIRE(shdr::aligned16(workspace) && shdr::aligned16(dz) && shdr::aligned16(dx), SHDR_E_ALIGN, "conv2d_dgrad: tensors must be 16-byte aligned");
  const DgradGeom g = dgrad_geom(d, which);
  char* ws = reinterpret_cast<char*>(workspace);
  float* wt = reinterpret_cast<float*>(ws + g.off_wt);
  hipStream_t st = S(stream);
  const int Ct = d->C1 + d->C2;
  const float scale = which ? d->x2_scale : 1.0f;
  // the split-operand plans pack wt next: its maximum comes out of this pass (header slot 0 of the packed filter)
  unsigned* maxslot = (d->stride == 1 && (g.x3n || g.x3)) ? reinterpret_cast<unsigned*>(ws + g.off_u) : nullptr;
  if (maxslot && hipMemsetAsync(maxslot, 0, 64, st) != hipSuccess) return shdr::fail(SHDR_E_LAUNCH, "conv2d_dgrad: memset failed");
  long fgrid = shdr::stream_grid((long)d->KH * d->KW * g.CZ * g.CC);
  if (maxslot && fgrid > 256) fgrid = 256;                 // (every wave ends in an atomicMax on one address)
  hipLaunchKernelGGL(dgrad_filter_kernel, dim3((unsigned)fgrid), dim3(256), 0, st, w, wt, d->KH, d->KW, Ct,
                     d->Cout, g.cout_real, g.c_begin, g.c_count, g.CZ, g.CC, scale, maxslot);
  if (int rc = shdr::check_launch("dgrad_filter")) return rc;
  const float* dzp = dz;
  if (g.pad_dz) {
    float* t = reinterpret_cast<float*>(ws + g.off_dz);
    if (int rc = shdr_pad_channels_f32(dz, t, (int64_t)d->N * d->Ho * d->Wo, g.cout_real, g.CZ, stream)) return rc;
    dzp = t;
  }
  shdr_conv2d_desc c{};
  c.N = d->N; c.H = d->Ho; c.W = d->Wo; c.C1 = g.CZ; c.C2 = 0; c.Cout = g.CC; c.stride = 1; c.x2_scale = 1.0f;
  c.cout_valid = g.c_count; c.algo = d->algo;             // (the reduced-precision operand modes carry over to the gradient convs)
  if (d->stride == 1) {
    c.KH = d->KH; c.KW = d->KW; c.pad_t = (d->KH - 1) - d->pad_t; c.pad_l = (d->KW - 1) - d->pad_l; c.Ho = d->H; c.Wo = d->W;
    if (g.x3n) {
      float* u = reinterpret_cast<float*>(ws + g.off_u);
      c.algo = SHDR_ALGO_AUTO;
      if (int rc = shdr_conv2d_x3n_prepare_filter_premax_f32(&c, wt, u, 1, stream)) return rc;
      if (!dz_range) {
        if (int rc = shdr_conv2d_x3_input_absmax_f32(dzp, (int64_t)c.N * c.H * c.W * c.C1, u, stream)) return rc;     // header slot 2, as the wide kernel
        c.prologue = SHDR_PROLOGUE_RANGE_SCALE;
      }
      return shdr_conv2d_fwd_x3n_ranged_f32(&c, dzp, nullptr, u, nullptr, nullptr, nullptr, nullptr, dx, nullptr, dz_range, nullptr, dx_range, stream);
    }
    if (g.x3) {
      float* u = reinterpret_cast<float*>(ws + g.off_u);
      c.cout_valid = g.CC; c.algo = SHDR_ALGO_AUTO;
      if (int rc = shdr_conv2d_x3_prepare_filter_premax_f32(&c, wt, u, 1, stream)) return rc;
      // output gradients sit far below the fp16 range (max |dz| 3e-8 ... 2e-2 in the joint step): scaled in the kernel by a power of two
      if (!dz_range) {
        if (int rc = shdr_conv2d_x3_input_absmax_f32(dzp, (int64_t)c.N * c.H * c.W * c.C1, u, stream)) return rc;
        c.prologue = SHDR_PROLOGUE_RANGE_SCALE;
      }
      return shdr_conv2d_fwd_x3_ranged_f32(&c, dzp, nullptr, u, nullptr, nullptr, nullptr, dx, nullptr, dz_range, nullptr, dx_range, stream);
    }
    if (g.wino) {
      SHDR_REQUIRE(!dx_range, SHDR_E_SHAPE, "conv2d_dgrad: the Winograd plan does not track dx_range (shdr_conv2d_dgrad_tracks_range_f32)");
      float* u = reinterpret_cast<float*>(ws + g.off_u);
      if (int rc = shdr_winograd_filter_packed_f32(wt, u, g.CZ, g.CC, stream)) return rc;
      return shdr_conv2d_winograd_fused2_f32(dzp, nullptr, u, nullptr, nullptr, nullptr, dx, nullptr, c.N, c.H, c.W, g.CZ, 0, g.CC, SHDR_ACT_NONE,
                                             SHDR_ACT_NONE, stream);
    }
    return shdr_conv2d_fwd_yrange_f32(&c, dzp, nullptr, wt, nullptr, nullptr, nullptr, nullptr, dx, dx_range, stream);
  }
  SHDR_REQUIRE(!dx_range, SHDR_E_SHAPE, "conv2d_dgrad: dx_range is written by the stride-1 kernels only (shdr_conv2d_dgrad_tracks_range_f32)");
  // ---- stride 2: the conv output lands on every second pixel of dx (strided placement) -----------------------------------------
  c.y_pix_stride = 2; c.y_H = d->H; c.y_W = d->W;
  const size_t dx_bytes = (size_t)d->N * d->H * d->W * g.c_count * sizeof(float);
  if (d->KH == 1 && d->KW == 1) {                         // 1x1 / 2 (linearization_net.py:12,16): dgrad on the coarse grid, zero elsewhere
    if (hipMemsetAsync(dx, 0, dx_bytes, st) != hipSuccess) return shdr::fail(SHDR_E_LAUNCH, "conv2d_dgrad: memset failed");
    c.KH = 1; c.KW = 1; c.pad_t = 0; c.pad_l = 0; c.Ho = d->Ho; c.Wo = d->Wo; c.y_off_h = 0; c.y_off_w = 0;
    return shdr_conv2d_fwd_f32(&c, dzp, nullptr, wt, nullptr, nullptr, nullptr, nullptr, dx, stream);
  }
  // general stride 2 (the 7x7 / 2 stem, linearization_net.py:91), polyphase form: the input pixels of parity (p, q) only see the filter
  // taps of one parity, so dx[:, p::2, q::2] is a stride-1 correlation of dz with a sub-filter of the flipped filter -- four small
  // convs with together exactly the forward's FLOPs instead of one k x k conv over a zero-inserted dz (4x the FLOPs)
  float* sub = reinterpret_cast<float*>(ws + g.off_sub);
  const long cc = (long)g.CZ * g.CC;
  auto phase = [](int par_in, int pad_fwd, int k, int* first, int* pad, int* taps) {
    const int par = (par_in + pad_fwd) % 2;
    int n = 0;
    for (int t = par; t < k; t += 2) ++n;
    const int off = (par_in + pad_fwd - par) / 2;
    *first = k - 1 - par - 2 * (n - 1);
    *pad = n - 1 - off;
    *taps = n;
  };
  bool zeroed = false;
  for (int p = 0; p < 2; ++p) {
    int a0, pad_t, th;
    phase(p, d->pad_t, d->KH, &a0, &pad_t, &th);
    const int mh = (d->H - p + 1) / 2;
    for (int q = 0; q < 2; ++q) {
      int b0, pad_l, tw;
      phase(q, d->pad_l, d->KW, &b0, &pad_l, &tw);
      const int mw = (d->W - q + 1) / 2;
      if (mh == 0 || mw == 0) continue;
      if (th == 0 || tw == 0) {                            // this parity sees no tap: its pixels are zero
        if (!zeroed) {
          SHDR_REQUIRE(p == 0 && q == 0, SHDR_E_SHAPE, "conv2d_dgrad: an empty phase after a written one is not handled");
          if (hipMemsetAsync(dx, 0, dx_bytes, st) != hipSuccess) return shdr::fail(SHDR_E_LAUNCH, "conv2d_dgrad: memset failed");
          zeroed = true;
        }
        continue;
      }
      hipLaunchKernelGGL(subfilter_kernel, dim3(shdr::stream_grid((long)th * tw * cc)), dim3(256), 0, st, wt, sub, d->KW, a0, b0, th, tw, cc);
      if (int rc = shdr::check_launch("subfilter")) return rc;
      c.KH = th; c.KW = tw; c.pad_t = pad_t; c.pad_l = pad_l; c.Ho = mh; c.Wo = mw; c.y_off_h = p; c.y_off_w = q;
      if (int rc = shdr_conv2d_fwd_f32(&c, dzp, nullptr, sub, nullptr, nullptr, nullptr, nullptr, dx, stream)) return rc;
    }
  }
  return SHDR_OK;
}

// generic workspace query of the boundary: bytes the caller has to provide for one call of `op`
extern "C" int64_t shdr_workspace_bytes(int op, const shdr_conv2d_desc* d, int arg) {
  switch (op) {
    case SHDR_OP_CONV2D_FWD: return shdr_conv2d_workspace_bytes_f32(d, arg);
    case SHDR_OP_CONV2D_DGRAD: return shdr_conv2d_dgrad_workspace_bytes_f32(d, arg);
    case SHDR_OP_CONV2D_WGRAD_WINOGRAD: return d ? (int64_t)16 * (arg ? d->C2 : d->C1) * d->Cout * (int64_t)sizeof(float) : -1;
    case SHDR_OP_BATCHNORM: return arg > 0 ? (int64_t)2 * arg * (1 + SHDR_BN_MAX_BLOCKS) * (int64_t)sizeof(double) : -1;     // arg = channels
    case SHDR_OP_ACT_BWD_BIAS: return arg > 0 ? (int64_t)arg * shdr::kBiasMaxBlocks * (int64_t)sizeof(float) : -1;
    default: return -1;
  }
}
