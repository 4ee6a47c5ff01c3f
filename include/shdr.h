/*
 * libshdr -- MI355X (gfx950) native kernels for the SingleHDR hot path.
 *
 * C ABI drop-in boundary.  The reference
 * (ShinYwings/SingleHDR-tf2) has no native code and no FFI: every entry point
 * below replaces a *TensorFlow op call site* of the reference, cited per
 * function as file:line under /root/reference.
 *
 * Conventions
 *  - all tensors are dense NHWC float32 in device memory, owned by the caller;
 *    the library allocates nothing and keeps no mutable global state;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *    no implicit device synchronisation;
 *  - return value: SHDR_OK (0) or a negative SHDR_E_* code; the message of the
 *    last error on the calling thread is available from shdr_last_error();
 *    shapes, alignment and strides are validated on the host before launch;
 *  - filters are HWIO ([KH][KW][Cin][Cout] row-major), as in Keras.
 */
#ifndef SHDR_H_
#define SHDR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library reads its SHDR_* environment switches once per process; call this after changing one in a running process. */
void shdr_config_reload(void);

#define SHDR_OK          0
#define SHDR_E_SHAPE    (-1)  /* inconsistent / unsupported dimensions          */
#define SHDR_E_ALIGN    (-2)  /* pointer or channel count not suitably aligned  */
#define SHDR_E_ARCH     (-3)  /* no gfx950 device / wrong code object           */
#define SHDR_E_LAUNCH   (-4)  /* hipLaunchKernel reported an error              */
#define SHDR_E_NULL     (-5)  /* required pointer is NULL                       */

/* activation codes for the conv epilogue */
#define SHDR_ACT_NONE   0
#define SHDR_ACT_RELU   1
#define SHDR_ACT_LRELU  2   /* leaky relu, slope 0.1 (dequantization_net.py:13) */
#define SHDR_ACT_TANH   3

/* kernel selection for shdr_conv2d_fwd_f32 */
#define SHDR_ALGO_AUTO    0
#define SHDR_ALGO_MFMA    1  /* fp32-MFMA implicit GEMM (needs (C1+C2)%4==0, Cout%16==0) */
#define SHDR_ALGO_DIRECT  2  /* VALU direct convolution (any shape)                     */
#define SHDR_ALGO_MFMA_REG 3 /* MFMA kernel with register-staged LDS fill (the LDS-DMA  */
                             /* variant is preferred whenever x2_scale == 1)           */
/* Reduced-precision MFMA operands for BASELINE configs[4] ("finetune_real_dataset.py ... fp16 MFMA conv path"):
 * tensors stay fp32 in HBM; activations and filters are rounded to nearest-even to fp16 (bf16) when they are
 * packed into the operands of v_mfma_f32_16x16x32_{f16,bf16}; products accumulate in fp32.  Needs x2_scale == 1.
 * Accepted by shdr_conv2d_fwd_f32 and, through the same descriptor, by shdr_conv2d_wgrad_f32. */
#define SHDR_ALGO_MFMA_F16  4  /* force the MFMA path with fp16 operands                 */
#define SHDR_ALGO_MFMA_BF16 5  /* force the MFMA path with bf16 operands                 */
#define SHDR_ALGO_AUTO_F16  6  /* AUTO; fp16 operands wherever the MFMA path is taken    */
#define SHDR_ALGO_AUTO_BF16 7  /* AUTO; bf16 operands wherever the MFMA path is taken    */
#define SHDR_ALGO_AUTO_EXACT 8 /* AUTO without the split-operand fp16 kernels (SHDR_PLAN_X3): every product an fp32 FMA / fp32 MFMA */

const char* shdr_last_error(void);
/* library / code-object version string, e.g. "libshdr 0.1 gfx950" */
const char* shdr_version(void);

/* CRC-32C (Castagnoli) of `n` bytes, continuing from `crc` (0 to start).  Host-side helper of the TensorFlow
 * tensor-bundle checkpoint reader / writer (the format tf_utils.py:149-169 saves). */
uint32_t shdr_crc32c(const void* data, uint64_t n, uint32_t crc);

/* TF 'SAME' rule: out = ceil(in/stride); total = max((out-1)*stride+k-in,0);
 * *pad_before = total/2 (extra cell goes to the bottom/right). */
int shdr_same_pad(int in_size, int k, int stride, int* out_size, int* pad_before);

/*
 * Convolution forward with fused prologue/epilogue.
 * Replaces tf.keras.layers.Conv2D / tf.nn.conv2d (+bias_add, activation,
 * inference BatchNormalization, residual add, channel concat) at
 *   dequantization_net.py:8-9,21-22,27,35-36,46,62-63
 *   refinement_net.py:8-9,21-22,27,35-36,47,63-66
 *   linearization_net.py:12-25,55-64,91-92,28-48,67-83
 *   hallucination_net.py:47-48,63-65,81,87-89,97,101-105,121-123,140-142
 *   vgg16.py:33-35
 *
 * Input  = channel-concat [x1 (C1 ch), x2_scale * x2 (C2 ch)]   (x2 may be NULL, C2 = 0)
 * y[n,oh,ow,co] = act2( affine( act1( conv + bias ) ) + residual ),  co < cout_valid
 *   affine(v) = v*scale[co] + shift[co]      (folded inference BN; scale==NULL -> identity)
 *   residual  = res[n,oh,ow,co] read with channel stride res_cstride (NULL -> 0)
 */
typedef struct shdr_conv2d_desc {
  int32_t N, H, W;          /* input batch / height / width                     */
  int32_t C1, C2;           /* channels of x1 and x2                            */
  int32_t Cout, KH, KW;
  int32_t stride;           /* same in both directions                          */
  int32_t pad_t, pad_l;     /* zero padding before (top / left)                 */
  int32_t Ho, Wo;           /* output height / width                            */
  float   x2_scale;         /* multiplier applied to x2 (hallucination_net.py:101) */
  int32_t act1, act2;       /* SHDR_ACT_*                                        */
  int32_t res_cstride;      /* channels per pixel of the residual tensor         */
  int32_t y_cstride;        /* channels per pixel of y (0 -> cout_valid)          */
  int32_t algo;             /* SHDR_ALGO_*                                       */
  int32_t cout_valid;       /* channels stored to y (0 -> Cout).  Cout may be a  */
                            /* zero-padded filter width (multiple of 16) so that */
                            /* e.g. a 3-channel head runs on the MFMA tile       */
  int64_t w_batch_stride;   /* filter elements between consecutive images; 0 =   */
                            /* one filter for the whole batch.  Used by the      */
                            /* Winograd path: 16 GEMMs with 16 filters, 1 launch */
  /* Strided placement of the output (0 / 1 = dense [N,Ho,Wo,...]): pixel (oh, ow) is written at (oh * y_pix_stride + y_off_h,
   * ow * y_pix_stride + y_off_w) of a [N, y_H, y_W, ...] tensor; the residual is read at the same place.  Used by the input
   * gradient of the stride-2 convolutions (linearization_net.py:12,16,91): their phases are written in place, interleaved. */
  int32_t y_pix_stride, y_off_h, y_off_w, y_H, y_W;
  /* Operator fused IN FRONT of the convolution (SHDR_PROLOGUE_*, forward calls that take a prepared filter only).
   * SHDR_PROLOGUE_BILINEAR2X: x1 is the LOW-RES tensor [N, H/2, W/2, C1] and the convolution runs on
   * tf.image.resize(x1, 2x, BILINEAR) (hallucination_net.py:86-88, dequantization_net.py:25-27); H, W stay the dimensions of the
   * convolution's input, i.e. of the up-sampled image.  On the fused Winograd plan the up-sampled tensor never exists in HBM;
   * every other plan materialises it in the workspace (shdr_conv2d_workspace_bytes_f32 accounts for it). */
  int32_t prologue;
  /* Kind of the optional 2x2 / stride-2 pooled second output y_pool of the forward calls that take a prepared filter:
   * SHDR_POOL_MAX (MaxPool2D(2): hallucination_net.py:47-49, vgg16.py:72-83) or SHDR_POOL_AVG (AveragePooling2D(2):
   * dequantization_net.py:9-10, the encoder of both U-Nets pools each level's output for the next one).  Written by the conv
   * kernel's own epilogue on the fused Winograd (max), split-operand and narrow split-operand plans, by a pooling launch otherwise. */
  int32_t pool;
} shdr_conv2d_desc;
enum { SHDR_POOL_MAX = 0, SHDR_POOL_AVG = 1 };
enum { SHDR_PROLOGUE_NONE = 0, SHDR_PROLOGUE_BILINEAR2X = 1,
       /* split-operand kernel only (shdr_conv2d_fwd_x3_f32): x1 is scaled in the kernel by the power of two that brings max |x1| -- written
        * into the prepared filter's header by shdr_conv2d_x3_input_absmax_f32 -- into the fp16 range; for inputs far below it (gradients) */
       SHDR_PROLOGUE_RANGE_SCALE = 2 };

int shdr_conv2d_fwd_f32(const shdr_conv2d_desc* d,
                        const float* x1, const float* x2, const float* w,
                        const float* bias, const float* scale, const float* shift,
                        const float* residual, float* y, void* stream);

/* the same; y_range (or NULL) = range slot that receives max |y| from the kernel's epilogue (see shdr_conv2d_fwd_prepared_ranged_f32) */
int shdr_conv2d_fwd_yrange_f32(const shdr_conv2d_desc* d,
                               const float* x1, const float* x2, const float* w,
                               const float* bias, const float* scale, const float* shift,
                               const float* residual, float* y, float* y_range, void* stream);

/* ---- dispatch below the ABI: plan / prepare / run.  algo = SHDR_ALGO_AUTO resolves to one of these kernel families from the
 *      descriptor alone (and from whether a residual is fused), the same way in all the calls below. */
enum {
  SHDR_PLAN_DIRECT = 0,           /* VALU direct convolution (shapes the MFMA tile cannot take)                       */
  SHDR_PLAN_MFMA = 1,             /* implicit GEMM: LDS-DMA / register-staged / register-A kernel, chosen by shape    */
  SHDR_PLAN_WINOGRAD_FUSED = 2,   /* one-kernel Winograd F(2x2,3x3) (3x3 stride 1 SAME, Cin % 8 == 0, Cout % 64 == 0) */
  SHDR_PLAN_WINOGRAD_PLANES = 3,  /* three-kernel Winograd for wide layers whose Cout is not a multiple of 64         */
  SHDR_PLAN_X3 = 4,               /* 3x3 stride 1, C % 32 == 0 per source, Cout % 64 == 0, enough tiles to fill the chip: fp32 operands split
                                     into two fp16 terms, three v_mfma_f32_16x16x32_f16 per product, fp32 accumulation (conv_x3.hip) */
  SHDR_PLAN_X3N = 5               /* the same arithmetic for the narrow layers (Cout 16 / 32, <= 32 channels per tap; conv_x3n.hip) */
};
int shdr_conv2d_plan_f32(const shdr_conv2d_desc* d, int has_residual);
/* The prepared form of the HWIO filter `w` for that plan: the packed Winograd transform U = G g G^T, or the plain filter with
 * x2_scale (hallucination_net.py:101) folded into the rows of the second source.  Prepare once per filter version. */
int64_t shdr_conv2d_prepared_filter_elems_f32(const shdr_conv2d_desc* d, int has_residual);
int shdr_conv2d_filter_is_plain_f32(const shdr_conv2d_desc* d, int has_residual);   /* 1: `w` itself is the prepared filter */
int shdr_conv2d_prepare_filter_f32(const shdr_conv2d_desc* d, int has_residual, const float* w, float* prepared, void* stream);
/* caller-provided scratch of shdr_conv2d_fwd_prepared_f32 (0 for every plan but WINOGRAD_PLANES) */
int64_t shdr_conv2d_workspace_bytes_f32(const shdr_conv2d_desc* d, int has_residual);
/* y = act2(affine(act1(conv + bias)) + residual) with the planned kernel; y_pool (optional) = MaxPool2D(2)(y) or, with
 * desc.pool = SHDR_POOL_AVG, AveragePooling2D(2)(y) -- written by the same launch on the fused Winograd (max only) and
 * split-operand plans (where y itself may then be NULL on the wide ones), by a pooling launch otherwise
 * (hallucination_net.py:47-49,63-66: the conv + max-pool pairs of the encoder). */
int shdr_conv2d_fwd_prepared_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared,
                                 const float* bias, const float* scale, const float* shift, const float* residual, float* y,
                                 float* y_pool, void* workspace, void* stream);
/* The same with RANGE SLOTS (see shdr_conv2d_fwd_x3_f32): x1_range / x2_range = device words holding an upper bound of max |x| of the
 * sources, or NULL = unknown.  The split-operand plans (SHDR_PLAN_X3 / X3N) scale their input by it; an unknown range is MEASURED here
 * (one absmax pass over that source into the tail of the workspace: shdr_conv2d_workspace_bytes_f32 accounts for it), so these plans
 * are as range-safe as the fp32 kernels whatever the caller passes -- shdr_conv2d_fwd_prepared_f32 is this call with three NULLs.
 * y_range (or NULL): the slot that receives max |y| (atomicMax, the caller zeroes it) -- from the kernel's own epilogue on the
 * split-operand plans, from one pass over y otherwise -- to be handed to the consumer of y (a pooled / resized / clipped copy of y
 * has the same bound).  Replaces the fp32 Conv2D call sites of hallucination_net.py:43-75,146-190 and vgg16.py:33-35 at full range. */
int shdr_conv2d_fwd_prepared_ranged_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared,
                                        const float* bias, const float* scale, const float* shift, const float* residual, float* y,
                                        float* y_pool, void* workspace, const float* x1_range, const float* x2_range, float* y_range,
                                        void* stream);
/* The planned forward call with a PROJECTED output: y_proj[n,h,w,j] = sum_c proj[j][c] * y[n,h,w,c] (proj: [3][Cout] floats, y_proj:
 * [N,Ho,Wo,3]), formed in the epilogue that holds y in registers; y and y_pool are written only when given (either may be NULL).
 * Replaces the pair "3x3 conv, then a 1x1 conv to 3 channels" where the 1x1 map is linear in the conv's output: the tail of the
 * Hallucination-Net (hallucination_net.py:179-185: skip layer s1 on concat[u1, d1 / 255], then conv2 -- two linear maps in a row)
 * needs only such a projection of u1 and of d1, whose full-resolution 64-channel tensors are then never written nor read back.
 * shdr_conv2d_projected_ok_f32: 1 if the layer's planned kernel can do it (the split-operand plan, 64 output channels, stride 1);
 * the call refuses other layers with SHDR_E_SHAPE (run the two convolutions then). */
int shdr_conv2d_projected_ok_f32(const shdr_conv2d_desc* d);
int shdr_conv2d_fwd_prepared_projected_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared,
                                           const float* bias, const float* scale, const float* shift, const float* proj, float* y_proj,
                                           float* y, float* y_pool, void* workspace, const float* x1_range, const float* x2_range,
                                           float* y_range, void* stream);

/*
 * Convolution backward (GradientTape.gradient through Conv2D: joint_training.py:185,
 * train.py:175,195,242, finetune_real_dataset.py:177).
 *
 * Weight gradient w.r.t. the rows of source `which` (0: x1, 1: x2) of the forward conv `d`:
 *   dw[kh][kw][ci_off + ci][co] += scale * sum_{n,oh,ow} x[n, oh*s+kh-pad_t, ow*s+kw-pad_l, ci] * dz[n,oh,ow,co]
 * (ci_off = 0 / C1, scale = 1 / x2_scale).  dz = gradient w.r.t. the pre-activation conv output,
 * [N,Ho,Wo,cout_valid]; dw = the full [KH,KW,C1+C2,cout_valid] gradient, ACCUMULATED into (the
 * caller zeroes it).  fp32 atomics: the summation order over pixel slices is not fixed.
 */
int shdr_conv2d_wgrad_f32(const shdr_conv2d_desc* d, const float* x, int which, const float* dz,
                          float* dw, void* stream);
/* dgrad filter: wt[kh][kw][co][ci - c_begin] = scale * w[KH-1-kh][KW-1-kw][ci][co], ci in
 * [c_begin, c_begin + c_count): the input gradient of a stride-1 SAME conv is
 * shdr_conv2d_fwd_f32(dz, wt) with the same padding. */
int shdr_filter_transform_f32(const float* w, float* wt, int KH, int KW, int Cin, int Cout,
                              int c_begin, int c_count, float scale, void* stream);
/* Input gradient of the forward conv `d` w.r.t. source `which` (0: x1, 1: x2): dx [N,H,W,C_which] from dz [N,Ho,Wo,cout_valid] and
 * the forward HWIO filter w [KH,KW,C1+C2,Cout] (GradientTape.gradient through Conv2D, joint_training.py:185).  Stride 1 (odd
 * filters; the fused Winograd kernel where the transposed layer qualifies), 1x1 stride 2 (linearization_net.py:12,16: coarse-grid
 * conv written to every second pixel) and general stride 2 (linearization_net.py:91: polyphase form, four stride-1 convs written in
 * place).  Narrow tensors are zero-padded onto the MFMA tile inside.  workspace: shdr_conv2d_dgrad_workspace_bytes_f32 bytes. */
int64_t shdr_conv2d_dgrad_workspace_bytes_f32(const shdr_conv2d_desc* d, int which);
int shdr_conv2d_dgrad_f32(const shdr_conv2d_desc* d, int which, const float* dz, const float* w, float* dx, void* workspace,
                          void* stream);
/* The same with range slots (see shdr_conv2d_fwd_prepared_ranged_f32): dz_range = upper bound of max |dz| or NULL (then measured where the
 * split-operand kernels need it: one pass over dz); dx_range = slot that receives max |dx| from the kernel's epilogue, allowed where
 * shdr_conv2d_dgrad_tracks_range_f32 answers 1 (stride-1 layers on the split-operand / exact MFMA kernels).  A chain conv <- activation <- conv
 * of a backward pass hands the slot on (|act'(.) dz| <= |dz|) and never measures. */
int shdr_conv2d_dgrad_tracks_range_f32(const shdr_conv2d_desc* d, int which);
int shdr_conv2d_dgrad_ranged_f32(const shdr_conv2d_desc* d, int which, const float* dz, const float* w, float* dx, void* workspace,
                                 const float* dz_range, float* dx_range, void* stream);
/* workspace a caller must provide for one call of an op: arg = has_residual (CONV2D_FWD), source (CONV2D_DGRAD,
 * CONV2D_WGRAD_WINOGRAD: the dU scratch), channel count (BATCHNORM: the double-precision partial sums) */
enum { SHDR_OP_CONV2D_FWD = 0, SHDR_OP_CONV2D_DGRAD = 1, SHDR_OP_CONV2D_WGRAD_WINOGRAD = 2, SHDR_OP_BATCHNORM = 3,
       SHDR_OP_ACT_BWD_BIAS = 4 /* arg = channels: one row of C floats per reduction block; optional (ws = NULL: atomics) */ };
/* BATCHNORM workspace = 2 C (1 + SHDR_BN_MAX_BLOCKS) doubles: the sums and one partial row per reduction block (the fp16 kernels
 * reduce without atomics and without a memset; the fp32 kernels use the first 2 C doubles) */
#define SHDR_BN_MAX_BLOCKS 1024
int64_t shdr_workspace_bytes(int op, const shdr_conv2d_desc* d, int arg);
/* db[c] += sum_p dz[p][c] (bias gradient; the caller zeroes db). */
int shdr_bias_grad_f32(const float* dz, float* db, int64_t npix, int C, void* stream);

/* Spatial-aware soft histogram, parametric bin count
 * (linearization_net.py:336-350).  x [npix, C] -> y [npix, B*C], channel
 * order [bin1.c0..c(C-1), bin2...].  Bit-exact w.r.t. the IEEE fp32
 * evaluation of the reference formula. */
int shdr_soft_hist_fwd_f32(const float* x, float* y, int64_t npix, int C, int B, void* stream);
/* its gradient: dx[p][c] = sum_i dy[p][(i-1)*C + c] * (-+B inside the support of bin i)  (tape through linearization_net.py:336-350) */
int shdr_soft_hist_bwd_f32(const float* x, const float* dy, float* dx, int64_t npix, int C, int B, void* stream);

/* Linearization-Net front end (linearization_net.py:310-322):
 * y = concat[img(3), sobel(6), hist4(12), hist8(24), hist16(48)] zero-padded
 * to y_channels (93 <= y_channels, y_channels % 4 == 0 recommended: 96). */
int shdr_lin_frontend_fwd_f32(const float* img, float* y, int N, int H, int W,
                              int y_channels, void* stream);

/* AveragePooling2D(2,2) VALID (dequantization_net.py:10); H, W even. */
int shdr_avgpool2_fwd_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* MaxPool2D(2,2,SAME) on even dims (hallucination_net.py:49,66; vgg16.py:54). */
int shdr_maxpool2_fwd_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* MaxPool2D(3,3,stride 2,SAME) (linearization_net.py:94). */
int shdr_maxpool3s2_fwd_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* tf.image.resize(x, 2x, BILINEAR), half-pixel centres
 * (dequantization_net.py:25, hallucination_net.py:86). */
int shdr_resize2x_fwd_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* tf.reduce_mean(x,[1,2]) (linearization_net.py:118): x [N,HW,C] -> y [N,C]. */
int shdr_gap_fwd_f32(const float* x, float* y, int N, int HW, int C, void* stream);

/* Dense(11) + EMoR PCA decode (linearization_net.py:185-192, 231-253):
 * out[b,k] = g0[k] + sum_j hinv[k,j] * (feat[b,:] @ wfc[:,j] + bfc[j]).
 * table = [K,12] (col 0 = g0, cols 1..11 = hinv). */
int shdr_invcrf_decode_fwd_f32(const float* feat, const float* wfc, const float* bfc,
                               const float* table, float* out, int B, int F, int K,
                               void* stream);
/* linearization_net.py:368-392 (_increase): rf [B,K] -> monotone CDF [B,K]. */
int shdr_increase_fwd_f32(const float* rf, float* out, int B, int K, void* stream);
/* tf_utils.apply_rf (tf_utils.py:54-105): x [B, n_per_batch], rf [B,K]. */
int shdr_apply_rf_fwd_f32(const float* x, const float* rf, float* y, int B,
                          int64_t n_per_batch, int K, void* stream);

/* Elementwise glue of the step closures ---------------------------------- */
/* tf.clip_by_value (test_real_refinement.py:91). */
int shdr_clip_fwd_f32(const float* x, float* y, int64_t n, float lo, float hi, void* stream);
/* x*255, RGB->BGR, subtract VGG mean (hallucination_net.py:149-153, vgg16.py:101-109).
 * x [npix,3] -> y [npix,out_channels], out_channels 3 or 4 (4: zero 4th channel). */
int shdr_vgg_preprocess_fwd_f32(const float* x, float* y, int64_t npix, int out_channels,
                                void* stream);
/* channel reversal of 3-channel pixels (tf_utils.py:5-13). */
int shdr_reverse3_fwd_f32(const float* x, float* y, int64_t npix, void* stream);
/* alpha = clamp((max_c b - 1 + thr)/thr, 0, 1); a = b + alpha * reverse3(hal)
 * (test_real_refinement.py:98-105, joint_training.py:141-145,165).
 * alpha_out (npix floats) may be NULL. */
int shdr_alpha_blend_fwd_f32(const float* b, const float* hal, float* a, float* alpha_out,
                             int64_t npix, float thr, void* stream);
/* concat of up to four 3-channel images into out_channels (>= 3*nsrc, zero
 * padded) -- the Refinement-Net input tf.concat([A,B,C],-1)
 * (test_real_refinement.py:108). */
int shdr_pack3_fwd_f32(const float* s0, const float* s1, const float* s2, const float* s3,
                       int nsrc, float* y, int out_channels, int64_t npix, void* stream);
/* y[p][c] = c < Cin ? x[p][c] : 0 for c < Cout (zero channel padding for the MFMA / DMA tiles). */
int shdr_pad_channels_f32(const float* x, float* y, int64_t npix, int Cin, int Cout, void* stream);
/* log(1+10x)/log(11) (joint_training.py:166,173). */
int shdr_logc_fwd_f32(const float* x, float* y, int64_t n, void* stream);
/* y = act(x * scale[c] + shift[c] + residual) on [npix, C] (scale / shift / residual may be null): the inference
 * BatchNormalization + residual join + relu of linearization_net.py:13-47,55-82 and hallucination_net.py:88-89,164-165 as a
 * stand-alone op, for a `net(x, training=False)` call recorded on a gradient tape (frozen BatchNorm statistics). */
int shdr_affine_act_f32(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                        int64_t npix, int C, int act, void* stream);

/* ---- backward / training (GradientTape.gradient + Adam.apply_gradients:
 *      joint_training.py:185-186, train.py:175-176,195-196,242-243,
 *      finetune_real_dataset.py:177-178).  Parameter gradients (dw, db, dgamma,
 *      dbeta, dwfc, dbfc, drf) are ACCUMULATED into buffers the caller zeroes. ---- */
/* dx = dy * act'(.) evaluated from the activation OUTPUT y (relu / lrelu 0.1 / tanh). */
int shdr_act_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, int act, void* stream);
/* tf.clip_by_value gradient: passes where lo <= x <= hi. */
int shdr_clip_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, float lo, float hi, void* stream);
int shdr_add_f32(const float* a, const float* b, float* y, int64_t n, void* stream);
/* the same; y_range (or NULL) = range slot that receives max |y|: the sums of a backward pass (a forked tensor's gradients, a residual
 * join) feed split-operand input / weight gradients, which then need not measure them */
int shdr_add_ranged_f32(const float* a, const float* b, float* y, int64_t n, float* y_range, void* stream);
/* input gradients of the pooling / resize ops; N,H,W,C describe the op's INPUT x. */
int shdr_avgpool2_bwd_f32(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
int shdr_maxpool2_bwd_f32(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* y = the forward's pooled output (an element takes a window's gradient only where it equals y and no
 * earlier element of the window does). */
int shdr_maxpool3s2_bwd_f32(const float* x, const float* y, const float* dy, float* dx, int N, int H, int W, int C, void* stream);
int shdr_resize2x_bwd_f32(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* the same; dx_range (or NULL) = range slot that receives max |dx| (shdr_add_ranged_f32) */
int shdr_resize2x_bwd_ranged_f32(const float* dy, float* dx, int N, int H, int W, int C, float* dx_range, void* stream);
int shdr_gap_bwd_f32(const float* dy, float* dx, int N, int HW, int C, void* stream);
/* dx[n,2i,2j,:] = dy[n,i,j,:], zero elsewhere (input gradient of a 1x1 stride-2 conv). */
int shdr_upsample_zero2_f32(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* BatchNormalization, training mode (linearization_net.py:13-25, hallucination_net.py:82,122,141):
 * batch mean / biased variance over (N,H,W) accumulated in double (ws = caller workspace of
 * shdr_workspace_bytes(SHDR_OP_BATCHNORM, NULL, C) bytes: the 2*C sums + one partial row per reduction
 * block, no atomics); optional Keras moving-average update (momentum 0.99, unbiased variance). */
int shdr_bn_stats_f32(const float* x, double* ws, float* mean, float* var, float* moving_mean,
                      float* moving_var, int64_t npix, int C, float momentum, void* stream);
int shdr_bn_train_apply_f32(const float* x, const float* mean, const float* var, const float* gamma,
                            const float* beta, float* y, int64_t npix, int C, float eps, int relu,
                            void* stream);
/* the same; y_range (or NULL) = range slot that receives max |y| (see shdr_conv2d_fwd_prepared_ranged_f32): the consumer of a
 * training-mode BatchNorm output is usually a split-operand convolution, which then need not measure its input */
int shdr_bn_train_apply_ranged_f32(const float* x, const float* mean, const float* var, const float* gamma, const float* beta, float* y,
                                   int64_t npix, int C, float eps, int relu, float* y_range, void* stream);
/* y_relu != NULL: the forward applied relu after BN; dy is masked by y_relu > 0 first. */
int shdr_bn_bwd_f32(const float* dy, const float* x, const float* y_relu, const float* mean,
                    const float* var, const float* gamma, double* ws, float* dgamma, float* dbeta,
                    float* dx, int64_t npix, int C, float eps, void* stream);
/* the same; dx_range (or NULL) = range slot that receives max |dx| */
int shdr_bn_bwd_ranged_f32(const float* dy, const float* x, const float* y_relu, const float* mean,
                           const float* var, const float* gamma, double* ws, float* dgamma, float* dbeta,
                           float* dx, int64_t npix, int C, float eps, float* dx_range, void* stream);
int shdr_invcrf_decode_bwd_f32(const float* dinv, const float* feat, const float* wfc,
                               const float* table, float* dfeat, float* dwfc, float* dbfc, int B,
                               int F, int K, void* stream);
int shdr_increase_bwd_f32(const float* rf, const float* dout, float* drf, int B, int K, void* stream);
/* drf accumulated; dx may be NULL (joint training feeds data, not a prediction). */
int shdr_apply_rf_bwd_f32(const float* x, const float* rf, const float* dy, float* drf, float* dx,
                          int B, int64_t n_per_batch, int K, void* stream);
/* per-sample mean loss out[b] = mean((a-b)^2) (mode 0, tf_utils.py:110-111) or mean|a-b| (mode 1,
 * joint_training.py:169-173) and its gradient da = g[b] * d out[b] / d a. */
int shdr_diff_loss_f32(const float* a, const float* b, float* out, int B, int64_t n_per_sample,
                       int mode, void* stream);
int shdr_diff_loss_bwd_f32(const float* a, const float* b, const float* g, float* da, int B,
                           int64_t n_per_sample, int mode, int accumulate, void* stream);
/* batch-global TV loss with SYMMETRIC pad (joint_training.py:175-179): out[0]. */
int shdr_tv_loss_f32(const float* y, float* out, int N, int H, int W, int C, void* stream);
int shdr_tv_loss_bwd_f32(const float* y, const float* g, float* dy, int N, int H, int W, int C,
                         int accumulate, void* stream);
int shdr_logc_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, void* stream);
/* alpha[p] = clamp((max_c x - 1 + thr)/thr, 0, 1) (joint_training.py:141-145). */
int shdr_alpha_mask_f32(const float* x, float* alpha, int64_t npix, float thr, void* stream);
/* d hal = reverse3(alpha * dA) for A = B + alpha * reverse3(hal), alpha constant. */
int shdr_alpha_blend_bwd_f32(const float* dA, const float* alpha, float* dhal, int64_t npix, void* stream);
int shdr_vgg_preprocess_bwd_f32(const float* dy, float* dx, int64_t npix, int in_channels, void* stream);
/* Keras Adam on a flat buffer: g' = grad_scale*g; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2;
 * p -= lr_t * m / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller. */
int shdr_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                  float beta2, float eps, float grad_scale, void* stream);

/* ---- Winograd F(2x2,3x3) for wide 3x3 stride-1 SAME convs (hallucination_net.py:47-48,63-65,81,121;
 *      vgg16.py:33): y = output_transform( V[xi] @ U[xi] ),  xi = 0..15.  The 16 GEMMs are ONE call of
 *      shdr_conv2d_fwd_f32 with N=16, H=rows/16, W=16, 1x1, w_batch_stride = Cin*Cout. ---- */
/* u[16][Cin][Cout] = G g G^T of the HWIO 3x3 filter w. */
int shdr_winograd_filter_f32(const float* w, float* u, int Cin, int Cout, void* stream);
/* rows of each V / M plane: N*ceil(H/2)*ceil(W/2) rounded up to 128. */
int64_t shdr_winograd_tiles(int N, int H, int W);
/* v[16][rows][C] = B^T d B of the 4x4 patches of x [N,H,W,C] (zero padded). */
int shdr_winograd_input_f32(const float* x, float* v, int N, int H, int W, int C, void* stream);
/* y [N,H,W,C] = act2(affine(act1(A^T m A + bias))), m [16][rows][C]. */
int shdr_winograd_output_f32(const float* m, float* y, const float* bias, const float* scale,
                             const float* shift, int N, int H, int W, int C, int act1, int act2,
                             void* stream);

/* ---- chained fine-tuning step (finetune_real_dataset.py:144-183) ---- */
/* dimg[N,H,W,3] = J^T dF of the front end (identity + REFLECT sobel + soft-histogram slopes). */
int shdr_lin_frontend_bwd_f32(const float* img, const float* dF, float* dimg, int N, int H, int W,
                              int y_channels, void* stream);
/* A = B + alpha(B)*reverse3(hal): gradients w.r.t. B (through the alpha mask too) and hal. */
int shdr_alpha_blend_full_bwd_f32(const float* b, const float* hal, const float* dA, float* dB,
                                  float* dhal, int64_t npix, float thr, void* stream);
/* o_s[p][0..2] = y[p][3s..3s+2], s < nout (backward of pack3; slice of the Refinement-Net input). */
int shdr_unpack3_f32(const float* y, float* o0, float* o1, float* o2, float* o3, int nout,
                     int channels, int64_t npix, void* stream);
/* out[b] = sum_i a[b][i] * (b ? b[b][i] : 1). */
int shdr_sample_dot_f32(const float* a, const float* b, float* out, int B, int64_t n_per_sample, void* stream);
/* out = r / (eps + sum[b]/n) * target and its gradient (finetune_real_dataset.py:170). */
int shdr_mean_norm_fwd_f32(const float* r, const float* sum, float* out, int B, int64_t n_per_sample,
                           float eps, float target, void* stream);
int shdr_mean_norm_bwd_f32(const float* g, const float* sum, const float* gdot, float* dr, int B,
                           int64_t n_per_sample, float eps, float target, void* stream);

/* Backward of  y = act(conv + bias):  dz = dy * act'(y) (from the activation OUTPUT) and db[c] += sum_p dz[p][c] in one
 * pass (the caller zeroes db).  act == SHDR_ACT_NONE: dz is not written (dz = dy) and only db is accumulated.
 * C / 4 must be a power of two <= 256.  Replaces the act_bwd + bias_grad pair of GradientTape.gradient through
 * tf.nn.leaky_relu / relu(conv(x) + b) (dequantization_net.py:13-14, hallucination_net.py:48-52). */
int shdr_act_bwd_bias_f32(const float* dy, const float* y, float* dz, float* db, float* ws, int64_t npix, int C, int act, void* stream);

/* U = G g G^T of a 3x3 HWIO filter in the operand order of the fused kernel (one coalesced float4 per lane, chunk and
 * operand group; layout documented at winograd_filter_packed_kernel).  up: 16*Cin*Cout floats. */
int shdr_winograd_filter_packed_f32(const float* w, float* up, int Cin, int Cout, void* stream);
/* Fused Winograd F(2x2,3x3): 3x3 / stride 1 / SAME convolution with the input transform, the 16 GEMMs and the output
 * transform in one kernel (no V / M planes in HBM).  u = shdr_winograd_filter_packed_f32(w);
 * y = act2(affine(act1(conv + bias))).  Needs Cin % 8 == 0, Cout % 64 == 0.  Same call sites as shdr_conv2d_fwd_f32. */
int shdr_conv2d_winograd_fused_f32(const float* x, const float* u, const float* bias, const float* scale,
                                   const float* shift, float* y, int N, int H, int W, int Cin, int Cout,
                                   int act1, int act2, void* stream);

/* The same kernel on a channel concatenation [x, x2] (the skip connections of the U-Net decoders:
 * dequantization_net.py:50-58, hallucination_net.py:115-144 -- tf.concat + Conv2D): both sources [N,H,W,C1], C2 == C1 (or
 * x2 == NULL, C2 == 0); u packs the filter over all C1 + C2 input channels, x's channels first.
 * y_pool (or NULL): [N,H/2,W/2,Cout] = MaxPool2D(2)(y) written by the same epilogue (H, W even) -- the conv + max_pool
 * pairs of the VGG-shaped encoders (hallucination_net.py:43-75 needs both tensors; vgg16.py:72-83 only the pooled one:
 * y may then be NULL and is not stored). */
int shdr_conv2d_winograd_fused2_f32(const float* x, const float* x2, const float* u, const float* bias,
                                    const float* scale, const float* shift, float* y, float* y_pool, int N, int H, int W,
                                    int C1, int C2, int Cout, int act1, int act2, void* stream);

/* The same kernel with tf.image.resize(x, 2x, BILINEAR) fused in front (hallucination_net.py:86-88, dequantization_net.py:25-27:
 * resize + Conv2D 3x3): x is the LOW-RES tensor [N,H/2,W/2,Cin], H x W (even) the size of the up-sampled image and of
 * y [N,H,W,Cout].  The block stages the low-res patch and expands it in LDS with the arithmetic of shdr_resize2x_fwd_f32; the
 * up-sampled tensor is never written.  Reached through shdr_conv2d_fwd_prepared_f32 with desc.prologue = SHDR_PROLOGUE_BILINEAR2X. */
int shdr_conv2d_winograd_fused_up2_f32(const float* x, const float* u, const float* bias, const float* scale,
                                       const float* shift, float* y, int N, int H, int W, int Cin, int Cout,
                                       int act1, int act2, void* stream);

/* fp32 3x3 / stride-1 / SAME convolution on the fp16 matrix pipe (SHDR_PLAN_X3; csrc/conv_x3.hip): x = xh + xl 2^-11, w 2^S = wh + wl in
 * fp16, x w = xh wh + xl (wh 2^-11) + xh wl accumulated in fp32 -- 3 * 2^-22 relative per product, the level of fp32 rounding itself.
 * Also the 7x7 / stride-2 stem (linearization_net.py:91) as four stride-1 phase launches over the parity-subsampled input, accumulating in y.
 * Needs C1 % 32 == 0, C2 % 32 == 0, Cout % 64 == 0.  RANGE: fp16 overflows at 65504 and loses mantissa below 2^-14, an fp32 convolution
 * (hallucination_net.py:47-48, vgg16.py:33-35) does not -- the _ranged entry points take a RANGE SLOT per source (device word: upper
 * bound of max |x|, from the producer's y_range or shdr_absmax_f32) and scale the input by the power of two that brings it to
 * [2^10, 2^11) while splitting (exact; undone in the epilogue), and can write the range of their own output (y_range, atomicMax; the
 * caller zeroes the slot).  Elements down to 2^-25 of the tensor maximum keep 22 mantissa bits; non-finite inputs give non-finite
 * outputs on their receptive field.  The entry points WITHOUT range slots split the input as it stands (|x| < 65504 required) unless
 * desc.prologue = SHDR_PROLOGUE_RANGE_SCALE names the slot in the prepared filter's header (shdr_conv2d_x3_input_absmax_f32).
 * prepared: shdr_conv2d_x3_filter_elems_f32 floats, written by
 * shdr_conv2d_x3_prepare_filter_f32 (the skip scale of the second source folded in).  y = act2(affine(act1(conv + bias)));
 * y_pool (or NULL) = MaxPool2D(2)(y) from the same epilogue (y may then be NULL); desc.prologue = SHDR_PROLOGUE_BILINEAR2X: x1 is the
 * low-res tensor and the 2x bilinear up-sampling runs inside the kernel's patch loader.
 * Same call sites as shdr_conv2d_winograd_fused2_f32; reached through shdr_conv2d_fwd_prepared_f32 / shdr_conv2d_dgrad_f32. */
int shdr_conv2d_x3_ok_f32(const shdr_conv2d_desc* d);
int64_t shdr_conv2d_x3_filter_elems_f32(const shdr_conv2d_desc* d);
int shdr_conv2d_x3_prepare_filter_f32(const shdr_conv2d_desc* d, const float* w, float* prepared, void* stream);
/* max |x| over n floats -> header of `prepared` (after shdr_conv2d_x3_prepare_filter_f32, before the SHDR_PROLOGUE_RANGE_SCALE launch) */
int shdr_conv2d_x3_input_absmax_f32(const float* x, int64_t n, float* prepared, void* stream);
int shdr_conv2d_fwd_x3_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                           const float* scale, const float* shift, float* y, float* y_pool, void* stream);
int shdr_conv2d_fwd_x3_ranged_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                                  const float* scale, const float* shift, float* y, float* y_pool, const float* x1_range,
                                  const float* x2_range, float* y_range, void* stream);
/* the same with a residual [N,Ho,Wo,res_cstride] added between the affine and act2 (stride-1 layers: the ResNet joins of
 * linearization_net.py:6-48 on the 1x1 layers); reached through shdr_conv2d_fwd_prepared_f32 */
int shdr_conv2d_fwd_x3_residual_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                                    const float* scale, const float* shift, const float* residual, float* y, const float* x1_range,
                                    const float* x2_range, float* y_range, void* stream);
/* the same launch with a projected output (see shdr_conv2d_fwd_prepared_projected_f32); proj [3][64], Cout = 64 */
int shdr_conv2d_fwd_x3_projected_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                                     const float* scale, const float* shift, const float* proj, float* y_proj, float* y, float* y_pool,
                                     const float* x1_range, const float* x2_range, float* y_range, void* stream);
/* range slot of a tensor: *range = max(*range, max |x|) (bit pattern of a non-negative float, atomicMax; the caller zeroes the slot) */
int shdr_absmax_f32(const float* x, int64_t n, float* range, void* stream);

/* The split-operand arithmetic for the NARROW layers (SHDR_PLAN_X3N; csrc/conv_x3n.hip): stride 1, 3x3 / 5x5 / 7x7 SAME, Cout 16 or 32
 * (cout_valid <= Cout stored), one source of 4 ... 32 channels (C1 % 4 == 0) or two of 16 -- the full-resolution layers of the
 * Dequantization- / Refinement-Net U-Nets (dequantization_net.py:8-63).  Whole filter + raw patch in LDS, persistent blocks.
 * y = act2(affine(act1(conv + bias)) + residual).  prepared: shdr_conv2d_x3n_filter_elems_f32 floats from
 * shdr_conv2d_x3n_prepare_filter_f32 (w: HWIO with the descriptor's Cout columns).  Reached through shdr_conv2d_fwd_prepared_f32. */
int shdr_conv2d_x3n_ok_f32(const shdr_conv2d_desc* d);
int64_t shdr_conv2d_x3n_filter_elems_f32(const shdr_conv2d_desc* d);
int shdr_conv2d_x3n_prepare_filter_f32(const shdr_conv2d_desc* d, const float* w, float* prepared, void* stream);
int shdr_conv2d_fwd_x3n_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                            const float* scale, const float* shift, const float* residual, float* y, float* y_pool, void* stream);
int shdr_conv2d_fwd_x3n_ranged_f32(const shdr_conv2d_desc* d, const float* x1, const float* x2, const float* prepared, const float* bias,
                                   const float* scale, const float* shift, const float* residual, float* y, float* y_pool,
                                   const float* x1_range, const float* x2_range, float* y_range, void* stream);

/* Split-operand weight gradient (csrc/wgrad_x3.hip): dW of the fp32 convolution on the fp16 matrix pipe, both operands split --
 * X 2^Tx = Xh + Xl 2^-11, dZ 2^Tz = Zh + Zl 2^-11, three v_mfma_f32_16x16x32_f16 per operand pair into two fp32 accumulator sets, per-product
 * error <= 3 * 2^-22 (GradientTape.gradient w.r.t. the Conv2D kernels, joint_training.py:185-186).  shdr_x3_split_planes_f32 writes the two
 * fp16 planes of a tensor (n % 8 == 0) scaled by the power of two its range slot asks for; shdr_conv2d_wgrad_x3_f32 accumulates
 * dw [KH*KW][C1+C2][Cout] += x_scale * sum_p X[p + tap] dZ[p] for source `which` from the planes and the SAME two slots (any stride / padding of
 * the descriptor; channels of the source and Cout multiples of 64: shdr_conv2d_wgrad_x3_ok_f32). */
int shdr_x3_split_planes_f32(const float* x, int64_t n, const float* range, void* hi, void* lo, void* stream);
int shdr_conv2d_wgrad_x3_ok_f32(const shdr_conv2d_desc* d, int which);
int shdr_conv2d_wgrad_x3_f32(const shdr_conv2d_desc* d, const void* xh, const void* xl, int which, const void* zh, const void* zl,
                             const float* x_range, const float* z_range, float* dw, void* stream);

/* Winograd-domain weight gradient of a 3x3 / stride-1 / SAME convolution (the backward counterpart of the fused Winograd
 * forward): dU[xi] += V[xi]^T Q[xi] over all 2x2 tiles (du: 16*Cx*Cout floats, zeroed by the caller), then
 * dw[3][3][Ct][Cout] (rows ci_off .. ci_off+Cx) += x_scale * G^T dU G.  x [N,H,W,Cx], dz [N,H,W,Cout];
 * Cx % 32 == 0, Cout % 64 == 0.  Same call sites as shdr_conv2d_wgrad_f32. */
int shdr_conv2d_wgrad_winograd_f32(const float* x, const float* dz, float* du, float* dw, int N, int H, int W, int Cx,
                                   int Cout, int Ct, int ci_off, float x_scale, void* stream);

/* ---- inference-tool image plumbing (test_real_refinement.py:119-155) -------------- */
/* y[p][c] = x[p][reverse ? 2-c : c] / 255: the decoded 8-bit image as float in [0,1] (:125). */
int shdr_u8_to_unit_f32(const uint8_t* x, float* y, int64_t npix, int reverse_channels, void* stream);
/* cv2.resize(..., interpolation=cv2.INTER_CUBIC) on NHWC float (:133, :146): A = -0.75 bicubic, half-pixel
 * centres, replicated border, no antialiasing. */
int shdr_resize_cubic_f32(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, void* stream);
/* np.pad(x, pad, 'symmetric') over H and W (:136): y [N, H+2*pad, W+2*pad, C]. */
int shdr_pad_symmetric_f32(const float* x, float* y, int N, int H, int W, int C, int pad, void* stream);
/* float RGB -> Radiance RGBE bytes [npix,4], the pixel conversion of cv2.imwrite("*.hdr") (:150);
 * reverse_channels != 0 reads the pixel as BGR. */
int shdr_rgbe_encode_f32(const float* x, uint8_t* y, int64_t npix, int reverse_channels, void* stream);

/* Host-side Radiance scanline RLE of an RGBE image [height][width][4] (the adaptive run-length form of "*.hdr"
 * files, :150).  Returns the number of bytes written to `out`, or -1 (see shdr_last_error) if `capacity` is less than
 * height * (4 + 4 * (width + width / 127 + 2)). */
int64_t shdr_rgbe_rle_encode(const uint8_t* rgbe, int width, int height, uint8_t* out, int64_t capacity);

/* ---- camera-pipeline simulator (joint_training.py:26-69 `_preprocessing`) ------------- */
/* Philox4x32-10 block function (host): the counter-based generator the noise kernel uses; exported so that tests can
 * check the published known-answer vectors. */
int shdr_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);
/* hdr_t = relu(hdr*t + N(0,1)*(sigma_s*hdr*t) + sigma_c*N(0,1)), sigma_s = 0.08/6*U, sigma_c = 0.005*U per (sample,
 * channel) (:30-40); clipped = min(hdr_t, 1) (:43).  hdr [N,H,W,3], t [N].  Stateless: the same seed gives the same
 * noise field for any launch geometry. */
int shdr_camera_expose_f32(const float* hdr, const float* t, float* hdr_t, float* clipped, int N, int H, int W,
                           uint64_t seed, void* stream);
/* jpeg = tf.cast(tf.image.adjust_jpeg_quality(uint8(round(ldr*255)), quality[n]), float32) / 255 (:46-52) -- a
 * baseline-JPEG round trip (4:2:0, islow DCT, fancy upsampling) in libjpeg's integer arithmetic, bit for bit -- and
 * loss_mask [N] (:54-63; may be NULL).  H % 16 == W % 16 == 0.  ws_planes: N*H*W*3/2 bytes, ws_counts: 2*N int32. */
int shdr_jpeg_round_trip_f32(const float* ldr, const int32_t* quality, float* jpeg, float* loss_mask,
                             uint8_t* ws_planes, int32_t* ws_counts, int N, int H, int W, void* stream);

/* HDR-Real record augmentation (finetune_real_dataset.py:49-61): per sample,
 * y = rot90(flip_left_right(x) if flip[n] else x, k = rot[n]) / divisor on square NHWC images [N,S,S,C]
 * (tf.image.rot90 = counter-clockwise). */
int shdr_flip_rot90_f32(const float* x, float* y, const int32_t* flip, const int32_t* rot, int N, int S, int C,
                        float divisor, void* stream);


/* ---- native-fp16 activation path: BASELINE configs[4] ("finetune_real_dataset.py with Refinement-Net, 1024x1024 tiles,
 *      fp16 MFMA conv path").  Feature maps are NHWC fp16 (`void*` = _Float16 / IEEE binary16) with C % 8 == 0, image-like
 *      3-channel tensors and every parameter / parameter gradient stay fp32, accumulation is fp32.  These are the `_f16` twins of
 *      the entry points above and replace the same TF call sites (finetune_real_dataset.py:144-178). ---- */
/* number of fp16 elements of the packed filter of a [KH,KW,C1+C2,Cout] conv */
int64_t shdr_conv2d_packed_filter_elems_f16(int KH, int KW, int C1, int C2, int Cout);
/* fp32 HWIO filter -> packed fp16 [k-chunk][Cout][32] in the k order of shdr_conv2d_fwd_f16; x2_scale is folded into the rows of
 * the second source (hallucination_net.py:101) */
int shdr_conv2d_pack_filter_f16(const float* w, void* wp, int KH, int KW, int C1, int C2, int Cout, float x2_scale, void* stream);
/* y = act1(conv(concat[x1, x2], wp) + bias): tf.keras.layers.Conv2D forward, and -- run on dZ with the flipped / transposed
 * filter -- GradientTape.gradient w.r.t. its input.  d->C1 % 8 == d->C2 % 8 == 0, d->Cout % 16 == 0; y fp16 [N,Ho,Wo,Cout], or
 * fp32 [N,Ho,Wo,cout_valid] when y_is_f32 (the 3-channel heads); scale / shift / residual / act2 of the desc are not used. */
int shdr_conv2d_fwd_f16(const shdr_conv2d_desc* d, const void* x1, const void* x2, const void* wp, const float* bias, void* y,
                        int y_is_f32, void* stream);
/* dw[kh][kw][ci_off + ci][co] += scale * sum_p x[p + tap][ci] * dz[p][co]  (fp32, accumulated with atomics; the caller zeroes dw).
 * x = source `which` of the forward conv `d` (d->C1 / d->C2 = channels per pixel of the fp16 tensors, possibly zero-padded),
 * dz fp16 with dz_channels per pixel; dw = [KH,KW,c1_rows + c2_rows, d->Cout], of which the first cout_valid columns are written. */
int shdr_conv2d_wgrad_f16(const shdr_conv2d_desc* d, const void* x, int which, const void* dz, int dz_channels, int c1_rows,
                          int c2_rows, float* dw, void* stream);
/* specialised forms the two entry points above dispatch to (exported so that a host can test / force them): the narrow-layer
 * forward with the raw patch and the whole filter resident in LDS, and the all-taps weight gradient of stride-1 layers */
int shdr_conv2d_patch_ok_f16(const shdr_conv2d_desc* d);
int shdr_conv2d_fwd_patch_f16(const shdr_conv2d_desc* d, const void* x1, const void* x2, const void* wp, const float* bias, void* y,
                              int y_is_f32, void* stream);
int shdr_conv2d_w3_ok_f16(const shdr_conv2d_desc* d);       /* wide 3x3 / stride-1 layers: raw patch per 32-channel chunk in LDS */
int shdr_conv2d_fwd_w3_f16(const shdr_conv2d_desc* d, const void* x1, const void* x2, const void* wp, const float* bias, void* y,
                           void* stream);
int shdr_conv2d_wgrad_alltaps_ok_f16(const shdr_conv2d_desc* d, int which, int dz_channels);
int shdr_conv2d_wgrad_alltaps_f16(const shdr_conv2d_desc* d, const void* x, int which, const void* dz, int dz_channels, int c1_rows,
                                  int c2_rows, float* dw, void* stream);
int shdr_cast_f32_to_f16(const float* x, void* y, int64_t n, void* stream);
int shdr_cast_f16_to_f32(const void* x, float* y, int64_t n, void* stream);
/* fp32 [npix, Cin] -> fp16 [npix, Cout] zero-padded (narrow gradients of the 3-channel heads onto 8-channel groups) */
int shdr_pad_channels_f32_to_f16(const float* x, void* y, int64_t npix, int Cin, int Cout, void* stream);
/* tf.concat of up to four 3-channel fp32 images -> fp16 [npix, out_channels] (zero-padded; test_real_refinement.py:108);
 * vgg_preprocess: x*255, RGB->BGR, minus mean of hallucination_net.py:149-153 on the single source */
int shdr_pack3_f16(const float* s0, const float* s1, const float* s2, const float* s3, int nsrc, void* y, int out_channels,
                   int64_t npix, int vgg_preprocess, void* stream);
/* its backward: the first nout 3-channel slices of fp16 y [npix, channels] -> fp32 images */
int shdr_unpack3_f16(const void* y, float* o0, float* o1, float* o2, float* o3, int nout, int channels, int64_t npix,
                     int vgg_preprocess_bwd, void* stream);
/* dz = dy * act'(y) (skipped for act NONE), db[c] += sum_p dz[p][c] (skipped when db is NULL) */
int shdr_act_bwd_bias_f16(const void* dy, const void* y, void* dz, float* db, float* ws, int64_t npix, int C, int act, void* stream);
/* y = a + b, optionally relu (residual joins of linearization_net.py:45-47,80-82; gradient accumulation at fan-out points) */
int shdr_add_f16(const void* a, const void* b, void* y, int64_t n, int relu, void* stream);
int shdr_avgpool2_fwd_f16(const void* x, void* y, int N, int H, int W, int C, void* stream);
int shdr_avgpool2_bwd_f16(const void* dy, void* dx, int N, int H, int W, int C, void* stream);
int shdr_maxpool2_fwd_f16(const void* x, void* y, int N, int H, int W, int C, void* stream);
int shdr_maxpool2_bwd_f16(const void* x, const void* dy, void* dx, int N, int H, int W, int C, void* stream);
int shdr_maxpool3s2_fwd_f16(const void* x, void* y, int N, int H, int W, int C, void* stream);
int shdr_maxpool3s2_bwd_f16(const void* x, const void* y, const void* dy, void* dx, int N, int H, int W, int C, void* stream);
int shdr_resize2x_fwd_f16(const void* x, void* y, int N, int H, int W, int C, void* stream);
int shdr_resize2x_bwd_f16(const void* dy, void* dx, int N, int H, int W, int C, void* stream);
int shdr_upsample_zero2_f16(const void* dy, void* dx, int N, int H, int W, int C, void* stream);
/* tf.reduce_mean(x,[1,2]): fp16 [N,HW,C] -> fp32 [N,C], and its backward (fp32 dy -> fp16 dx) */
int shdr_gap_fwd_f16(const void* x, float* y, int N, int HW, int C, void* stream);
int shdr_gap_bwd_f16(const float* dy, void* dx, int N, int HW, int C, void* stream);
/* training-mode BatchNormalization on fp16 tensors; statistics, parameters and their gradients fp32 (sums in double);
 * ws: shdr_workspace_bytes(SHDR_OP_BATCHNORM, NULL, C) bytes */
int shdr_bn_stats_f16(const void* x, double* ws, float* mean, float* var, float* moving_mean, float* moving_var, int64_t npix, int C,
                      float momentum, void* stream);
int shdr_bn_train_apply_f16(const void* x, const float* mean, const float* var, const float* gamma, const float* beta, void* y,
                            int64_t npix, int C, float eps, int relu, void* stream);
int shdr_bn_bwd_f16(const void* dy, const void* x, const void* y_relu, const float* mean, const float* var, const float* gamma,
                    double* ws, float* dgamma, float* dbeta, void* dx, int64_t npix, int C, float eps, void* stream);
/* Linearization-Net front end (linearization_net.py:310-322) fp32 image -> fp16 [N,H,W,y_channels >= 93], and its backward */
int shdr_lin_frontend_fwd_f16(const float* img, void* y, int N, int H, int W, int y_channels, void* stream);
int shdr_lin_frontend_bwd_f16(const float* img, const void* dF, float* dimg, int N, int H, int W, int y_channels, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* SHDR_H_ */
