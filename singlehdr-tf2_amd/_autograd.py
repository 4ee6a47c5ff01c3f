"""Differentiable counterparts of the `_ops` wrappers.

torch.autograd is used as the tape only (plumbing): every forward AND backward
computation below is a libshdr HIP kernel.  `_ops.<op>` dispatches here when
gradients are enabled and an input requires them; the fused inference
epilogues (folded BatchNorm, residual, second activation) are not
differentiable -- the training-mode network code does not use them.

Replaces tf.GradientTape.gradient for the ops of the hot path
(joint_training.py:147-185, train.py:165-242).
"""
import torch
import torch.nn.functional as F

try:
    from . import _ops as K
except ImportError:
    import _ops as K


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _acc(p):
    """The gradient buffer of a variable that a training step has placed in its flat gradient buffer (pipeline.FlatParams
    marks them): the backward kernels ACCUMULATE into it directly (they add with atomics anyway) and the tape entry returns no
    gradient for it -- no zero-filled temporary and no separate accumulation pass per variable."""
    if p is None or not getattr(p, "_shdr_accum", False) or not p.is_leaf:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


# ---------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------
def _dgrad(dz, w, which, c1, c2, x2_scale, stride, x_shape):
    """input gradient w.r.t. source `which` of the forward filter w: one library call -- the decompositions (filter flip and
    slice, zero-padding of narrow tensors, Winograd, 1x1 / 2 on the coarse grid, polyphase 7x7 / 2) live below the C ABI
    (csrc/conv_plan.hip: shdr_conv2d_dgrad_f32)"""
    return K.conv2d_dgrad(dz, w, x_shape, c1, c2, which, stride, x2_scale)


class Conv2dFn(torch.autograd.Function):
    """y = act1(conv(concat[x, x2_scale*x2], w) + bias)"""

    @staticmethod
    def forward(ctx, x, x2, w, bias, stride, x2_scale, act1, cout_valid, algo):
        wf = w
        if K.PRECISION in ("fp16op", "bf16") and x2 is not None and x2_scale != 1.0:
            # the reduced-precision kernels take x2_scale folded into the x2 rows of the filter
            c1 = x.shape[3]
            wf = torch.cat([w[:, :, :c1], w[:, :, c1:] * x2_scale], dim=2)
        y = K.conv2d(x, wf, bias, stride=stride, x2=x2, x2_scale=x2_scale if wf is w else 1.0, act1=act1, algo=algo,
                     cout_valid=cout_valid)
        ctx.save_for_backward(x, x2, w, y)
        ctx.meta = (stride, x2_scale, act1, bias is not None)
        ctx.bias_ref = bias             # (not a saved tensor: only its .grad buffer is looked up in backward)
        ctx.prec = K.PRECISION          # the backward kernels run at the precision of the forward
        # the range slots of the sources (K: "range slots"): saved tensors come back as fresh objects without their attributes, and the
        # split-operand weight gradient would measure max |x| again
        ctx.ranges = (K._range_of(x), K._range_of(x2)) if x.dtype == torch.float32 else (None, None)
        return y

    @staticmethod
    def backward(ctx, dy):
        with K.precision(ctx.prec):
            return Conv2dFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        x, x2, w, y = ctx.saved_tensors
        for t, r in zip((x, x2), ctx.ranges):
            if t is not None and r is not None:
                K._set_range(t, r)
        stride, x2_scale, act1, has_bias = ctx.meta
        dy = _c(dy)
        need_x, need_x2, need_w, need_b = ctx.needs_input_grad[:4]
        dx = dx2 = dw = db = None
        if need_b and has_bias:
            bacc = _acc(ctx.bias_ref)
            dz, db = K.act_bwd_bias(dy, y, act1, out=bacc)         # one pass: activation backward + bias gradient
            if bacc is not None:
                db = None
        else:
            dz = K.act_bwd(dy, y, act1) if act1 != K.ACT_NONE else dy
        c1 = x.shape[3]
        if need_w:
            kh, kw, cin, cout_gemm = w.shape
            wacc = _acc(w) if dz.shape[3] == cout_gemm else None
            dw = K.conv2d_wgrad(x, x2, dz, (kh, kw, cin, dz.shape[3]), stride, x2_scale, out=wacc)
            if wacc is not None:
                dw = None
            elif dz.shape[3] != cout_gemm:
                dw = F.pad(dw, (0, cout_gemm - dz.shape[3]))
        c2 = 0 if x2 is None else x2.shape[3]
        if need_x:
            dx = _dgrad(dz, w, 0, c1, c2, x2_scale, stride, tuple(x.shape))
        if need_x2 and x2 is not None:
            dx2 = _dgrad(dz, w, 1, c1, c2, x2_scale, stride, tuple(x2.shape))
        return dx, dx2, dw, db, None, None, None, None, None


def conv2d(x, w, bias=None, stride=1, x2=None, x2_scale=1.0, act1=K.ACT_NONE, scale=None, shift=None,
           residual=None, act2=K.ACT_NONE, algo=K.ALGO_AUTO, cout_valid=None):
    if scale is not None or shift is not None or residual is not None or act2 != K.ACT_NONE:
        raise NotImplementedError("the fused inference epilogue (folded BN / residual / act2) is not differentiable; "
                                  "the training path applies these as separate ops")
    return Conv2dFn.apply(x, x2, w, bias, stride, x2_scale, act1, cout_valid, algo)


# ---------------------------------------------------------------------------
# convolution on fp16 feature maps (native-fp16 path of BASELINE configs[4])
# ---------------------------------------------------------------------------
def _dgrad_h(dz, w, c_begin, c_count, scale, stride, x_shape):
    """fp16 input gradient w.r.t. source channels [c_begin, c_begin + c_count) of the forward filter w (fp32 HWIO): the fp16
    conv kernel run on dz with the flipped / transposed filter -- same decomposition as _dgrad (stride 1; 1x1 / 2 on the coarse
    grid; general stride 2 in polyphase form)"""
    if c_count % 16:
        raise NotImplementedError("fp16 input gradient needs a multiple of 16 input channels, got %d" % c_count)
    kh, kw = w.shape[0], w.shape[1]
    cz = dz.shape[3]                                   # channels per pixel of dz (>= the filter's true output channels)
    cols = min(w.shape[3], cz)
    if w.shape[3] != cols:                             # zero-padded filter columns carry no gradient
        w = w[..., :cols].contiguous()
    wt = K.filter_transform(w, c_begin, c_count, scale)            # fp32 [kh, kw, cols, c_count]
    if cz != cols:
        wt = F.pad(wt, (0, 0, 0, cz - cols))

    def run(filt, **kw_):
        return K.conv2d_h(dz, K.pack_filter_h(filt, cz), None, tuple(filt.shape[:2]), filt.shape[3], **kw_)
    if stride == 1:
        return run(wt)
    if stride == 2 and kh == 1 and kw == 1:
        return K.upsample_zero2(run(wt), x_shape)
    if stride == 2:
        n, h, wd, _ = x_shape
        _, pt = K.same_pad(h, kh, 2)
        _, pl = K.same_pad(wd, kw, 2)
        dx = torch.empty((n, h, wd, c_count), device=dz.device, dtype=dz.dtype)

        def phase(par_in, pad_fwd, k):
            par = (par_in + pad_fwd) % 2
            taps = len(range(par, k, 2))
            off = (par_in + pad_fwd - par) // 2
            return k - 1 - par - 2 * (taps - 1), taps - 1 - off, taps
        for p_ in range(2):
            a0, pad_t, th = phase(p_, pt, kh)
            mh = (h - p_ + 1) // 2
            for q_ in range(2):
                b0, pad_l, tw = phase(q_, pl, kw)
                mw = (wd - q_ + 1) // 2
                if mh == 0 or mw == 0:
                    continue
                if th == 0 or tw == 0:
                    dx[:, p_::2, q_::2] = 0.0
                    continue
                dx[:, p_::2, q_::2] = run(wt[a0::2, b0::2].contiguous(), pad=(pad_t, pad_l), out_hw=(mh, mw))
        return dx
    raise NotImplementedError("input gradient of a %dx%d stride-%d convolution is not built" % (kh, kw, stride))


class Conv2dHFn(torch.autograd.Function):
    """y = act1(conv(concat[x, x2_scale * x2], w) + bias) with fp16 x / x2 / y, fp32 w / bias and fp32 parameter gradients.
    A narrow head (cout_valid < w.shape[3], the zero-padded 3-channel outputs) returns fp32."""

    @staticmethod
    def forward(ctx, x, x2, w, bias, stride, x2_scale, act1, cout_valid):
        c1 = x.shape[3]
        c2 = 0 if x2 is None else x2.shape[3]
        wp = K.pack_filter_h(w, c1, c2, x2_scale)
        y = K.conv2d_h(x, wp, bias, tuple(w.shape[:2]), w.shape[3], stride=stride, x2=x2, act1=act1, cout_valid=cout_valid)
        ctx.save_for_backward(x, x2, w, y)
        ctx.meta = (stride, x2_scale, act1, bias is not None, cout_valid)
        ctx.bias_ref = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x2, w, y = ctx.saved_tensors
        stride, x2_scale, act1, has_bias, cout_valid = ctx.meta
        dy = _c(dy)
        need_x, need_x2, need_w, need_b = ctx.needs_input_grad[:4]
        db = None
        if y.dtype == torch.float32:                   # head: the 3-channel gradient is fp32; pad it onto an 8-channel fp16 group
            dz32 = K.act_bwd(dy, y, act1) if act1 != K.ACT_NONE else dy
            if need_b and has_bias:
                bacc = _acc(ctx.bias_ref)
                db = K.bias_grad(dz32, out=bacc)
                if bacc is not None:
                    db = None
            dz = K.pad_channels_h(dz32, 8)
            cols = y.shape[3]
        else:
            bacc = _acc(ctx.bias_ref) if (need_b and has_bias) else None
            dz, db = K.act_bwd_bias_h(dy, y, act1, need_b and has_bias, out=bacc)
            if bacc is not None:
                db = None
            cols = None
        dx = dx2 = dw = None
        c1 = x.shape[3]
        if need_w:
            wacc = _acc(w)
            dw = K.conv2d_wgrad_h(x, x2, dz, tuple(w.shape), stride, x2_scale, cout_valid=cols, out=wacc)
            if wacc is not None:
                dw = None
        if need_x:
            dx = _dgrad_h(dz, w, 0, c1, 1.0, stride, x.shape)
        if need_x2 and x2 is not None:
            dx2 = _dgrad_h(dz, w, c1, x2.shape[3], x2_scale, stride, x2.shape)
        return dx, dx2, dw, db, None, None, None, None


def conv2d_h(x, w, bias=None, stride=1, x2=None, x2_scale=1.0, act1=K.ACT_NONE, cout_valid=None):
    return Conv2dHFn.apply(x, x2, w, bias, stride, x2_scale, act1, cout_valid)


# ---------------------------------------------------------------------------
# BatchNormalization, training mode
# ---------------------------------------------------------------------------
class BatchNormTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum, relu):
        x = _c(x)
        mean, var = K.bn_stats(x, moving_mean, moving_var, momentum)
        y = K.bn_train_apply(x, mean, var, gamma, beta, eps, relu)
        ctx.save_for_backward(x, y if relu else None, mean, var, gamma)
        ctx.eps = eps
        ctx.beta_ref = beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, var, gamma = ctx.saved_tensors
        gacc, bacc = _acc(gamma), _acc(ctx.beta_ref)
        if gacc is None or bacc is None:
            gacc = bacc = None
        dx, dgamma, dbeta = K.bn_bwd(_c(dy), x, y, mean, var, gamma, ctx.eps, gacc, bacc)
        if gacc is not None:
            dgamma = dbeta = None
        return dx, dgamma, dbeta, None, None, None, None, None


def batch_norm_train(x, gamma, beta, moving_mean, moving_var, eps, momentum, relu):
    return BatchNormTrainFn.apply(x, gamma, beta, moving_mean, moving_var, eps, momentum, relu)


class BatchNormFrozenFn(torch.autograd.Function):
    """y = act(gamma * (x - moving_mean) / sqrt(moving_var + eps) + beta [+ residual]): BatchNormalization called with
    training=False while a tape is recording (tf.keras semantics: the moving statistics are constants, gamma and beta still
    receive gradients)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps, residual, act):
        x = _c(x.detach())
        scale = (gamma.detach() * torch.rsqrt(var + eps)).contiguous()       # C-sized vectors: plumbing
        shift = (beta.detach() - mean * scale).contiguous()
        y = K.affine_act(x, scale, shift, None if residual is None else _c(residual.detach()), act)
        ctx.save_for_backward(x, y if act != K.ACT_NONE else None, gamma, mean, var, scale)
        ctx.meta = (eps, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, var, scale = ctx.saved_tensors
        eps, act = ctx.meta
        d = K.act_bwd(_c(dy), y, act) if act != K.ACT_NONE else _c(dy)
        dx = K.affine_act(d, scale) if ctx.needs_input_grad[0] else None
        dgamma = dbeta = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # sum(d * (x - mean) * rstd), sum(d): the reductions of the training-mode backward on the moving statistics
            _, dgamma, dbeta = K.bn_bwd(d, x, None, mean.detach(), var.detach(), gamma, eps)
        return dx, dgamma, dbeta, None, None, None, (d if ctx.needs_input_grad[6] else None), None


def batch_norm_frozen(x, gamma, beta, moving_mean, moving_var, eps, residual=None, act=K.ACT_NONE):
    return BatchNormFrozenFn.apply(x, gamma, beta, moving_mean, moving_var, eps, residual, act)


# ---------------------------------------------------------------------------
# pooling / resize / elementwise
# ---------------------------------------------------------------------------
def _simple(name, fwd, bwd):
    """Function whose backward needs (saved tensors chosen by `fwd`, dy)"""
    def forward(ctx, x, *args):
        y, saved, meta = fwd(x, *args)
        ctx.save_for_backward(*saved)
        ctx.meta = meta
        ctx.nargs = len(args)
        return y

    def backward(ctx, dy):
        return (bwd(ctx.saved_tensors, ctx.meta, _c(dy)),) + (None,) * ctx.nargs

    return type(name, (torch.autograd.Function,), {"forward": staticmethod(forward), "backward": staticmethod(backward)})


AvgPool2Fn = _simple("AvgPool2Fn", lambda x: (K.avgpool2(x), (), tuple(x.shape)),
                     lambda s, m, dy: K.avgpool2_bwd(dy, m))
MaxPool2Fn = _simple("MaxPool2Fn", lambda x: (K.maxpool2(x), (x,), None),
                     lambda s, m, dy: K.maxpool2_bwd(s[0], dy))
def _maxpool3s2_fwd(x):
    y = K.maxpool3s2(x)
    return y, (x, y), None


MaxPool3s2Fn = _simple("MaxPool3s2Fn", _maxpool3s2_fwd, lambda s, m, dy: K.maxpool3s2_bwd(s[0], s[1], dy))
Resize2xFn = _simple("Resize2xFn", lambda x: (K.resize2x(x), (), tuple(x.shape)),
                     lambda s, m, dy: K.resize2x_bwd(dy, m))
GapFn = _simple("GapFn", lambda x: (K.global_avg_pool(x), (), (tuple(x.shape), x.dtype)),
                lambda s, m, dy: K.gap_bwd(dy, m[0], m[1]))
ClipFn = _simple("ClipFn", lambda x, lo, hi: (K.clip(x, lo, hi), (x,), (lo, hi)),
                 lambda s, m, dy: K.clip_bwd(dy, s[0], m[0], m[1]))
LogcFn = _simple("LogcFn", lambda x: (K.logc(x), (x,), None),
                 lambda s, m, dy: K.logc_bwd(dy, s[0]))
Reverse3Fn = _simple("Reverse3Fn", lambda x: (K.reverse3(x), (), None),
                     lambda s, m, dy: K.reverse3(dy))
VggPreFn = _simple("VggPreFn", lambda x, oc, dt: (K.vgg_preprocess(x, oc, dt), (), None),
                   lambda s, m, dy: K.vgg_preprocess_bwd(dy))
IncreaseFn = _simple("IncreaseFn", lambda rf: (K.increase(rf), (rf,), None),
                     lambda s, m, dy: K.increase_bwd(s[0], dy))


def avgpool2(x):
    return AvgPool2Fn.apply(x)


def maxpool2(x):
    return MaxPool2Fn.apply(x)


def maxpool3s2(x):
    return MaxPool3s2Fn.apply(x)


def resize2x(x):
    return Resize2xFn.apply(x)


def global_avg_pool(x):
    return GapFn.apply(x)


def clip(x, lo, hi):
    return ClipFn.apply(x, lo, hi)


def logc(x):
    return LogcFn.apply(x)


def reverse3(x):
    return Reverse3Fn.apply(x)


def vgg_preprocess(x, out_channels=3, dtype=torch.float32):
    return VggPreFn.apply(x, out_channels, dtype)


def increase(rf):
    return IncreaseFn.apply(rf)


class AffineActFn(torch.autograd.Function):
    """y = act(x * scale[c] + shift[c] + residual); scale / shift are constants (folded moving statistics)"""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, act):
        y = K.affine_act(x.detach(), _det(scale), _det(shift), _det(residual), act)
        ctx.save_for_backward(y if act != K.ACT_NONE else None, scale)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        y, scale = ctx.saved_tensors
        d = K.act_bwd(_c(dy), y, ctx.act) if ctx.act != K.ACT_NONE else _c(dy)
        dx = K.affine_act(d, _det(scale)) if (scale is not None and ctx.needs_input_grad[0]) else d
        return dx, None, None, (d if ctx.needs_input_grad[3] else None), None


def _det(t):
    return None if t is None else t.detach()


def affine_act(x, scale, shift, residual, act):
    if (scale is not None and scale.requires_grad) or (shift is not None and shift.requires_grad):
        raise NotImplementedError("affine_act: scale / shift are folded constants, not trainable inputs")
    return AffineActFn.apply(x, scale, shift, residual, act)


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return K.add(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return AddFn.apply(a, b)


class AddReluFn(torch.autograd.Function):
    """relu(a + b) -- the residual join of the Linearization-Net blocks in training mode"""

    @staticmethod
    def forward(ctx, a, b):
        y = K.add(a, b, relu=True)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        d = K.act_bwd(_c(dy), y, K.ACT_RELU)
        return d, d


def add_relu(a, b):
    return AddReluFn.apply(a, b)


class ForkFn(torch.autograd.Function):
    """A fan-out point of the tape: n aliases of x, each for ONE consumer.  autograd would sum the consumers' gradients with its own
    at::add; here the sum is libshdr's add kernel (fp32 or fp16), so the training step stays free of torch arithmetic."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)          # an unused alias contributes nothing: no zero tensor, no add launch for it
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [_c(g) for g in gs if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        for g in gs[1:]:
            acc = K.add(acc, g)
        return acc, None


def fork(x, n=2):
    return ForkFn.apply(x, n)


class MarkFn(torch.autograd.Function):
    """Identity with a callback in its backward: the callback runs (on the autograd thread, torch's current stream = the stream of
    the forward op) once every backward node BEHIND this point of the tape has been enqueued -- the data-parallel steps hang the
    launch of a gradient bucket's all-reduce on it."""

    @staticmethod
    def forward(ctx, x, callback):
        ctx.callback = callback
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.callback()
        return g, None


def mark(x, callback):
    return MarkFn.apply(x, callback)


# ---------------------------------------------------------------------------
# inverse-CRF head, CRF application
# ---------------------------------------------------------------------------
class InvcrfDecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, wfc, bfc, table):
        ctx.save_for_backward(feat, wfc, table)
        return K.invcrf_decode(feat, wfc, bfc, table)

    @staticmethod
    def backward(ctx, dinv):
        feat, wfc, table = ctx.saved_tensors
        dfeat, dwfc, dbfc = K.invcrf_decode_bwd(_c(dinv), feat, wfc, table)
        return dfeat, dwfc, dbfc, None


def invcrf_decode(feat, wfc, bfc, table):
    return InvcrfDecodeFn.apply(feat, wfc, bfc, table)


class ApplyRfFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rf):
        ctx.save_for_backward(x, rf)
        return K.apply_rf(x, rf)

    @staticmethod
    def backward(ctx, dy):
        x, rf = ctx.saved_tensors
        drf, dx = K.apply_rf_bwd(x, rf, _c(dy), ctx.needs_input_grad[0])
        return dx, (drf if ctx.needs_input_grad[1] else None)


def apply_rf(x, rf):
    return ApplyRfFn.apply(x, rf)


# ---------------------------------------------------------------------------
# losses and the blend of the joint step
# ---------------------------------------------------------------------------
class DiffLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mode):
        ctx.save_for_backward(a, b)
        ctx.mode = mode
        return K.diff_loss(a, b, mode)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _c(g)
        da = K.diff_loss_bwd(a, b, g, ctx.mode) if ctx.needs_input_grad[0] else None
        db = K.diff_loss_bwd(b, a, g, ctx.mode) if ctx.needs_input_grad[1] else None
        return da, db, None


def diff_loss(a, b, mode):
    return DiffLossFn.apply(a, b, mode)


class TvLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        ctx.save_for_backward(y)
        return K.tv_loss(y)

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return K.tv_loss_bwd(y, _c(g))


def tv_loss(y):
    return TvLossFn.apply(y)


class BlendConstFn(torch.autograd.Function):
    """A = base + alpha * reverse3(hal); base, alpha are data (joint_training.py:163-165)"""

    @staticmethod
    def forward(ctx, base, alpha, hal_bgr, thr):
        ctx.save_for_backward(alpha)
        return K.alpha_blend(base, hal_bgr, thr)   # recomputes the same alpha from `base`

    @staticmethod
    def backward(ctx, dA):
        (alpha,) = ctx.saved_tensors
        return None, None, K.alpha_blend_bwd(_c(dA), alpha), None


def blend_const(base, alpha, hal_bgr, thr):
    return BlendConstFn.apply(base, alpha, hal_bgr, thr)


# ---------------------------------------------------------------------------
# fine-tuning chain (finetune_real_dataset.py:144-183)
# ---------------------------------------------------------------------------
class LinFrontendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, channels, dtype):
        ctx.save_for_backward(img)
        return K.lin_frontend(img, channels, dtype)

    @staticmethod
    def backward(ctx, dF):
        (img,) = ctx.saved_tensors
        return K.lin_frontend_bwd(img, _c(dF)), None, None


def lin_frontend(img, channels, dtype=torch.float32):
    return LinFrontendFn.apply(img, channels, dtype)


class AlphaBlendFn(torch.autograd.Function):
    """A = B + alpha(B) * reverse3(hal), gradients through B, the alpha mask and hal"""

    @staticmethod
    def forward(ctx, b_pred, hal_bgr, thr):
        ctx.save_for_backward(b_pred, hal_bgr)
        ctx.thr = thr
        return K.alpha_blend(b_pred, hal_bgr, thr)

    @staticmethod
    def backward(ctx, dA):
        b_pred, hal_bgr = ctx.saved_tensors
        dB, dhal = K.alpha_blend_full_bwd(b_pred, hal_bgr, _c(dA), ctx.thr)
        return dB, dhal, None


def alpha_blend(b_pred, hal_bgr, thr):
    return AlphaBlendFn.apply(b_pred, hal_bgr, thr)


class Pack3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out_channels, dtype, *srcs):
        ctx.n = len(srcs)
        return K.pack3(list(srcs), out_channels, dtype)

    @staticmethod
    def backward(ctx, dy):
        return (None, None) + tuple(K.unpack3(_c(dy), ctx.n))


def pack3(srcs, out_channels, dtype=torch.float32):
    return Pack3Fn.apply(out_channels, dtype, *srcs)


class Unpack3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, nout):
        ctx.channels, ctx.dtype = y.shape[-1], y.dtype
        return K.unpack3(y, nout)

    @staticmethod
    def backward(ctx, *grads):
        return K.pack3([_c(g) for g in grads], ctx.channels, ctx.dtype), None


def unpack3(y, nout):
    return Unpack3Fn.apply(y, nout)


class MeanNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, eps, target):
        ssum = K.sample_dot(r)
        ctx.save_for_backward(r, ssum)
        ctx.meta = (eps, target)
        return K.mean_norm_fwd(r, ssum, eps, target)

    @staticmethod
    def backward(ctx, g):
        r, ssum = ctx.saved_tensors
        g = _c(g)
        return K.mean_norm_bwd(g, ssum, K.sample_dot(g, r), *ctx.meta), None, None


def mean_norm(r, eps, target):
    return MeanNormFn.apply(r, eps, target)


import sys  # noqa: E402

K.AUTOGRAD = sys.modules[__name__]
