"""Tensor-level wrappers around the libshdr C ABI.

PyTorch is used for device memory and streams only: every function here takes
NHWC float32 CUDA(HIP) tensors, allocates the output with torch and launches the
hand-written gfx950 kernel on torch's current stream through ctypes.  There is
no fallback path: CPU tensors or a missing library raise.
"""
import ctypes

import torch

try:
    from . import _lib
except ImportError:  # package directory itself on sys.path (drop-in module layout)
    import _lib

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
ALGO_AUTO, ALGO_MFMA, ALGO_DIRECT, ALGO_MFMA_REG = 0, 1, 2, 3


def same_pad(in_size, k, stride):
    """TF 'SAME': out = ceil(in/s); pad_before = max((out-1)*s + k - in, 0) // 2."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return out, total // 2


def _chk(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s: expected a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s: tensor is on %s -- the SingleHDR hot path runs only on a HIP device "
                           "(no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s: expected float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s: tensor must be contiguous NHWC" % name)
    return t


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _d(t):
    """detach parameters so that raw kernels can consume them"""
    return None if t is None else t.detach()


def conv2d(x, w, bias=None, stride=1, x2=None, x2_scale=1.0, act1=ACT_NONE, scale=None,
           shift=None, residual=None, act2=ACT_NONE, algo=ALGO_AUTO, out=None, cout_valid=None):
    """y = act2(affine(act1(conv(concat[x, x2_scale*x2], w) + bias)) + residual), SAME padding.

    `cout_valid` < w.shape[3] says the filter is zero-padded along Cout (to a multiple of 16 so
    that a narrow head runs on the MFMA tile); only the first `cout_valid` channels are stored."""
    lib = _lib.load()
    x = _chk(_d(x), "x")
    w = _chk(_d(w), "w")
    n, h, wd, c1 = x.shape
    kh, kw, cin, cout_gemm = w.shape
    cout = cout_gemm if cout_valid is None else int(cout_valid)
    c2 = 0
    if x2 is not None:
        x2 = _chk(_d(x2), "x2")
        if x2.shape[:3] != x.shape[:3]:
            raise ValueError("conv2d: x2 spatial shape %s != x %s" % (tuple(x2.shape), tuple(x.shape)))
        c2 = x2.shape[3]
    if cin != c1 + c2:
        raise ValueError("conv2d: filter expects %d input channels, got %d+%d" % (cin, c1, c2))
    ho, pt = same_pad(h, kh, stride)
    wo, pl = same_pad(wd, kw, stride)
    d = _lib.ConvDesc()
    d.N, d.H, d.W, d.C1, d.C2 = n, h, wd, c1, c2
    d.Cout, d.KH, d.KW, d.stride = cout_gemm, kh, kw, stride
    d.cout_valid = cout
    d.pad_t, d.pad_l, d.Ho, d.Wo = pt, pl, ho, wo
    d.x2_scale = float(x2_scale)
    d.act1, d.act2 = act1, act2
    d.algo = algo
    res_cs = 0
    if residual is not None:
        residual = _chk(_d(residual), "residual")
        if tuple(residual.shape[:3]) != (n, ho, wo) or residual.shape[3] < cout:
            raise ValueError("conv2d: residual shape %s incompatible with output [%d,%d,%d,%d]"
                             % (tuple(residual.shape), n, ho, wo, cout))
        res_cs = residual.shape[3]
    d.res_cstride = res_cs
    if out is None:
        out = torch.empty((n, ho, wo, cout), device=x.device, dtype=torch.float32)
    else:
        _chk(out, "out")
        if tuple(out.shape) != (n, ho, wo, cout):
            raise ValueError("conv2d: bad out shape")
    d.y_cstride = cout
    for t, nm, ln in ((bias, "bias", cout), (scale, "scale", cout), (shift, "shift", cout)):
        if t is not None and (_chk(_d(t), nm).numel() != ln):
            raise ValueError("conv2d: %s must have %d elements" % (nm, ln))
    rc = lib.shdr_conv2d_fwd_f32(ctypes.byref(d), _ptr(x), _ptr(x2), _ptr(w), _ptr(_d(bias)),
                                 _ptr(_d(scale)), _ptr(_d(shift)), _ptr(residual), _ptr(out), _stream())
    _lib.check(rc, "shdr_conv2d_fwd_f32")
    return out


def _nhwc_op(fn_name, x, out_shape):
    lib = _lib.load()
    x = _chk(_d(x), "x")
    n, h, w, c = x.shape
    y = torch.empty(out_shape, device=x.device, dtype=torch.float32)
    rc = getattr(lib, fn_name)(_ptr(x), _ptr(y), n, h, w, c, _stream())
    _lib.check(rc, fn_name)
    return y


def avgpool2(x):
    n, h, w, c = x.shape
    return _nhwc_op("shdr_avgpool2_fwd_f32", x, (n, h // 2, w // 2, c))


def maxpool2(x):
    n, h, w, c = x.shape
    return _nhwc_op("shdr_maxpool2_fwd_f32", x, (n, h // 2, w // 2, c))


def maxpool3s2(x):
    n, h, w, c = x.shape
    return _nhwc_op("shdr_maxpool3s2_fwd_f32", x, (n, same_pad(h, 3, 2)[0], same_pad(w, 3, 2)[0], c))


def resize2x(x):
    n, h, w, c = x.shape
    return _nhwc_op("shdr_resize2x_fwd_f32", x, (n, 2 * h, 2 * w, c))


def global_avg_pool(x):
    lib = _lib.load()
    x = _chk(_d(x), "x")
    n, h, w, c = x.shape
    y = torch.empty((n, c), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_gap_fwd_f32(_ptr(x), _ptr(y), n, h * w, c, _stream()), "shdr_gap_fwd_f32")
    return y


def soft_hist(img, max_bin):
    """linearization_net.py:336-350 on [..., C] tensors -> [..., max_bin*C]."""
    lib = _lib.load()
    img = _chk(_d(img), "img")
    c = img.shape[-1]
    npix = img.numel() // c
    y = torch.empty(tuple(img.shape[:-1]) + (int(max_bin) * c,), device=img.device, dtype=torch.float32)
    _lib.check(lib.shdr_soft_hist_fwd_f32(_ptr(img), _ptr(y), npix, c, int(max_bin), _stream()),
               "shdr_soft_hist_fwd_f32")
    return y


def lin_frontend(img, channels=96):
    lib = _lib.load()
    img = _chk(_d(img), "img")
    n, h, w, c = img.shape
    if c != 3:
        raise ValueError("lin_frontend: expected 3 channels, got %d" % c)
    y = torch.empty((n, h, w, channels), device=img.device, dtype=torch.float32)
    _lib.check(lib.shdr_lin_frontend_fwd_f32(_ptr(img), _ptr(y), n, h, w, channels, _stream()),
               "shdr_lin_frontend_fwd_f32")
    return y


def invcrf_decode(feat, wfc, bfc, table):
    lib = _lib.load()
    feat, wfc, bfc, table = (_chk(_d(t), nm) for t, nm in
                             ((feat, "feat"), (wfc, "wfc"), (bfc, "bfc"), (table, "table")))
    b, f = feat.shape
    k = table.shape[0]
    if tuple(wfc.shape) != (f, 11) or bfc.numel() != 11 or tuple(table.shape) != (k, 12):
        raise ValueError("invcrf_decode: expected wfc [F,11], bfc [11], table [K,12]")
    out = torch.empty((b, k), device=feat.device, dtype=torch.float32)
    _lib.check(lib.shdr_invcrf_decode_fwd_f32(_ptr(feat), _ptr(wfc), _ptr(bfc), _ptr(table), _ptr(out),
                                              b, f, k, _stream()), "shdr_invcrf_decode_fwd_f32")
    return out


def increase(rf):
    lib = _lib.load()
    rf = _chk(_d(rf), "rf")
    b, k = rf.shape
    out = torch.empty_like(rf)
    _lib.check(lib.shdr_increase_fwd_f32(_ptr(rf), _ptr(out), b, k, _stream()), "shdr_increase_fwd_f32")
    return out


def apply_rf(x, rf):
    lib = _lib.load()
    x = _chk(_d(x), "x")
    rf = _chk(_d(rf), "rf")
    b = x.shape[0]
    if rf.shape[0] != b:
        raise ValueError("apply_rf: batch mismatch %d vs %d" % (b, rf.shape[0]))
    y = torch.empty_like(x)
    _lib.check(lib.shdr_apply_rf_fwd_f32(_ptr(x), _ptr(rf), _ptr(y), b, x.numel() // b, rf.shape[1],
                                         _stream()), "shdr_apply_rf_fwd_f32")
    return y


def clip(x, lo, hi):
    lib = _lib.load()
    x = _chk(_d(x), "x")
    y = torch.empty_like(x)
    _lib.check(lib.shdr_clip_fwd_f32(_ptr(x), _ptr(y), x.numel(), float(lo), float(hi), _stream()),
               "shdr_clip_fwd_f32")
    return y


def logc(x):
    lib = _lib.load()
    x = _chk(_d(x), "x")
    y = torch.empty_like(x)
    _lib.check(lib.shdr_logc_fwd_f32(_ptr(x), _ptr(y), x.numel(), _stream()), "shdr_logc_fwd_f32")
    return y


def _pix3(x, name):
    x = _chk(_d(x), name)
    if x.shape[-1] != 3:
        raise ValueError("%s: expected 3 channels, got %d" % (name, x.shape[-1]))
    return x, x.numel() // 3


def vgg_preprocess(x, out_channels=3):
    """x*255, RGB->BGR, minus VGG mean; out_channels=4 appends a zero channel (MFMA-friendly)."""
    lib = _lib.load()
    x, npix = _pix3(x, "x")
    y = torch.empty(tuple(x.shape[:-1]) + (out_channels,), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_vgg_preprocess_fwd_f32(_ptr(x), _ptr(y), npix, out_channels, _stream()),
               "shdr_vgg_preprocess_fwd_f32")
    return y


def reverse3(x):
    lib = _lib.load()
    x, npix = _pix3(x, "x")
    y = torch.empty_like(x)
    _lib.check(lib.shdr_reverse3_fwd_f32(_ptr(x), _ptr(y), npix, _stream()), "shdr_reverse3_fwd_f32")
    return y


def alpha_blend(b_pred, hal_bgr, thr=0.12, return_alpha=False):
    lib = _lib.load()
    b_pred, npix = _pix3(b_pred, "b_pred")
    hal_bgr, npix2 = _pix3(hal_bgr, "hal_bgr")
    if npix != npix2:
        raise ValueError("alpha_blend: shape mismatch")
    a = torch.empty_like(b_pred)
    alpha = torch.empty(tuple(b_pred.shape[:-1]) + (1,), device=a.device, dtype=torch.float32) if return_alpha else None
    _lib.check(lib.shdr_alpha_blend_fwd_f32(_ptr(b_pred), _ptr(hal_bgr), _ptr(a), _ptr(alpha), npix,
                                            float(thr), _stream()), "shdr_alpha_blend_fwd_f32")
    return (a, alpha) if return_alpha else a


def pack3(srcs, out_channels=None):
    lib = _lib.load()
    srcs = [_pix3(s, "src%d" % i)[0] for i, s in enumerate(srcs)]
    n = len(srcs)
    if not 1 <= n <= 4:
        raise ValueError("pack3: 1..4 sources")
    if any(s.shape != srcs[0].shape for s in srcs):
        raise ValueError("pack3: shape mismatch")
    oc = out_channels or 3 * n
    y = torch.empty(tuple(srcs[0].shape[:-1]) + (oc,), device=srcs[0].device, dtype=torch.float32)
    p = [_ptr(s) for s in srcs] + [None] * (4 - n)
    _lib.check(lib.shdr_pack3_fwd_f32(p[0], p[1], p[2], p[3], n, _ptr(y), oc, srcs[0].numel() // 3,
                                      _stream()), "shdr_pack3_fwd_f32")
    return y
