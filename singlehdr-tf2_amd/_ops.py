"""Tensor-level wrappers around the libshdr C ABI.

PyTorch is used for device memory and streams only: every function here takes
NHWC float32 CUDA(HIP) tensors, allocates the output with torch and launches the
hand-written gfx950 kernel on torch's current stream through ctypes.  There is
no fallback path: CPU tensors or a missing library raise.
"""
import collections
import ctypes

import torch

try:
    from . import _lib
except ImportError:  # package directory itself on sys.path (drop-in module layout)
    import _lib

AUTOGRAD = None   # set by _autograd on import: module with the differentiable counterparts


def _needs_grad(*tensors):
    return AUTOGRAD is not None and torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


WINOGRAD = True   # False: ask the library for the plain (non-Winograd) kernels only (experiments, tests)


ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
ALGO_AUTO, ALGO_MFMA, ALGO_DIRECT, ALGO_MFMA_REG = 0, 1, 2, 3
PROLOGUE_NONE, PROLOGUE_BILINEAR2X = 0, 1      # shdr_conv2d_desc.prologue
POOL_MAX, POOL_AVG = 0, 1                      # shdr_conv2d_desc.pool
ALGO_MFMA_F16, ALGO_MFMA_BF16, ALGO_AUTO_F16, ALGO_AUTO_BF16 = 4, 5, 6, 7
ALGO_AUTO_EXACT = 8    # AUTO without the split-operand fp16 kernel (plan "x3"): every product an fp32 FMA / fp32 MFMA
EXACT_FP32 = False     # True: ask the library for ALGO_AUTO_EXACT wherever this module would ask for ALGO_AUTO


def _auto(algo):
    return ALGO_AUTO_EXACT if (algo == ALGO_AUTO and EXACT_FP32) else algo

# Precision of the conv path.
#   "fp32"  : the parity path (exact-fp32 MFMA, Winograd where it pays).
#   "fp16"  : BASELINE configs[4], NATIVE fp16: the networks turn their inputs into fp16 feature maps (NHWC, C % 8 == 0) and every
#             conv forward / dgrad / wgrad, BatchNorm, pooling, resize and activation-backward pass reads and writes fp16
#             (csrc/conv_f16.hip, wgrad_f16.hip, elem_f16.hip); parameters, their gradients, statistics and the 3-channel
#             image-like tensors stay fp32, accumulation is fp32.  Ops dispatch on the dtype of their input tensor.
#   "fp16op" / "bf16" : the round-1 operand-rounding modes -- fp32 tensors in HBM, operands rounded when they are packed for
#             v_mfma_f32_16x16x32_{f16,bf16} (kept for comparison; Winograd is not used then).
PRECISION = "fp32"
_AUTO_ALGO = {"fp32": ALGO_AUTO, "fp16": ALGO_AUTO, "fp16op": ALGO_AUTO_F16, "bf16": ALGO_AUTO_BF16}


def native_fp16():
    """True inside `precision("fp16")`: feature maps are created as fp16"""
    return PRECISION == "fp16"


class precision:
    """with precision("fp16"): ...   -- scoped override of the conv operand precision"""

    def __init__(self, name):
        if name not in _AUTO_ALGO:
            raise ValueError("precision must be one of %s" % sorted(_AUTO_ALGO))
        self._name = name

    def __enter__(self):
        global PRECISION
        self._prev, PRECISION = PRECISION, self._name
        return self

    def __exit__(self, *exc):
        global PRECISION
        PRECISION = self._prev
        return False


def same_pad(in_size, k, stride):
    """TF 'SAME': out = ceil(in/s); pad_before = max((out-1)*s + k - in, 0) // 2."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return out, total // 2


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s: expected a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s: tensor is on %s -- the SingleHDR hot path runs only on a HIP device "
                           "(no CPU fallback)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s: expected %s, got %s" % (name, str(dtype).replace("torch.", ""), t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s: tensor must be contiguous NHWC" % name)
    return t


HALF = torch.float16


def _is_h(t):
    return isinstance(t, torch.Tensor) and t.dtype == torch.float16


def _chkh(t, name):
    """fp16 feature map of the native-fp16 path: contiguous NHWC, C % 8 == 0"""
    t = _chk(t, name, torch.float16)
    if t.shape[-1] % 8:
        raise ValueError("%s: fp16 feature maps need C %% 8 == 0, got %d" % (name, t.shape[-1]))
    return t


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _d(t):
    """detach parameters so that raw kernels can consume them"""
    return None if t is None else t.detach()


# ---------------------------------------------------------------------------
# range slots (include/shdr.h: shdr_conv2d_fwd_prepared_ranged_f32)
# ---------------------------------------------------------------------------
# The split-operand conv plans scale their input by a power of two taken from a RANGE SLOT: a device float holding an upper bound of
# max |x| of the tensor.  The kernels that produce a tensor write its slot from their epilogue (`y_range`), bound-preserving ops
# (pooling, bilinear resize, clip, channel reversal) hand their input's slot on, host-known bounds become constant slots, and a tensor
# that arrives without one is measured by the library (one pass over it).  Slots travel as the attribute `_shdr_range` of the tensor.
_RANGE_SCOPES = []
_CONST_SLOTS = {}


class range_scope:
    """Slots taken inside the scope come out of ONE zero-initialised slab (one memset per step instead of one per tensor).  The step
    closures of `pipeline` open a scope per forward and per stream; outside a scope every slot is its own zeroed one-element tensor."""

    SLOTS = 256

    def __enter__(self):
        self._slabs = {}                   # (device, stream) -> [slab, slots taken]
        _RANGE_SCOPES.append(self)
        return self

    def __exit__(self, *exc):
        _RANGE_SCOPES.remove(self)
        return False

    def take(self, device):
        # one slab per stream: its zero-fill is ordered in front of the kernels that write its slots by the stream itself (the backward
        # pass of a multi-stream step takes slots on the stream of each node's forward op)
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        ent = self._slabs.get(key)
        if ent is None or ent[1] == self.SLOTS:
            ent = self._slabs[key] = [torch.zeros(self.SLOTS, device=device, dtype=torch.float32), 0]
        ent[1] += 1
        return ent[0][ent[1] - 1:ent[1]]


PROJECTED_LAUNCHES = [0]                  # convolutions that wrote a projected output from their epilogue (diagnostic / tests)
RANGE_MISSES = collections.Counter()      # split-operand layers whose input arrived without a range (diagnostic: each costs one pass over x)


def _new_slot(device):
    if _RANGE_SCOPES:
        return _RANGE_SCOPES[-1].take(device)
    return torch.zeros(1, device=device, dtype=torch.float32)


def _range_of(t):
    """the range slot of a tensor (1-element fp32 device tensor) or None; a host-known bound (`_shdr_bound`) becomes a constant slot"""
    if t is None:
        return None
    r = getattr(t, "_shdr_range", None)
    if r is not None:
        # a slot describes the tensor as it was written: an in-place update since then (autograd accumulating a second gradient
        # into the same buffer) voids it -- torch bumps the version counter on every in-place op
        return r if getattr(t, "_shdr_range_ver", t._version) == t._version else None
    b = getattr(t, "_shdr_bound", None)
    if b is None:
        return None
    key = (t.device, float(b))
    slot = _CONST_SLOTS.get(key)
    if slot is None:
        slot = _CONST_SLOTS[key] = torch.full((1,), float(b), device=t.device, dtype=torch.float32)
    return slot


def _set_range(t, slot):
    t._shdr_range = slot
    t._shdr_range_ver = t._version
    return t


def _carry_range(y, x, factor=None):
    """y is bounded by x's bound (pooling, convex resampling, channel permutation, clipping of an already bounded tensor, the
    derivative of an activation applied to a gradient)"""
    r = _range_of(x) if getattr(x, "_shdr_range", None) is not None else None
    if r is not None:
        _set_range(y, r)
    b = getattr(x, "_shdr_bound", None)
    if b is not None:
        y._shdr_bound = b
    return y


def set_bound(t, bound):
    """declare a host-known upper bound of max |t| (e.g. an image in [0, 1])"""
    t._shdr_bound = float(bound)
    return t


def absmax_slot(x):
    """measure max |x| into a fresh range slot and attach it to x"""
    lib = _lib.load()
    xd = _chk(_d(x), "x")
    slot = _new_slot(xd.device)
    _lib.check(lib.shdr_absmax_f32(_ptr(xd), xd.numel(), _ptr(slot), _stream()), "shdr_absmax_f32")
    _set_range(x, slot)
    return slot


def conv2d(x, w, bias=None, stride=1, x2=None, x2_scale=1.0, act1=ACT_NONE, scale=None,
           shift=None, residual=None, act2=ACT_NONE, algo=ALGO_AUTO, out=None, cout_valid=None,
           pad=None, out_hw=None, w_batch_stride=0):
    """y = act2(affine(act1(conv(concat[x, x2_scale*x2], w) + bias)) + residual), SAME padding.

    `cout_valid` < w.shape[3] says the filter is zero-padded along Cout (to a multiple of 16 so
    that a narrow head runs on the MFMA tile); only the first `cout_valid` channels are stored.
    `pad=(top, left)` / `out_hw=(Ho, Wo)` override the SAME rule (used by the strided dgrad)."""
    if _is_h(x):
        if scale is not None or shift is not None or residual is not None or act2 != ACT_NONE or pad is not None or out is not None \
                or w_batch_stride or algo != ALGO_AUTO:
            raise NotImplementedError("conv2d: the native-fp16 path takes bias + act1 only (training-mode layers)")
        if _needs_grad(x, x2, w, bias):
            return AUTOGRAD.conv2d_h(x, w, bias, stride, x2, x2_scale, act1, cout_valid)
        return conv2d_h(x, pack_filter_h(w, x.shape[3], 0 if x2 is None else x2.shape[3], x2_scale), bias, tuple(w.shape[:2]),
                        w.shape[3], stride=stride, x2=x2, act1=act1, cout_valid=cout_valid)
    if algo == ALGO_AUTO:
        algo = _AUTO_ALGO[PRECISION]
    fused = scale is not None or shift is not None or residual is not None or act2 != ACT_NONE
    if _needs_grad(x, x2, w, bias, scale, shift, residual):
        if pad is not None or out_hw is not None or out is not None or w_batch_stride:
            raise NotImplementedError("conv2d: pad / out_hw / out / w_batch_stride are raw-kernel options (input-gradient and "
                                      "Winograd plumbing) and cannot be recorded on a gradient tape")
        y = AUTOGRAD.conv2d(x, w, bias, stride=stride, x2=x2, x2_scale=x2_scale, act1=act1, algo=algo, cout_valid=cout_valid)
        if not fused:
            return y
        # A fused inference epilogue (folded BatchNorm, residual join, second activation) under a tape -- e.g.
        # `lin(x, training=False)` while fine-tuning with frozen BatchNorm statistics: recorded as conv + affine/join/activation,
        # two tape entries, instead of silently returning a tensor that is cut off from the graph.
        return AUTOGRAD.affine_act(y, scale, shift, residual, act2)
    return _conv2d_raw(x, w, bias, stride, x2, x2_scale, act1, scale, shift, residual, act2, algo, out, cout_valid, pad, out_hw,
                       w_batch_stride, None)


def conv2d_up2(x, w, bias=None, act1=ACT_NONE, scale=None, shift=None, act2=ACT_NONE, proj=None):
    """Conv2D(SAME, stride 1)(tf.image.resize(x, 2x, BILINEAR)) with the conv2d epilogue -- the `up` blocks of
    hallucination_net.py:86-88 and dequantization_net.py:25-27.  Without a gradient tape the library fuses the resize into the
    convolution where the plan allows it (desc.prologue, csrc/conv_plan.hip); under a tape, and in the reduced-precision operand
    modes, it is the two recorded ops.
    proj [3, Cout]: the PROJECTED output sum_c proj[j, c] y[..., c] instead of y (include/shdr.h:
    shdr_conv2d_fwd_prepared_projected_f32), or None when the planned kernel of the layer cannot form it."""
    algo = _AUTO_ALGO[PRECISION]
    if _is_h(x) or algo != ALGO_AUTO or not WINOGRAD or _needs_grad(x, w, bias, scale, shift):
        if proj is not None:
            return None
        return conv2d(resize2x(x), w, bias, act1=act1, scale=scale, shift=shift, act2=act2)
    return _conv2d_raw(x, w, bias, 1, None, 1.0, act1, scale, shift, None, act2, algo, None, None, None, None, 0, None,
                       prologue=PROLOGUE_BILINEAR2X, proj=proj)


def _conv2d_raw(x, w, bias, stride, x2, x2_scale, act1, scale, shift, residual, act2, algo, out, cout_valid, pad, out_hw,
                w_batch_stride, pool, prologue=0, proj=None):
    """One convolution through the C ABI.  The kernel family (one-kernel Winograd, three-kernel Winograd, register-A / LDS-DMA
    implicit GEMM, direct) is chosen BELOW the ABI (shdr_conv2d_plan_f32, csrc/conv_plan.hip); this wrapper only checks shapes,
    caches the prepared filter of persistent variables per version and provides memory.  pool: None, True (also return
    MaxPool2D(2)(y)), "only" (the pooled tensor alone) or "avg" (also return AveragePooling2D(2)(y)).
    proj [3, Cout]: return the PROJECTED output [N, Ho, Wo, 3] (sum_c proj[j, c] y[..., c]) instead of y -- (y_proj, pooled) with a
    pool -- or None when the layer's planned kernel cannot form it (conv2d_projected falls back to two convolutions)."""
    lib = _lib.load()
    x_in, x2_in = x, x2                # the tensors as handed over: they carry the range slots (a detached copy does not)
    x = _chk(_d(x), "x")
    w_var = w                          # the variable itself: its version / leaf status key the prepared-filter cache
    w = _chk(_d(w), "w")
    n, h, wd, c1 = x.shape
    if prologue == PROLOGUE_BILINEAR2X:
        h, wd = 2 * h, 2 * wd              # the descriptor describes the convolution: its input is the up-sampled image
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
    if pad is not None:
        pt, pl = pad
    if out_hw is not None:
        ho, wo = out_hw
    d = _lib.ConvDesc()
    d.N, d.H, d.W, d.C1, d.C2 = n, h, wd, c1, c2
    d.Cout, d.KH, d.KW, d.stride = cout_gemm, kh, kw, stride
    d.cout_valid = cout
    d.pad_t, d.pad_l, d.Ho, d.Wo = pt, pl, ho, wo
    d.x2_scale = float(x2_scale)
    d.act1, d.act2 = act1, act2
    d.algo = _auto(algo)
    d.w_batch_stride = int(w_batch_stride)
    d.prologue = int(prologue)
    d.pool = POOL_AVG if pool == "avg" else POOL_MAX
    res_cs = 0
    if residual is not None:
        residual = _chk(_d(residual), "residual")
        if tuple(residual.shape[:3]) != (n, ho, wo) or residual.shape[3] < cout:
            raise ValueError("conv2d: residual shape %s incompatible with output [%d,%d,%d,%d]"
                             % (tuple(residual.shape), n, ho, wo, cout))
        res_cs = residual.shape[3]
    d.res_cstride = res_cs
    if proj is not None:
        if algo != ALGO_AUTO or not WINOGRAD or residual is not None or not int(lib.shdr_conv2d_projected_ok_f32(ctypes.byref(d))):
            return None
        proj = _chk(_d(proj), "proj")
        if tuple(proj.shape) != (3, cout):
            raise ValueError("conv2d: proj must be [3, %d]" % cout)
    elif out is None:
        out = None if pool == "only" else torch.empty((n, ho, wo, cout), device=x.device, dtype=torch.float32)
    else:
        _chk(out, "out")
        if tuple(out.shape) != (n, ho, wo, cout):
            raise ValueError("conv2d: bad out shape")
    d.y_cstride = cout
    for t, nm, ln in ((bias, "bias", cout), (scale, "scale", cout), (shift, "shift", cout)):
        if t is not None and (_chk(_d(t), nm).numel() != ln):
            raise ValueError("conv2d: %s must have %d elements" % (nm, ln))
    if algo != ALGO_AUTO or not WINOGRAD:
        # explicit kernel family / reduced-precision operand modes / Winograd switched off: the plain entry point
        if pool:
            raise ValueError("conv2d: the pooled output needs algo AUTO")
        rc = lib.shdr_conv2d_fwd_f32(ctypes.byref(d), _ptr(x), _ptr(x2), _ptr(w), _ptr(_d(bias)),
                                     _ptr(_d(scale)), _ptr(_d(shift)), _ptr(residual), _ptr(out), _stream())
        _lib.check(rc, "shdr_conv2d_fwd_f32")
        return out
    has_res = int(residual is not None)
    prepared = _prepared_filter(lib, w_var, d, has_res)
    plan = int(lib.shdr_conv2d_plan_f32(ctypes.byref(d), has_res))
    split = plan in (4, 5)             # SHDR_PLAN_X3 / X3N: the input is scaled by its range, the epilogue tracks the output's
    xr1 = _range_of(x_in) if split else None
    xr2 = _range_of(x2_in) if (split and x2 is not None) else None
    if split and (xr1 is None or (x2 is not None and xr2 is None)):
        RANGE_MISSES[(tuple(x.shape), None if x2 is None else tuple(x2.shape), tuple(w.shape))] += 1     # measured below the ABI
    ws = None
    nws = int(lib.shdr_conv2d_workspace_bytes_f32(ctypes.byref(d), has_res))
    if nws > 256 or (nws > 0 and (not split or xr1 is None or (x2 is not None and xr2 is None))):
        ws = torch.empty(nws, device=x.device, dtype=torch.uint8)      # (a split plan with known ranges needs no scratch slots)
    yp = None
    if pool:
        if ho % 2 or wo % 2:
            raise ValueError("conv2d: pool needs even output height / width")
        yp = torch.empty((n, ho // 2, wo // 2, cout), device=x.device, dtype=torch.float32)
        if out is None and plan not in (2, 4):      # the fused Winograd and split kernels can skip y
            out = torch.empty((n, ho, wo, cout), device=x.device, dtype=torch.float32)
    # the output's range slot: written by the epilogue of the split-operand and of the exact-fp32 MFMA / direct kernels (no extra pass)
    yr = _new_slot(x.device) if (split or (plan in (0, 1) and prologue == PROLOGUE_NONE and out is not None)) else None
    if proj is not None:
        yj = torch.empty((n, ho, wo, 3), device=x.device, dtype=torch.float32)
        rc = lib.shdr_conv2d_fwd_prepared_projected_f32(ctypes.byref(d), _ptr(x), _ptr(x2), _ptr(prepared), _ptr(_d(bias)), _ptr(_d(scale)),
                                                        _ptr(_d(shift)), _ptr(proj), _ptr(yj), None, _ptr(yp), _ptr(ws), _ptr(xr1),
                                                        _ptr(xr2), _ptr(yr), _stream())
        _lib.check(rc, "shdr_conv2d_fwd_prepared_projected_f32")
        PROJECTED_LAUNCHES[0] += 1
        if yp is not None:
            _set_range(yp, yr)             # (the pooled copy of y is bounded by max |y|)
        return (yj, yp) if pool else yj
    rc = lib.shdr_conv2d_fwd_prepared_ranged_f32(ctypes.byref(d), _ptr(x), _ptr(x2), _ptr(prepared), _ptr(_d(bias)), _ptr(_d(scale)),
                                                 _ptr(_d(shift)), _ptr(residual), _ptr(out), _ptr(yp), _ptr(ws), _ptr(xr1), _ptr(xr2),
                                                 _ptr(yr), _stream())
    _lib.check(rc, "shdr_conv2d_fwd_prepared_ranged_f32")
    if yr is not None:
        for t in (out, yp):
            if t is not None:
                _set_range(t, yr)          # (the pooled copy is bounded by the same maximum)
    if pool == "only":
        return yp
    return (out, yp) if pool else out


def _is_persistent(w):
    """requires_grad leaves (the layers' kernels) and frozen constants (_shdr_const: VGG16, padded / composed filters)"""
    return (w.requires_grad and w.is_leaf) or getattr(w, "_shdr_const", False)


def _filter_cache_get(w, attr, key):
    """Derived forms of a persistent filter (prepared / packed / fp16-packed) live ON the filter tensor in a dict
    {key: (tensor, producing stream, ready event)} tagged with the variable's version.  Entries are dropped only when the version
    changes: the same weight takes different plans at different input shapes (x3 needs >= 192 blocks, x3n >= 256 tiles), and a
    captured HIP graph holds the raw pointer of the form it was captured with -- evicting one shape's form for another's would
    leave that graph reading freed memory.  A consumer on another stream than the producer waits for the producer's event."""
    store = getattr(w, attr, None)
    if store is None or store[0] != w._version:
        return None
    ent = store[1].get(key)
    if ent is None:
        return None
    tensor, stream_id, event = ent
    # (no event calls while a stream capture is open: hipEventQuery invalidates the capture; GraphedInference warms the caches up first)
    if event is not None and torch.cuda.current_stream().cuda_stream != stream_id and not torch.cuda.is_current_stream_capturing() \
            and not event.query():
        torch.cuda.current_stream().wait_event(event)
    return tensor


def _filter_cache_put(w, attr, key, tensor):
    store = getattr(w, attr, None)
    if store is None or store[0] != w._version:
        store = (w._version, {})
        setattr(w, attr, store)
    event = None
    if not torch.cuda.is_current_stream_capturing():
        event = torch.cuda.Event()
        event.record()
    store[1][key] = (tensor, torch.cuda.current_stream().cuda_stream, event)
    return tensor


def clear_filter_caches(w):
    """forget every derived form kept on a filter tensor (the tensor moved to another device, or was rewritten behind torch's back)"""
    for attr in ("_shdr_packed", "_shdr_packed_h", "_shdr_wino"):
        if hasattr(w, attr):
            delattr(w, attr)


def _prepared_filter(lib, w, d, has_res):
    """shdr_conv2d_prepare_filter_f32(w) for the plan of this layer -- `w` itself when the library says the prepared form is the
    plain filter; kept ON the filter tensor per version and per (plan, sources, skip scale) for persistent variables
    (requires_grad leaves: the layers' kernels, frozen VGG16 constants), so an inference step prepares nothing; temporaries
    (transposed dgrad filters) per call.  Every kernel that rewrites a variable through a raw pointer bumps its version
    (_mutated, KerasAdam)."""
    if int(lib.shdr_conv2d_filter_is_plain_f32(ctypes.byref(d), has_res)):
        return _d(w)
    persistent = _is_persistent(w)
    key = None
    if persistent:
        key = (int(lib.shdr_conv2d_plan_f32(ctypes.byref(d), has_res)), d.C1, d.C2, float(d.x2_scale))
        cached = _filter_cache_get(w, "_shdr_packed", key)
        if cached is not None:
            return cached
    n = int(lib.shdr_conv2d_prepared_filter_elems_f32(ctypes.byref(d), has_res))
    prepared = torch.empty(n, device=w.device, dtype=torch.float32)
    _lib.check(lib.shdr_conv2d_prepare_filter_f32(ctypes.byref(d), has_res, _ptr(_d(w)), _ptr(prepared), _stream()),
               "shdr_conv2d_prepare_filter_f32")
    if persistent:
        _filter_cache_put(w, "_shdr_packed", key, prepared)
    return prepared


_PLAN_NAMES = {0: "direct", 1: "mfma", 2: "fused", 3: "planes", 4: "x3", 5: "x3n"}


def conv2d_plan(x_shape, w_shape, c2=0, stride=1, x2_scale=1.0, has_residual=False, cout_valid=None):
    """the kernel family shdr_conv2d_fwd_prepared_f32 runs for a SAME-padded layer: "fused" (one-kernel Winograd), "planes"
    (three-kernel Winograd), "mfma" (register-A / LDS-DMA implicit GEMM) or "direct" -- asked from the library, not decided here"""
    lib = _lib.load()
    n, h, wd, c1 = x_shape
    kh, kw, cin, cout = w_shape
    d = _lib.ConvDesc()
    d.N, d.H, d.W, d.C1, d.C2 = n, h, wd, c1, c2
    d.Cout, d.KH, d.KW, d.stride = cout, kh, kw, stride
    d.cout_valid = cout_valid or cout
    d.Ho, d.pad_t = same_pad(h, kh, stride)
    d.Wo, d.pad_l = same_pad(wd, kw, stride)
    d.x2_scale = float(x2_scale)
    d.algo = _auto(ALGO_AUTO) if WINOGRAD else ALGO_MFMA
    return _PLAN_NAMES[int(lib.shdr_conv2d_plan_f32(ctypes.byref(d), int(has_residual)))]


def winograd_path(cin, cout):
    """"fused" / "planes" / None for a single-source 3x3 stride-1 layer with these channel counts (library policy)"""
    p = conv2d_plan((1, 64, 64, cin), (3, 3, cin, cout))
    return p if p in ("fused", "planes") else None


def conv2d_dgrad(dz, w, x_shape, c1, c2, which, stride=1, x2_scale=1.0):
    """dx [x_shape] of the SAME-padded forward conv(concat[x1 (c1 ch), x2_scale * x2 (c2 ch)], w) w.r.t. source `which` from
    dz = dL/d(conv output) [N,Ho,Wo,cout_valid]: ONE library call (shdr_conv2d_dgrad_f32: filter flip / slice / zero-padding,
    Winograd where the transposed layer qualifies, 1x1 / 2 on the coarse grid, polyphase form for general stride 2)."""
    lib = _lib.load()
    dz_in = dz
    dz, w = _chk(_d(dz), "dz"), _chk(_d(w), "w")
    n, h, wd, cx = x_shape
    kh, kw, cin, cout = w.shape
    if cin != c1 + c2 or cx != (c2 if which else c1):
        raise ValueError("conv2d_dgrad: filter %s / sources %d+%d / x %s do not match" % (tuple(w.shape), c1, c2, tuple(x_shape)))
    d = _lib.ConvDesc()
    d.N, d.H, d.W, d.C1, d.C2 = n, h, wd, c1, c2
    d.Cout, d.KH, d.KW, d.stride = cout, kh, kw, stride
    d.cout_valid = dz.shape[3]
    d.Ho, d.pad_t = same_pad(h, kh, stride)
    d.Wo, d.pad_l = same_pad(wd, kw, stride)
    d.x2_scale = float(x2_scale)
    d.algo = _auto(_AUTO_ALGO[PRECISION]) if WINOGRAD or PRECISION != "fp32" else ALGO_MFMA
    if tuple(dz.shape[:3]) != (n, d.Ho, d.Wo) or dz.shape[3] > cout:
        raise ValueError("conv2d_dgrad: dz shape %s does not match the forward output" % (tuple(dz.shape),))
    nws = int(lib.shdr_conv2d_dgrad_workspace_bytes_f32(ctypes.byref(d), int(which)))
    if nws < 0:
        raise ValueError("conv2d_dgrad: bad descriptor")
    ws = torch.empty(max(nws, 16), device=dz.device, dtype=torch.uint8)
    dx = torch.empty(tuple(x_shape), device=dz.device, dtype=torch.float32)
    # range slots (a chain conv <- activation <- conv of a backward pass never measures: |act' dz| <= |dz|, _carry_range)
    zr = _range_of(dz_in)
    xr = _new_slot(dz.device) if int(lib.shdr_conv2d_dgrad_tracks_range_f32(ctypes.byref(d), int(which))) else None
    if zr is None and xr is not None:
        RANGE_MISSES[("dgrad", tuple(dz.shape), tuple(w.shape))] += 1
    _lib.check(lib.shdr_conv2d_dgrad_ranged_f32(ctypes.byref(d), int(which), _ptr(dz), _ptr(w), _ptr(dx), _ptr(ws), _ptr(zr), _ptr(xr), _stream()),
               "shdr_conv2d_dgrad_ranged_f32")
    if xr is not None:
        _set_range(dx, xr)
    return dx


def soft_hist_bwd(img, dy, max_bin):
    lib = _lib.load()
    img, dy = _chk(_d(img), "img"), _chk(_d(dy), "dy")
    c = img.shape[-1]
    dx = torch.empty_like(img)
    _lib.check(lib.shdr_soft_hist_bwd_f32(_ptr(img), _ptr(dy), _ptr(dx), img.numel() // c, c, int(max_bin), _stream()), "shdr_soft_hist_bwd_f32")
    return dx


def _nhwc_op(fn_name, x, out_shape):
    lib = _lib.load()
    x_in = x
    if _is_h(x):                       # the _f16 twin (csrc/elem_f16.hip)
        x = _chkh(_d(x), "x")
        fn_name = fn_name[:-4] + "_f16"
    else:
        x = _chk(_d(x), "x")
    n, h, w, c = x.shape
    y = torch.empty(out_shape, device=x.device, dtype=x.dtype)
    rc = getattr(lib, fn_name)(_ptr(x), _ptr(y), n, h, w, c, _stream())
    _lib.check(rc, fn_name)
    return _carry_range(y, x_in)        # pooling / bilinear resize: bounded by the input's maximum


def avgpool2(x):
    if _needs_grad(x):
        return AUTOGRAD.avgpool2(x)
    n, h, w, c = x.shape
    return _nhwc_op("shdr_avgpool2_fwd_f32", x, (n, h // 2, w // 2, c))


def maxpool2(x):
    if _needs_grad(x):
        return AUTOGRAD.maxpool2(x)
    n, h, w, c = x.shape
    return _nhwc_op("shdr_maxpool2_fwd_f32", x, (n, h // 2, w // 2, c))


def maxpool3s2(x):
    if _needs_grad(x):
        return AUTOGRAD.maxpool3s2(x)
    n, h, w, c = x.shape
    return _nhwc_op("shdr_maxpool3s2_fwd_f32", x, (n, same_pad(h, 3, 2)[0], same_pad(w, 3, 2)[0], c))


def resize2x(x):
    if _needs_grad(x):
        return AUTOGRAD.resize2x(x)
    n, h, w, c = x.shape
    return _nhwc_op("shdr_resize2x_fwd_f32", x, (n, 2 * h, 2 * w, c))


def global_avg_pool(x):
    if _needs_grad(x):
        return AUTOGRAD.global_avg_pool(x)
    lib = _lib.load()
    n, h, w, c = x.shape
    y = torch.empty((n, c), device=x.device, dtype=torch.float32)
    if _is_h(x):
        _lib.check(lib.shdr_gap_fwd_f16(_ptr(_chkh(_d(x), "x")), _ptr(y), n, h * w, c, _stream()), "shdr_gap_fwd_f16")
        return y
    x = _chk(_d(x), "x")
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


def lin_frontend(img, channels=96, dtype=None):
    """dtype: torch.float16 inside precision("fp16") (native-fp16 feature maps), float32 otherwise"""
    dtype = dtype or (HALF if native_fp16() else torch.float32)
    if _needs_grad(img):
        return AUTOGRAD.lin_frontend(img, channels, dtype)
    lib = _lib.load()
    img_in = img
    img = _chk(_d(img), "img")
    n, h, w, c = img.shape
    if c != 3:
        raise ValueError("lin_frontend: expected 3 channels, got %d" % c)
    y = torch.empty((n, h, w, channels), device=img.device, dtype=dtype)
    fn = "shdr_lin_frontend_fwd_f16" if dtype == HALF else "shdr_lin_frontend_fwd_f32"
    _lib.check(getattr(lib, fn)(_ptr(img), _ptr(y), n, h, w, channels, _stream()), fn)
    b = getattr(img_in, "_shdr_bound", None)
    if b is not None:                  # image: b, sobel: sum |coefficient| = 8 times b, soft-histogram channels: [0, 1]
        set_bound(y, max(1.0, 8.0 * b))
    return y


def invcrf_decode(feat, wfc, bfc, table):
    if _needs_grad(feat, wfc, bfc):
        return AUTOGRAD.invcrf_decode(feat, wfc, bfc, table)
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
    if _needs_grad(rf):
        return AUTOGRAD.increase(rf)
    lib = _lib.load()
    rf = _chk(_d(rf), "rf")
    b, k = rf.shape
    out = torch.empty_like(rf)
    _lib.check(lib.shdr_increase_fwd_f32(_ptr(rf), _ptr(out), b, k, _stream()), "shdr_increase_fwd_f32")
    return out


def apply_rf(x, rf):
    if _needs_grad(x, rf):
        return AUTOGRAD.apply_rf(x, rf)
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
    if _needs_grad(x):
        return AUTOGRAD.clip(x, lo, hi)
    lib = _lib.load()
    x = _chk(_d(x), "x")
    y = torch.empty_like(x)
    _lib.check(lib.shdr_clip_fwd_f32(_ptr(x), _ptr(y), x.numel(), float(lo), float(hi), _stream()),
               "shdr_clip_fwd_f32")
    return set_bound(y, max(abs(float(lo)), abs(float(hi))))


def logc(x):
    if _needs_grad(x):
        return AUTOGRAD.logc(x)
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


def vgg_preprocess(x, out_channels=3, dtype=torch.float32):
    """x*255, RGB->BGR, minus VGG mean; out_channels=4 appends a zero channel (MFMA-friendly); dtype float16: the zero-padded
    fp16 feature map of the native-fp16 path (out_channels % 8 == 0)."""
    if _needs_grad(x):
        return AUTOGRAD.vgg_preprocess(x, out_channels, dtype)
    lib = _lib.load()
    x_in = x
    x, npix = _pix3(x, "x")
    if dtype == HALF:
        y = torch.empty(tuple(x.shape[:-1]) + (out_channels,), device=x.device, dtype=HALF)
        _lib.check(lib.shdr_pack3_f16(_ptr(x), None, None, None, 1, _ptr(y), out_channels, npix, 1, _stream()), "shdr_pack3_f16")
        return y
    y = torch.empty(tuple(x.shape[:-1]) + (out_channels,), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_vgg_preprocess_fwd_f32(_ptr(x), _ptr(y), npix, out_channels, _stream()),
               "shdr_vgg_preprocess_fwd_f32")
    b = getattr(x_in, "_shdr_bound", None)
    if b is not None:
        set_bound(y, 255.0 * b + 124.0)     # x * 255 - mean, |mean| < 124
    return y


def reverse3(x):
    if _needs_grad(x):
        return AUTOGRAD.reverse3(x)
    lib = _lib.load()
    x, npix = _pix3(x, "x")
    y = torch.empty_like(x)
    _lib.check(lib.shdr_reverse3_fwd_f32(_ptr(x), _ptr(y), npix, _stream()), "shdr_reverse3_fwd_f32")
    return y


def alpha_blend(b_pred, hal_bgr, thr=0.12, return_alpha=False):
    if _needs_grad(b_pred, hal_bgr):
        if return_alpha:
            raise NotImplementedError("alpha_blend: return_alpha is not available on the taped path")
        return AUTOGRAD.alpha_blend(b_pred, hal_bgr, thr)
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


def pack3(srcs, out_channels=None, dtype=torch.float32):
    if _needs_grad(*srcs):
        return AUTOGRAD.pack3(list(srcs), out_channels, dtype)
    lib = _lib.load()
    srcs_in = list(srcs)
    srcs = [_pix3(s, "src%d" % i)[0] for i, s in enumerate(srcs)]
    n = len(srcs)
    if not 1 <= n <= 4:
        raise ValueError("pack3: 1..4 sources")
    if any(s.shape != srcs[0].shape for s in srcs):
        raise ValueError("pack3: shape mismatch")
    oc = out_channels or 3 * n
    y = torch.empty(tuple(srcs[0].shape[:-1]) + (oc,), device=srcs[0].device, dtype=dtype)
    p = [_ptr(s) for s in srcs] + [None] * (4 - n)
    if dtype == HALF:
        _lib.check(lib.shdr_pack3_f16(p[0], p[1], p[2], p[3], n, _ptr(y), oc, srcs[0].numel() // 3, 0, _stream()), "shdr_pack3_f16")
        return y
    _lib.check(lib.shdr_pack3_fwd_f32(p[0], p[1], p[2], p[3], n, _ptr(y), oc, srcs[0].numel() // 3,
                                      _stream()), "shdr_pack3_fwd_f32")
    if n == 1:                               # a channel-padded copy: the same range
        return _carry_range(y, srcs_in[0])
    bounds = [getattr(t, "_shdr_bound", None) for t in srcs_in]
    if all(b is not None for b in bounds):   # (device-side slots of several sources are not merged: measured by the consumer)
        set_bound(y, max(bounds))
    return y


# ---------------------------------------------------------------------------
# raw wrappers of the backward / training kernels (no autograd; see _autograd.py)
# ---------------------------------------------------------------------------
def _conv_desc(x_shape, w_shape, stride, c2, x2_scale, cout_valid):
    n, h, wd, c1 = x_shape
    kh, kw, cin, cout_gemm = w_shape
    ho, pt = same_pad(h, kh, stride)
    wo, pl = same_pad(wd, kw, stride)
    d = _lib.ConvDesc()
    d.N, d.H, d.W, d.C1, d.C2 = n, h, wd, c1, c2
    d.Cout, d.KH, d.KW, d.stride = cout_gemm, kh, kw, stride
    d.cout_valid = cout_valid or cout_gemm
    d.pad_t, d.pad_l, d.Ho, d.Wo = pt, pl, ho, wo
    d.x2_scale = float(x2_scale)
    return d


def pad_channels(x, channels):
    """zero-pad the channel axis to `channels` (no-op if already that wide)"""
    lib = _lib.load()
    x = _chk(_d(x), "x")
    c = x.shape[-1]
    if c == channels:
        return x
    y = torch.empty(tuple(x.shape[:-1]) + (channels,), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_pad_channels_f32(_ptr(x), _ptr(y), x.numel() // c, c, channels, _stream()), "shdr_pad_channels_f32")
    return y


def _up16(c):
    return (c + 15) // 16 * 16


WGRAD_X3_MIN_CH = 128     # conv2d_wgrad: layers with at least this many channels per source and output channels take the split-operand kernel


def _split_planes(lib, t, slot):
    """(high plane, low plane, range slot) of an fp32 tensor for the split-operand weight gradient; the range is measured when the
    tensor carries no slot (output gradients never do)"""
    if slot is None:
        RANGE_MISSES[("wgrad operand", tuple(t.shape))] += 1
        slot = _new_slot(t.device)
        _lib.check(lib.shdr_absmax_f32(_ptr(t), t.numel(), _ptr(slot), _stream()), "shdr_absmax_f32")
    hi = torch.empty(t.shape, device=t.device, dtype=torch.float16)
    lo = torch.empty(t.shape, device=t.device, dtype=torch.float16)
    _lib.check(lib.shdr_x3_split_planes_f32(_ptr(t), t.numel(), _ptr(slot), _ptr(hi), _ptr(lo), _stream()), "shdr_x3_split_planes_f32")
    return hi, lo, slot


def conv2d_wgrad(x, x2, dz, w_shape, stride=1, x2_scale=1.0, out=None):
    """dW [kh,kw,C1+C2,Cout] of conv(concat[x, x2_scale*x2], W) given dz = dL/d(conv output).

    Channel counts that are not multiples of 16 (3/4-channel images, 3-channel heads) are zero-padded so
    that the layer runs on the MFMA weight-gradient tiles; the true rows / columns are sliced out.
    `out`: a gradient buffer of shape w_shape to ACCUMULATE into (the kernels add with atomics anyway): no zero fill and no
    separate add -- used for variables whose .grad is a view of the flat gradient buffer (pipeline.FlatParams)."""
    kh, kw, cin, cout = w_shape
    c1 = x.shape[3]
    c2 = 0 if x2 is None else x2.shape[3]
    if out is not None and (tuple(out.shape) != tuple(w_shape) or not out.is_contiguous() or out.dtype != torch.float32):
        raise ValueError("conv2d_wgrad: out must be a contiguous float32 tensor of shape %s" % (tuple(w_shape),))
    if c1 % 16 or c2 % 16 or cout % 16:
        if out is not None:
            out.add_(conv2d_wgrad(x, x2, dz, w_shape, stride, x2_scale))      # padded layers: sliced result, parameter-sized add
            return out
        c1p, c2p, coutp = _up16(c1), (_up16(c2) if c2 else 0), _up16(cout)
        dwp = conv2d_wgrad(pad_channels(x, c1p), None if x2 is None else pad_channels(x2, c2p), pad_channels(dz, coutp),
                           (kh, kw, c1p + c2p, coutp), stride, x2_scale)
        if c2:
            return torch.cat([dwp[:, :, :c1, :cout], dwp[:, :, c1p:c1p + c2, :cout]], dim=2).contiguous()
        return dwp[:, :, :c1, :cout].contiguous()
    lib = _lib.load()
    x_in, x2_in, dz_in = x, x2, dz
    x, dz = _chk(_d(x), "x"), _chk(_d(dz), "dz")
    if x2 is not None:
        _chk(_d(x2), "x2")
    # Deep k x k layers (>= WGRAD_X3_MIN_CH channels on both sides): the split-operand weight gradient on the fp16 matrix pipe
    # (csrc/wgrad_x3.hip) -- one split pass per tensor, then three fp16 MFMAs per operand pair.  Measured at batch 32 (tools/wgrad_x3_bench.py,
    # split passes included) against the Winograd-domain fp32 kernel: 512 -> 512 at 32^2 0.85 -> 0.60 ms, 512 -> 256 at 64^2 1.61 -> 1.06,
    # 256 -> 256 at 64^2 0.85 -> 0.62, 256 -> 128 at 128^2 1.59 -> 1.27, 128 -> 128 at 128^2 0.78 -> 0.72 (215-290 TFLOP/s in fp32 layer
    # FLOPs); 1x1 layers do too little arithmetic per element for the two split passes (0.33 -> 0.43 ms at 256 + 256 -> 256) and stay exact
    # (a 7x7 layer does 49 taps of arithmetic per split element: the stem of the Linearization-Net, 96 -> 64, 3.0 -> 1.9 ms at 32 x 256^2, split passes included)
    wide = min(c1, c2 or c1, cout) >= WGRAD_X3_MIN_CH or (kh * kw >= 25 and min(c1, c2 or c1, cout) >= 64)
    if (PRECISION == "fp32" and not EXACT_FP32 and WINOGRAD and kh * kw >= 9 and wide and cin == c1 + c2
            and cout == dz.shape[3] and x.numel() % 8 == 0 and dz.numel() % 8 == 0 and (x2 is None or x2.numel() % 8 == 0)):
        d = _conv_desc(x.shape, w_shape, stride, c2, x2_scale, None)
        if tuple(dz.shape[:3]) == (x.shape[0], d.Ho, d.Wo) and all(int(lib.shdr_conv2d_wgrad_x3_ok_f32(ctypes.byref(d), i)) for i in range(2 if c2 else 1)):
            dw = out if out is not None else torch.zeros(tuple(w_shape), device=x.device, dtype=torch.float32)
            zh, zl, zr = _split_planes(lib, dz, _range_of(dz_in))
            _set_range(dz_in, zr)           # the input gradient of the same layer (called next) takes the slot instead of measuring again
            for src, src_in, which in ((x, x_in, 0),) + (((_d(x2), x2_in, 1),) if x2 is not None else ()):
                xh, xl, xr = _split_planes(lib, src, _range_of(src_in))
                _lib.check(lib.shdr_conv2d_wgrad_x3_f32(ctypes.byref(d), _ptr(xh), _ptr(xl), which, _ptr(zh), _ptr(zl), _ptr(xr), _ptr(zr),
                                                        _ptr(dw), _stream()), "shdr_conv2d_wgrad_x3_f32")
            return dw
    # (in the reduced-precision modes too for <= 64 channels per source: the exact Winograd-domain kernel is faster there than the
    #  fp16-operand kernel -- 16 x 512^2 x 64 -> 64: 1.77 vs 2.98 ms -- and errs on the accurate side)
    if (WINOGRAD and (PRECISION == "fp32" or max(c1, c2) <= 64) and (kh, kw) == (3, 3) and stride == 1 and c1 % 32 == 0 and c2 % 32 == 0
            and cout % 64 == 0 and cin == c1 + c2 and cout == dz.shape[3] and tuple(dz.shape[:3]) == tuple(x.shape[:3])):
        # Winograd-domain weight gradient (csrc/wgrad_winograd.hip): 2.25x fewer MFMAs than the direct form
        n, h, w, _ = x.shape
        dw = out if out is not None else torch.zeros(tuple(w_shape), device=x.device, dtype=torch.float32)
        for src, cx, off, sc in ((x, c1, 0, 1.0),) + (((_d(x2), c2, c1, float(x2_scale)),) if x2 is not None else ()):
            du = torch.empty((16, cx, cout), device=x.device, dtype=torch.float32)       # scratch, zeroed by the library call
            _lib.check(lib.shdr_conv2d_wgrad_winograd_f32(_ptr(src), _ptr(dz), _ptr(du), _ptr(dw), n, h, w, cx, cout, cin, off, sc,
                                                          _stream()), "shdr_conv2d_wgrad_winograd_f32")
        return dw
    if cin != x.shape[3] + c2 or cout != dz.shape[3]:
        raise ValueError("conv2d_wgrad: filter %s does not match x %s (+%d) / dz %s"
                         % (tuple(w_shape), tuple(x.shape), c2, tuple(dz.shape)))
    d = _conv_desc(x.shape, w_shape, stride, c2, x2_scale, None)
    d.algo = _AUTO_ALGO[PRECISION]
    if tuple(dz.shape[:3]) != (x.shape[0], d.Ho, d.Wo):
        raise ValueError("conv2d_wgrad: dz spatial shape mismatch")
    dw = out if out is not None else torch.zeros(tuple(w_shape), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_conv2d_wgrad_f32(ctypes.byref(d), _ptr(x), 0, _ptr(dz), _ptr(dw), _stream()), "shdr_conv2d_wgrad_f32")
    if x2 is not None:
        _lib.check(lib.shdr_conv2d_wgrad_f32(ctypes.byref(d), _ptr(_d(x2)), 1, _ptr(dz), _ptr(dw), _stream()),
                   "shdr_conv2d_wgrad_f32")
    return dw


def filter_transform(w, c_begin, c_count, scale=1.0):
    """wt[kh,kw,co,ci] = scale * w[KH-1-kh, KW-1-kw, c_begin+ci, co]: the dgrad filter."""
    lib = _lib.load()
    w = _chk(_d(w), "w")
    kh, kw, cin, cout = w.shape
    wt = torch.empty((kh, kw, cout, c_count), device=w.device, dtype=torch.float32)
    _lib.check(lib.shdr_filter_transform_f32(_ptr(w), _ptr(wt), kh, kw, cin, cout, c_begin, c_count, float(scale), _stream()),
               "shdr_filter_transform_f32")
    return wt


def bias_grad(dz, out=None):
    lib = _lib.load()
    dz = _chk(_d(dz), "dz")
    c = dz.shape[-1]
    db = out if out is not None else torch.zeros(c, device=dz.device, dtype=torch.float32)
    _lib.check(lib.shdr_bias_grad_f32(_ptr(dz), _ptr(db), dz.numel() // c, c, _stream()), "shdr_bias_grad_f32")
    return db


def _bias_ws(c, device):
    """one row of partial channel sums per reduction block of act_bwd_bias (sized by the library)"""
    return torch.empty(int(_lib.load().shdr_workspace_bytes(_lib.OP_ACT_BWD_BIAS, None, int(c))) // 4, device=device, dtype=torch.float32)


def act_bwd_bias(dy, y, act, out=None):
    """(dz, db) of y = act(z + bias): dz = dy * act'(y), db = sum over pixels of dz -- one fused pass when the channel
    count allows (C / 4 a power of two), the act_bwd + bias_grad pair otherwise.  `out`: gradient buffer to accumulate db into."""
    if _is_h(dy):
        return act_bwd_bias_h(dy, y, act, True, out)
    dy_in = dy
    dy = _chk(_d(dy), "dy")
    c = dy.shape[-1]
    q = c // 4
    if c % 4 or q > 256 or q & (q - 1):
        dz = act_bwd(dy_in, y, act) if act != ACT_NONE else dy_in
        return dz, bias_grad(dz, out)
    lib = _lib.load()
    db = out if out is not None else torch.zeros(c, device=dy.device, dtype=torch.float32)
    if act == ACT_NONE:
        dz, yp, zp = dy_in, None, None  # (the caller's tensor object: it carries the range slot)
    else:
        y = _chk(_d(y), "y")
        dz = torch.empty_like(dy)
        yp, zp = _ptr(y), _ptr(dz)
    _lib.check(lib.shdr_act_bwd_bias_f32(_ptr(dy), yp, zp, _ptr(db), _ptr(_bias_ws(c, dy.device)), dy.numel() // c, c, act, _stream()),
               "shdr_act_bwd_bias_f32")
    if dz is not dy_in:
        _carry_range(dz, dy_in)         # |act'(y)| <= 1
    return dz, db


def act_bwd(dy, y, act):
    if _is_h(dy):
        return act_bwd_bias_h(dy, y, act, False)[0]
    lib = _lib.load()
    dy_in = dy
    dy, y = _chk(_d(dy), "dy"), _chk(_d(y), "y")
    dx = torch.empty_like(dy)
    _lib.check(lib.shdr_act_bwd_f32(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), act, _stream()), "shdr_act_bwd_f32")
    return _carry_range(dx, dy_in)     # |act'(y)| <= 1 for relu / leaky relu / tanh


def clip_bwd(dy, x, lo, hi):
    lib = _lib.load()
    dy_in = dy
    dy, x = _chk(_d(dy), "dy"), _chk(_d(x), "x")
    dx = torch.empty_like(dy)
    _lib.check(lib.shdr_clip_bwd_f32(_ptr(dy), _ptr(x), _ptr(dx), dy.numel(), float(lo), float(hi), _stream()), "shdr_clip_bwd_f32")
    return _carry_range(dx, dy_in)


def add(a, b, relu=False):
    """a + b (relu: max(a + b, 0), fp16 feature maps only)"""
    if _needs_grad(a, b):
        if relu:
            return AUTOGRAD.add_relu(a, b)
        return AUTOGRAD.add(a, b)
    lib = _lib.load()
    if _is_h(a):
        a, b = _chkh(_d(a), "a"), _chkh(_d(b), "b")
        if a.shape != b.shape:
            raise ValueError("add: shape mismatch")
        y = torch.empty_like(a)
        _lib.check(lib.shdr_add_f16(_ptr(a), _ptr(b), _ptr(y), a.numel(), int(bool(relu)), _stream()), "shdr_add_f16")
        return y
    if relu:
        return clip(add(a, b), 0.0, float("inf"))
    a, b = _chk(_d(a), "a"), _chk(_d(b), "b")
    if a.shape != b.shape:
        raise ValueError("add: shape mismatch")
    y = torch.empty_like(a)
    yr = _new_slot(a.device)            # max |a + b| out of the same pass (a gradient sum usually feeds a split-operand dgrad / wgrad)
    _lib.check(lib.shdr_add_ranged_f32(_ptr(a), _ptr(b), _ptr(y), a.numel(), _ptr(yr), _stream()), "shdr_add_ranged_f32")
    return _set_range(y, yr)


def mark(x, callback):
    """x itself; on a gradient tape `callback()` runs when the backward pass reaches this point (every backward op of the layers AFTER
    it in the forward has been enqueued): where the data-parallel steps launch a gradient bucket's all-reduce"""
    if callback is None or not _needs_grad(x):
        return x
    return _carry_range(AUTOGRAD.mark(x, callback), x)


def fork(x, n=2):
    """n aliases of x for n consumers (skip connections, residual shortcuts, multi-term losses): on a gradient tape the consumers'
    gradients are summed by the add kernel of this library; without a tape it is x itself, n times"""
    if not _needs_grad(x):
        return (x,) * n
    return tuple(_carry_range(t, x) for t in AUTOGRAD.fork(x, n))      # (the aliases are new tensor objects: hand the range slot on)


def affine_act(x, scale=None, shift=None, residual=None, act=ACT_NONE):
    """act(x * scale[c] + shift[c] + residual) on NHWC"""
    if _needs_grad(x, scale, shift, residual):
        return AUTOGRAD.affine_act(x, scale, shift, residual, act)
    lib = _lib.load()
    x = _chk(_d(x), "x")
    c = x.shape[-1]
    for t, nm in ((scale, "scale"), (shift, "shift")):
        if t is not None and _chk(_d(t), nm).numel() != c:
            raise ValueError("affine_act: %s must have %d elements" % (nm, c))
    if residual is not None and _chk(_d(residual), "residual").shape != x.shape:
        raise ValueError("affine_act: residual shape %s != %s" % (tuple(residual.shape), tuple(x.shape)))
    y = torch.empty_like(x)
    _lib.check(lib.shdr_affine_act_f32(_ptr(x), _ptr(_d(scale)), _ptr(_d(shift)), _ptr(_d(residual)), _ptr(y), x.numel() // c, c,
                                       act, _stream()), "shdr_affine_act_f32")
    return y


def _bwd_nhwc(fn_name, x_shape, *tensors):
    """input-gradient kernels taking (tensors..., dx, N, H, W, C) of the op's INPUT shape"""
    lib = _lib.load()
    if _is_h(tensors[0]):
        ts = [_chkh(_d(t), "t%d" % i) for i, t in enumerate(tensors)]
        fn_name = fn_name[:-4] + "_f16"
    else:
        ts = [_chk(_d(t), "t%d" % i) for i, t in enumerate(tensors)]
    n, h, w, c = x_shape
    dx = torch.empty(tuple(x_shape), device=ts[0].device, dtype=ts[0].dtype)
    _lib.check(getattr(lib, fn_name)(*[_ptr(t) for t in ts], _ptr(dx), n, h, w, c, _stream()), fn_name)
    return dx


def avgpool2_bwd(dy, x_shape):
    return _carry_range(_bwd_nhwc("shdr_avgpool2_bwd_f32", x_shape, dy), dy)          # dy / 4


def maxpool2_bwd(x, dy):
    return _carry_range(_bwd_nhwc("shdr_maxpool2_bwd_f32", x.shape, x, dy), dy)       # dy or 0 (disjoint windows)


def maxpool3s2_bwd(x, y, dy):
    """`y` is the forward's output maxpool3s2(x)"""
    return _bwd_nhwc("shdr_maxpool3s2_bwd_f32", x.shape, x, y, dy)


def resize2x_bwd(dy, x_shape):
    if _is_h(dy):
        return _bwd_nhwc("shdr_resize2x_bwd_f32", x_shape, dy)
    lib = _lib.load()
    dy = _chk(_d(dy), "dy")
    n, h, w, c = x_shape
    dx = torch.empty(tuple(x_shape), device=dy.device, dtype=torch.float32)
    xr = _new_slot(dy.device)           # max |dx| out of the same pass (the consumer is the input gradient of the conv in front)
    _lib.check(lib.shdr_resize2x_bwd_ranged_f32(_ptr(dy), _ptr(dx), n, h, w, c, _ptr(xr), _stream()), "shdr_resize2x_bwd_ranged_f32")
    return _set_range(dx, xr)


def upsample_zero2(dy, x_shape):
    return _carry_range(_bwd_nhwc("shdr_upsample_zero2_f32", x_shape, dy), dy)         # dy or 0


def gap_bwd(dy, x_shape, dtype=torch.float32):
    lib = _lib.load()
    dy = _chk(_d(dy), "dy")
    n, h, w, c = x_shape
    dx = torch.empty(tuple(x_shape), device=dy.device, dtype=dtype)
    fn = "shdr_gap_bwd_f16" if dtype == HALF else "shdr_gap_bwd_f32"
    _lib.check(getattr(lib, fn)(_ptr(dy), _ptr(dx), n, h * w, c, _stream()), fn)
    return dx


def _bn_ws(c, device):
    """workspace of the BatchNorm reductions (sums + one partial row per block), sized by the library"""
    nbytes = int(_lib.load().shdr_workspace_bytes(_lib.OP_BATCHNORM, None, int(c)))
    return torch.empty(nbytes // 8, device=device, dtype=torch.float64)


def bn_stats(x, moving_mean=None, moving_var=None, momentum=0.99):
    """batch mean / biased variance over (N,H,W); optionally updates the moving statistics in place."""
    lib = _lib.load()
    h = _is_h(x)
    x = _chkh(_d(x), "x") if h else _chk(_d(x), "x")
    c = x.shape[-1]
    mean = torch.empty(c, device=x.device, dtype=torch.float32)
    var = torch.empty(c, device=x.device, dtype=torch.float32)
    ws = _bn_ws(c, x.device)
    fn = "shdr_bn_stats_f16" if h else "shdr_bn_stats_f32"
    _lib.check(getattr(lib, fn)(_ptr(x), _ptr(ws), _ptr(mean), _ptr(var), _ptr(_d(moving_mean)), _ptr(_d(moving_var)),
                                x.numel() // c, c, float(momentum), _stream()), fn)
    _mutated(moving_mean, moving_var)
    return mean, var


def bn_train_apply(x, mean, var, gamma, beta, eps, relu):
    lib = _lib.load()
    h = _is_h(x)
    x = _chkh(_d(x), "x") if h else _chk(_d(x), "x")
    c = x.shape[-1]
    y = torch.empty_like(x)
    if not h:                          # fp32: the kernel also writes the range slot of y (its consumer is usually a split-operand conv)
        yr = _new_slot(x.device)
        _lib.check(lib.shdr_bn_train_apply_ranged_f32(_ptr(x), _ptr(mean), _ptr(var), _ptr(_d(gamma)), _ptr(_d(beta)), _ptr(y),
                                                      x.numel() // c, c, float(eps), int(bool(relu)), _ptr(yr), _stream()),
                   "shdr_bn_train_apply_ranged_f32")
        return _set_range(y, yr)
    _lib.check(lib.shdr_bn_train_apply_f16(_ptr(x), _ptr(mean), _ptr(var), _ptr(_d(gamma)), _ptr(_d(beta)), _ptr(y),
                                           x.numel() // c, c, float(eps), int(bool(relu)), _stream()), "shdr_bn_train_apply_f16")
    return y


def bn_bwd(dy, x, y_relu, mean, var, gamma, eps, dgamma_out=None, dbeta_out=None):
    """(dx, dgamma, dbeta); dgamma_out / dbeta_out: gradient buffers to accumulate into instead of fresh zero tensors"""
    lib = _lib.load()
    h = _is_h(x)
    dy, x = (_chkh(_d(dy), "dy"), _chkh(_d(x), "x")) if h else (_chk(_d(dy), "dy"), _chk(_d(x), "x"))
    c = x.shape[-1]
    dgamma = dgamma_out if dgamma_out is not None else torch.zeros(c, device=x.device, dtype=torch.float32)
    dbeta = dbeta_out if dbeta_out is not None else torch.zeros(c, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x)
    ws = _bn_ws(c, x.device)
    if h:
        _lib.check(lib.shdr_bn_bwd_f16(_ptr(dy), _ptr(x), _ptr(_d(y_relu)), _ptr(mean), _ptr(var), _ptr(_d(gamma)), _ptr(ws),
                                       _ptr(dgamma), _ptr(dbeta), _ptr(dx), x.numel() // c, c, float(eps), _stream()), "shdr_bn_bwd_f16")
        return dx, dgamma, dbeta
    xr = _new_slot(x.device)            # max |dx| out of the apply pass (the consumer is the input / weight gradient of the conv in front)
    _lib.check(lib.shdr_bn_bwd_ranged_f32(_ptr(dy), _ptr(x), _ptr(_d(y_relu)), _ptr(mean), _ptr(var), _ptr(_d(gamma)), _ptr(ws),
                                          _ptr(dgamma), _ptr(dbeta), _ptr(dx), x.numel() // c, c, float(eps), _ptr(xr), _stream()),
               "shdr_bn_bwd_ranged_f32")
    return _set_range(dx, xr), dgamma, dbeta


def invcrf_decode_bwd(dinv, feat, wfc, table):
    lib = _lib.load()
    dinv, feat, wfc, table = (_chk(_d(t), "t") for t in (dinv, feat, wfc, table))
    b, f = feat.shape
    k = table.shape[0]
    dfeat = torch.empty_like(feat)
    dwfc = torch.zeros_like(wfc)
    dbfc = torch.zeros(11, device=feat.device, dtype=torch.float32)
    _lib.check(lib.shdr_invcrf_decode_bwd_f32(_ptr(dinv), _ptr(feat), _ptr(wfc), _ptr(table), _ptr(dfeat), _ptr(dwfc),
                                              _ptr(dbfc), b, f, k, _stream()), "shdr_invcrf_decode_bwd_f32")
    return dfeat, dwfc, dbfc


def increase_bwd(rf, dout):
    lib = _lib.load()
    rf, dout = _chk(_d(rf), "rf"), _chk(_d(dout), "dout")
    b, k = rf.shape
    drf = torch.empty_like(rf)
    _lib.check(lib.shdr_increase_bwd_f32(_ptr(rf), _ptr(dout), _ptr(drf), b, k, _stream()), "shdr_increase_bwd_f32")
    return drf


def apply_rf_bwd(x, rf, dy, need_dx):
    lib = _lib.load()
    x, rf, dy = _chk(_d(x), "x"), _chk(_d(rf), "rf"), _chk(_d(dy), "dy")
    b = x.shape[0]
    drf = torch.zeros_like(rf)
    dx = torch.empty_like(x) if need_dx else None
    _lib.check(lib.shdr_apply_rf_bwd_f32(_ptr(x), _ptr(rf), _ptr(dy), _ptr(drf), _ptr(dx), b, x.numel() // b, rf.shape[1],
                                         _stream()), "shdr_apply_rf_bwd_f32")
    return drf, dx


def diff_loss(a, b, mode):
    """per-sample mean of (a-b)^2 (mode 0) or |a-b| (mode 1): [B]"""
    if _needs_grad(a, b):
        return AUTOGRAD.diff_loss(a, b, mode)
    lib = _lib.load()
    a, b = _chk(_d(a), "a"), _chk(_d(b), "b")
    if a.shape != b.shape:
        raise ValueError("diff_loss: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    bs = a.shape[0]
    out = torch.empty(bs, device=a.device, dtype=torch.float32)
    _lib.check(lib.shdr_diff_loss_f32(_ptr(a), _ptr(b), _ptr(out), bs, a.numel() // bs, mode, _stream()), "shdr_diff_loss_f32")
    return out


def diff_loss_bwd(a, b, g, mode):
    lib = _lib.load()
    a, b, g = _chk(_d(a), "a"), _chk(_d(b), "b"), _chk(_d(g), "g")
    bs = a.shape[0]
    da = torch.empty_like(a)
    _lib.check(lib.shdr_diff_loss_bwd_f32(_ptr(a), _ptr(b), _ptr(g), _ptr(da), bs, a.numel() // bs, mode, 0, _stream()),
               "shdr_diff_loss_bwd_f32")
    return da


def tv_loss(y):
    """batch-global TV loss (joint_training.py:175-179): tensor [1]"""
    if _needs_grad(y):
        return AUTOGRAD.tv_loss(y)
    lib = _lib.load()
    y = _chk(_d(y), "y")
    n, h, w, c = y.shape
    out = torch.empty(1, device=y.device, dtype=torch.float32)
    _lib.check(lib.shdr_tv_loss_f32(_ptr(y), _ptr(out), n, h, w, c, _stream()), "shdr_tv_loss_f32")
    return out


def tv_loss_bwd(y, g):
    lib = _lib.load()
    y, g = _chk(_d(y), "y"), _chk(_d(g), "g")
    n, h, w, c = y.shape
    dy = torch.empty_like(y)
    _lib.check(lib.shdr_tv_loss_bwd_f32(_ptr(y), _ptr(g), _ptr(dy), n, h, w, c, 0, _stream()), "shdr_tv_loss_bwd_f32")
    return dy


def logc_bwd(dy, x):
    lib = _lib.load()
    dy, x = _chk(_d(dy), "dy"), _chk(_d(x), "x")
    dx = torch.empty_like(x)
    _lib.check(lib.shdr_logc_bwd_f32(_ptr(dy), _ptr(x), _ptr(dx), x.numel(), _stream()), "shdr_logc_bwd_f32")
    return dx


def alpha_mask(x, thr=0.12):
    """alpha [b,h,w,1] = clamp((max_c x - 1 + thr)/thr, 0, 1) (joint_training.py:141-145); no gradient."""
    lib = _lib.load()
    x, npix = _pix3(x, "x")
    alpha = torch.empty(tuple(x.shape[:-1]) + (1,), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_alpha_mask_f32(_ptr(x), _ptr(alpha), npix, float(thr), _stream()), "shdr_alpha_mask_f32")
    return alpha


def alpha_blend_bwd(dA, alpha):
    lib = _lib.load()
    dA, npix = _pix3(dA, "dA")
    alpha = _chk(_d(alpha), "alpha")
    dhal = torch.empty_like(dA)
    _lib.check(lib.shdr_alpha_blend_bwd_f32(_ptr(dA), _ptr(alpha), _ptr(dhal), npix, _stream()), "shdr_alpha_blend_bwd_f32")
    return dhal


def blend_const(base, alpha, hal_bgr, thr=0.12):
    """A = base + alpha * reverse3(hal) with base and alpha constants (joint_training.py:163-165)."""
    if _needs_grad(hal_bgr):
        return AUTOGRAD.blend_const(base, alpha, hal_bgr, thr)
    return alpha_blend(base, hal_bgr, thr)


def vgg_preprocess_bwd(dy):
    lib = _lib.load()
    if _is_h(dy):
        dy = _chkh(_d(dy), "dy")
        dx = torch.empty(tuple(dy.shape[:-1]) + (3,), device=dy.device, dtype=torch.float32)
        _lib.check(lib.shdr_unpack3_f16(_ptr(dy), _ptr(dx), None, None, None, 1, dy.shape[-1], dy.numel() // dy.shape[-1], 1, _stream()),
                   "shdr_unpack3_f16")
        return dx
    dy = _chk(_d(dy), "dy")
    ic = dy.shape[-1]
    npix = dy.numel() // ic
    dx = torch.empty(tuple(dy.shape[:-1]) + (3,), device=dy.device, dtype=torch.float32)
    _lib.check(lib.shdr_vgg_preprocess_bwd_f32(_ptr(dy), _ptr(dx), npix, ic, _stream()), "shdr_vgg_preprocess_bwd_f32")
    return dx


def adam_step(p, g, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
    """in-place Keras Adam update of the flat parameter buffer `p`"""
    lib = _lib.load()
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(_d(t), nm)
    _lib.check(lib.shdr_adam_f32(_ptr(_d(p)), _ptr(_d(g)), _ptr(m), _ptr(v), p.numel(), float(lr_t), float(beta1), float(beta2),
                                 float(eps), float(grad_scale), _stream()), "shdr_adam_f32")
    _mutated(p, m, v)


def _mutated(*tensors):
    """The C ABI writes through raw pointers, so torch does not see an in-place update: bump the version counter (shared by a
    buffer and all its views) of every tensor a kernel has modified -- the per-version caches of the layers (zero-padded /
    x2-scaled filters, folded BatchNorm constants) key on it, and a stale cache would serve pre-update weights to an inference
    call that follows a training step."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def lin_frontend_bwd(img, dF):
    lib = _lib.load()
    h16 = _is_h(dF)
    img, dF = _chk(_d(img), "img"), (_chkh(_d(dF), "dF") if h16 else _chk(_d(dF), "dF"))
    n, h, w, _ = img.shape
    dimg = torch.empty_like(img)
    fn = "shdr_lin_frontend_bwd_f16" if h16 else "shdr_lin_frontend_bwd_f32"
    _lib.check(getattr(lib, fn)(_ptr(img), _ptr(dF), _ptr(dimg), n, h, w, dF.shape[3], _stream()), fn)
    return dimg


def alpha_blend_full_bwd(b_pred, hal_bgr, dA, thr):
    lib = _lib.load()
    b_pred, npix = _pix3(b_pred, "b_pred")
    hal_bgr, dA = _chk(_d(hal_bgr), "hal"), _chk(_d(dA), "dA")
    dB, dhal = torch.empty_like(b_pred), torch.empty_like(b_pred)
    _lib.check(lib.shdr_alpha_blend_full_bwd_f32(_ptr(b_pred), _ptr(hal_bgr), _ptr(dA), _ptr(dB), _ptr(dhal), npix, float(thr),
                                                 _stream()), "shdr_alpha_blend_full_bwd_f32")
    return dB, dhal


def unpack3(y, nout):
    """the first `nout` 3-channel slices of y [..., C]: (y[..., 0:3], y[..., 3:6], ...)"""
    if _needs_grad(y):
        return AUTOGRAD.unpack3(y, nout)
    lib = _lib.load()
    h16 = _is_h(y)
    y = _chkh(_d(y), "y") if h16 else _chk(_d(y), "y")
    c = y.shape[-1]
    npix = y.numel() // c
    outs = [torch.empty(tuple(y.shape[:-1]) + (3,), device=y.device, dtype=torch.float32) for _ in range(nout)]
    p = [_ptr(o) for o in outs] + [None] * (4 - nout)
    if h16:
        _lib.check(lib.shdr_unpack3_f16(_ptr(y), p[0], p[1], p[2], p[3], nout, c, npix, 0, _stream()), "shdr_unpack3_f16")
        return tuple(outs)
    _lib.check(lib.shdr_unpack3_f32(_ptr(y), p[0], p[1], p[2], p[3], nout, c, npix, _stream()), "shdr_unpack3_f32")
    return tuple(outs)


def sample_dot(a, b=None):
    lib = _lib.load()
    a = _chk(_d(a), "a")
    bs = a.shape[0]
    out = torch.empty(bs, device=a.device, dtype=torch.float32)
    _lib.check(lib.shdr_sample_dot_f32(_ptr(a), _ptr(None if b is None else _chk(_d(b), "b")), _ptr(out), bs, a.numel() // bs, _stream()),
               "shdr_sample_dot_f32")
    return out


def mean_norm(r, eps=1e-6, target=0.5):
    """r / (eps + mean over (1,2,3) of r) * target   (finetune_real_dataset.py:170)"""
    if _needs_grad(r):
        return AUTOGRAD.mean_norm(r, eps, target)
    return mean_norm_fwd(r, sample_dot(r), eps, target)


def mean_norm_fwd(r, ssum, eps, target):
    lib = _lib.load()
    r = _chk(_d(r), "r")
    bs = r.shape[0]
    out = torch.empty_like(r)
    _lib.check(lib.shdr_mean_norm_fwd_f32(_ptr(r), _ptr(ssum), _ptr(out), bs, r.numel() // bs, float(eps), float(target), _stream()),
               "shdr_mean_norm_fwd_f32")
    return out


def mean_norm_bwd(g, ssum, gdot, eps, target):
    lib = _lib.load()
    g = _chk(_d(g), "g")
    bs = g.shape[0]
    dr = torch.empty_like(g)
    _lib.check(lib.shdr_mean_norm_bwd_f32(_ptr(g), _ptr(ssum), _ptr(gdot), _ptr(dr), bs, g.numel() // bs, float(eps), float(target),
                                          _stream()), "shdr_mean_norm_bwd_f32")
    return dr


# ---------------------------------------------------------------------------
# native-fp16 path of BASELINE configs[4] (csrc/conv_f16.hip, wgrad_f16.hip, elem_f16.hip)
# ---------------------------------------------------------------------------
def to_half(x):
    lib = _lib.load()
    x = _chk(_d(x), "x")
    y = torch.empty(x.shape, device=x.device, dtype=HALF)
    _lib.check(lib.shdr_cast_f32_to_f16(_ptr(x), _ptr(y), x.numel(), _stream()), "shdr_cast_f32_to_f16")
    return y


def to_float(x):
    lib = _lib.load()
    x = _chk(_d(x), "x", HALF)
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_cast_f16_to_f32(_ptr(x), _ptr(y), x.numel(), _stream()), "shdr_cast_f16_to_f32")
    return y


def pad_channels_h(x, channels):
    """fp32 [..., c] -> fp16 [..., channels] zero-padded (the gradient of a 3-channel head onto an 8-channel group)"""
    lib = _lib.load()
    x = _chk(_d(x), "x")
    c = x.shape[-1]
    y = torch.empty(tuple(x.shape[:-1]) + (channels,), device=x.device, dtype=HALF)
    _lib.check(lib.shdr_pad_channels_f32_to_f16(_ptr(x), _ptr(y), x.numel() // c, c, channels, _stream()), "shdr_pad_channels_f32_to_f16")
    return y


def pack_filter_h(w, c1, c2=0, x2_scale=1.0):
    """fp32 HWIO filter -> the packed fp16 filter of conv2d_h ([k-chunk][Cout][32]; x2_scale folded into the x2 rows); kept ON
    the filter tensor per version for persistent variables, like the packed Winograd filters"""
    key = None
    if _is_persistent(w):
        key = (c1, c2, float(x2_scale))
        cached = _filter_cache_get(w, "_shdr_packed_h", key)
        if cached is not None:
            return cached
    lib = _lib.load()
    wd = _chk(_d(w), "w")
    kh, kw, cin, cout = wd.shape
    if cin != c1 + c2:
        raise ValueError("pack_filter_h: filter has %d input channels, sources have %d+%d" % (cin, c1, c2))
    n = int(lib.shdr_conv2d_packed_filter_elems_f16(kh, kw, c1, c2, cout))
    wp = torch.empty(n, device=wd.device, dtype=HALF)
    _lib.check(lib.shdr_conv2d_pack_filter_f16(_ptr(wd), _ptr(wp), kh, kw, c1, c2, cout, float(x2_scale), _stream()),
               "shdr_conv2d_pack_filter_f16")
    if key is not None:
        _filter_cache_put(w, "_shdr_packed_h", key, wp)
    return wp


def _conv_desc_h(x_shape, c2, khw, cout_gemm, stride, cout_valid, pad=None, out_hw=None):
    n, h, wd, c1 = x_shape
    kh, kw = khw
    ho, pt = same_pad(h, kh, stride)
    wo, pl = same_pad(wd, kw, stride)
    if pad is not None:
        pt, pl = pad
    if out_hw is not None:
        ho, wo = out_hw
    d = _lib.ConvDesc()
    d.N, d.H, d.W, d.C1, d.C2 = n, h, wd, c1, c2
    d.Cout, d.KH, d.KW, d.stride = cout_gemm, kh, kw, stride
    d.cout_valid = cout_valid or cout_gemm
    d.pad_t, d.pad_l, d.Ho, d.Wo = pt, pl, ho, wo
    d.x2_scale = 1.0
    return d


def conv2d_h(x, wp, bias, khw, cout_gemm, stride=1, x2=None, act1=ACT_NONE, cout_valid=None, pad=None, out_hw=None):
    """y = act1(conv(concat[x, x2], wp) + bias) on fp16 feature maps, `wp` from pack_filter_h.  The output is fp16
    [N,Ho,Wo,cout_gemm], or fp32 [N,Ho,Wo,cout_valid] for a narrow head (cout_valid < cout_gemm: image-like tensors stay fp32)."""
    lib = _lib.load()
    x = _chkh(_d(x), "x")
    c2 = 0
    if x2 is not None:
        x2 = _chkh(_d(x2), "x2")
        if x2.shape[:3] != x.shape[:3]:
            raise ValueError("conv2d_h: x2 spatial shape %s != x %s" % (tuple(x2.shape), tuple(x.shape)))
        c2 = x2.shape[3]
    head = cout_valid is not None and cout_valid < cout_gemm
    d = _conv_desc_h(x.shape, c2, khw, cout_gemm, stride, cout_valid if head else None, pad, out_hw)
    d.act1 = act1
    if bias is not None and _chk(_d(bias), "bias").numel() < (cout_valid if head else cout_gemm):
        raise ValueError("conv2d_h: bias too short")
    y = torch.empty((x.shape[0], d.Ho, d.Wo, cout_valid if head else cout_gemm), device=x.device, dtype=torch.float32 if head else HALF)
    _lib.check(lib.shdr_conv2d_fwd_f16(ctypes.byref(d), _ptr(x), _ptr(x2), _ptr(wp), _ptr(_d(bias)), _ptr(y), int(head), _stream()),
               "shdr_conv2d_fwd_f16")
    return y


def conv2d_wgrad_h(x, x2, dz, w_shape, stride=1, x2_scale=1.0, cout_valid=None, out=None):
    """dW [kh,kw,C1+C2,Cout] (fp32) of conv(concat[x, x2_scale*x2], W) from fp16 x / x2 / dz; columns >= cout_valid stay zero.
    `out`: gradient buffer of that shape to accumulate into."""
    lib = _lib.load()
    kh, kw, cin, cout = w_shape
    x, dz = _chkh(_d(x), "x"), _chkh(_d(dz), "dz")
    c1 = x.shape[3]
    c2 = 0 if x2 is None else x2.shape[3]
    if cin != c1 + c2:
        raise ValueError("conv2d_wgrad_h: filter %s does not match the sources (%d+%d channels)" % (tuple(w_shape), c1, c2))
    d = _conv_desc_h(x.shape, c2, (kh, kw), cout, stride, cout_valid)
    d.x2_scale = float(x2_scale)
    if tuple(dz.shape[:3]) != (x.shape[0], d.Ho, d.Wo):
        raise ValueError("conv2d_wgrad_h: dz spatial shape mismatch")
    if out is not None and (tuple(out.shape) != tuple(w_shape) or not out.is_contiguous() or out.dtype != torch.float32):
        raise ValueError("conv2d_wgrad_h: out must be a contiguous float32 tensor of shape %s" % (tuple(w_shape),))
    dw = out if out is not None else torch.zeros(tuple(w_shape), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_conv2d_wgrad_f16(ctypes.byref(d), _ptr(x), 0, _ptr(dz), dz.shape[3], c1, c2, _ptr(dw), _stream()),
               "shdr_conv2d_wgrad_f16")
    if x2 is not None:
        _lib.check(lib.shdr_conv2d_wgrad_f16(ctypes.byref(d), _ptr(_chkh(_d(x2), "x2")), 1, _ptr(dz), dz.shape[3], c1, c2, _ptr(dw),
                                             _stream()), "shdr_conv2d_wgrad_f16")
    return dw


def act_bwd_bias_h(dy, y, act, want_db, out=None):
    """(dz, db) on fp16 tensors: dz = dy * act'(y) (dy itself when act is NONE), db = sum over pixels of dz in fp32 (or None);
    `out`: gradient buffer to accumulate db into"""
    lib = _lib.load()
    dy = _chkh(_d(dy), "dy")
    c = dy.shape[-1]
    db = (out if out is not None else torch.zeros(c, device=dy.device, dtype=torch.float32)) if want_db else None
    if act == ACT_NONE:
        if want_db:
            _lib.check(lib.shdr_act_bwd_bias_f16(_ptr(dy), None, None, _ptr(db), _ptr(_bias_ws(c, dy.device)), dy.numel() // c, c, act,
                                                 _stream()), "shdr_act_bwd_bias_f16")
        return dy, db
    y = _chkh(_d(y), "y")
    dz = torch.empty_like(dy)
    ws = _bias_ws(c, dy.device) if want_db else None
    _lib.check(lib.shdr_act_bwd_bias_f16(_ptr(dy), _ptr(y), _ptr(dz), _ptr(db), _ptr(ws), dy.numel() // c, c, act, _stream()),
               "shdr_act_bwd_bias_f16")
    return dz, db


# ---------------------------------------------------------------------------
# Winograd F(2x2,3x3)
# ---------------------------------------------------------------------------
def winograd_filter(w):
    """U [16, Cin, Cout] = G g G^T of a 3x3 HWIO filter"""
    lib = _lib.load()
    w = _chk(_d(w), "w")
    kh, kw, cin, cout = w.shape
    if (kh, kw) != (3, 3):
        raise ValueError("winograd_filter: 3x3 filters only")
    u = torch.empty((16, cin, cout), device=w.device, dtype=torch.float32)
    _lib.check(lib.shdr_winograd_filter_f32(_ptr(w), _ptr(u), cin, cout, _stream()), "shdr_winograd_filter_f32")
    return u


def winograd_filter_packed(w):
    """U = G g G^T of a 3x3 HWIO filter in the operand order of the fused kernel: a [16, Cin, Cout]-sized tensor whose
    MEMORY is [Cout/64][8 waves][Cin/8][4][64 lanes][4] (see winograd_filter_packed_kernel)"""
    lib = _lib.load()
    w = _chk(_d(w), "w")
    kh, kw, cin, cout = w.shape
    if (kh, kw) != (3, 3) or cin % 8 or cout % 64:
        raise ValueError("winograd_filter_packed: 3x3 filters with Cin %% 8 == 0 and Cout %% 64 == 0 only")
    u = torch.empty((16, cin, cout), device=w.device, dtype=torch.float32)
    _lib.check(lib.shdr_winograd_filter_packed_f32(_ptr(w), _ptr(u), cin, cout, _stream()), "shdr_winograd_filter_packed_f32")
    return u


def conv2d_winograd_fused(x, u, bias=None, act1=ACT_NONE, scale=None, shift=None, act2=ACT_NONE, x2=None, pool=False):
    """3x3 / stride 1 / SAME convolution through the ONE-kernel Winograd F(2x2,3x3) (csrc/winograd_fused.hip);
    `u` from winograd_filter_packed().  `x2`: second source of a channel concatenation [x, x2] (same channel count).
    `pool=True` returns (y, MaxPool2D(2)(y)), the pooled tensor written by the same epilogue (H, W even); `pool="only"`
    returns the pooled tensor alone and y is never stored."""
    lib = _lib.load()
    x, u = _chk(_d(x), "x"), _chk(_d(u), "u")
    n, h, w, c = x.shape
    c2 = 0
    if x2 is not None:
        x2 = _chk(_d(x2), "x2")
        if tuple(x2.shape) != tuple(x.shape):
            raise ValueError("conv2d_winograd_fused: x2 %s must have the shape of x %s" % (tuple(x2.shape), tuple(x.shape)))
        c2 = c
    cout = u.shape[2]
    if u.shape[0] != 16 or u.shape[1] != c + c2 or c % 8 or cout % 64:
        raise ValueError("conv2d_winograd_fused: need u [16, Cin, Cout] with Cin %% 8 == 0 and Cout %% 64 == 0")
    y = None if pool == "only" else torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    yp = None
    if pool:
        if h % 2 or w % 2:
            raise ValueError("conv2d_winograd_fused: pool=True needs even H, W")
        yp = torch.empty((n, h // 2, w // 2, cout), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_conv2d_winograd_fused2_f32(_ptr(x), _ptr(x2), _ptr(u), _ptr(_d(bias)), _ptr(_d(scale)), _ptr(_d(shift)),
                                                   _ptr(y), _ptr(yp), n, h, w, c, c2, cout, act1, act2, _stream()),
               "shdr_conv2d_winograd_fused2_f32")
    return yp if pool == "only" else ((y, yp) if pool else y)


def _packed_filter(w):
    """winograd_filter_packed(w), kept ON the filter tensor per version for persistent variables (requires_grad leaves: the
    layers' kernels): an inference step then packs nothing (33 launches per step).  Temporaries (transposed dgrad filters)
    are packed per call.  Every kernel that rewrites a variable through a raw pointer bumps its version (_mutated, KerasAdam)."""
    if not _is_persistent(w):                      # _shdr_const: frozen weights (VGG16)
        return winograd_filter_packed(w)
    cached = _filter_cache_get(w, "_shdr_wino", "u")
    if cached is not None:
        return cached
    return _filter_cache_put(w, "_shdr_wino", "u", winograd_filter_packed(w))


def conv2d_maxpool2(x, w, bias=None, act1=ACT_NONE, keep_y=True, proj=None):
    """(y, MaxPool2D(2)(y)) with y = act1(conv(x, w) + bias): ONE launch where the planned kernel is the fused Winograd kernel
    (its epilogue writes the pooled tensor too), conv + pooling launch otherwise -- decided below the C ABI.
    keep_y=False returns the pooled tensor only.
    proj [3, Cout]: (sum_c proj[j, c] y[..., c], pooled) -- y itself is not written -- or None when the layer's planned kernel
    cannot form the projection."""
    fused = not _is_h(x) and not _needs_grad(x, w, bias) and PRECISION in ("fp32", "fp16") and WINOGRAD and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0
    if proj is not None:
        if not fused or PRECISION != "fp32":
            return None
        return _conv2d_raw(x, w, bias, 1, None, 1.0, act1, None, None, None, ACT_NONE, ALGO_AUTO, None, None, None, None, 0, True, proj=proj)
    if fused:
        return _conv2d_raw(x, w, bias, 1, None, 1.0, act1, None, None, None, ACT_NONE, ALGO_AUTO, None, None, None, None, 0,
                           True if keep_y else "only")
    y = conv2d(x, w, bias, act1=act1)
    if not keep_y:
        return maxpool2(y)
    ya, yb = fork(y)                   # the skip connection and the pooling both consume y
    return ya, maxpool2(yb)


def conv2d_avgpool2(x, w, bias=None, act1=ACT_NONE, x2=None):
    """(y, AveragePooling2D(2)(y)) with y = act1(conv(concat[x, x2], w) + bias): the encoder levels of the two U-Nets
    (dequantization_net.py:9-13 pools each level's output for the next one and keeps it as the skip connection).  Without a tape
    the pooled tensor comes out of the conv kernel's own epilogue on the split-operand plans -- decided below the C ABI."""
    if not _is_h(x) and not _needs_grad(x, w, bias, x2) and PRECISION in ("fp32", "fp16") and WINOGRAD and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0:
        return _conv2d_raw(x, w, bias, 1, x2, 1.0, act1, None, None, None, ACT_NONE, ALGO_AUTO, None, None, None, None, 0, "avg")
    y = conv2d(x, w, bias, x2=x2, act1=act1)
    ya, yb = fork(y)                   # the skip connection and the pooling both consume y
    return ya, avgpool2(yb)


def conv2d_winograd(x, u, bias=None, act1=ACT_NONE, scale=None, shift=None, act2=ACT_NONE):
    """3x3 / stride 1 / SAME convolution through Winograd F(2x2,3x3); `u` from winograd_filter()."""
    lib = _lib.load()
    x, u = _chk(_d(x), "x"), _chk(_d(u), "u")
    n, h, w, c = x.shape
    cout = u.shape[2]
    if u.shape[0] != 16 or u.shape[1] != c or c % 32 or cout % 16:
        raise ValueError("conv2d_winograd: need u [16, Cin, Cout] with Cin %% 32 == 0 and Cout %% 16 == 0")
    rows = int(lib.shdr_winograd_tiles(n, h, w))
    v = torch.empty((16, rows // 16, 16, c), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_winograd_input_f32(_ptr(x), _ptr(v), n, h, w, c, _stream()), "shdr_winograd_input_f32")
    # 16 GEMMs [rows, Cin] @ [Cin, Cout] as one "16-image" 1x1 conv; image xi uses filter plane u[xi]
    m = conv2d(v, u[0].view(1, 1, c, cout), w_batch_stride=c * cout)
    y = torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_winograd_output_f32(_ptr(m), _ptr(y), _ptr(_d(bias)), _ptr(_d(scale)), _ptr(_d(shift)), n, h, w, cout,
                                            act1, act2, _stream()), "shdr_winograd_output_f32")
    return y


# ---------------------------------------------------------------------------
# image plumbing of the inference tool (test_real_refinement.py:119-155), see csrc/imageio.hip
# ---------------------------------------------------------------------------
def u8_to_unit(img_u8, reverse_channels=False):
    """uint8 [..., 3] on the device -> float32 in [0,1]; optionally reversing the channel order (np.flip(img, -1))"""
    lib = _lib.load()
    if not (isinstance(img_u8, torch.Tensor) and img_u8.is_cuda and img_u8.dtype == torch.uint8 and img_u8.is_contiguous()
            and img_u8.shape[-1] == 3):
        raise TypeError("u8_to_unit: expected a contiguous uint8 device tensor [..., 3]")
    y = torch.empty(img_u8.shape, device=img_u8.device, dtype=torch.float32)
    _lib.check(lib.shdr_u8_to_unit_f32(_ptr(img_u8), _ptr(y), img_u8.numel() // 3, int(reverse_channels), _stream()),
               "shdr_u8_to_unit_f32")
    return set_bound(y, 1.0)


def resize_cubic(x, out_hw):
    """cv2.resize(x, (Wo, Ho), interpolation=cv2.INTER_CUBIC) on an NHWC float tensor"""
    lib = _lib.load()
    x = _chk(_d(x), "x")
    n, h, w, c = x.shape
    ho, wo = int(out_hw[0]), int(out_hw[1])
    y = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_resize_cubic_f32(_ptr(x), _ptr(y), n, h, w, c, ho, wo, _stream()), "shdr_resize_cubic_f32")
    return y


def pad_symmetric(x, pad):
    """np.pad(x, ((0,0),(pad,pad),(pad,pad),(0,0)), 'symmetric')"""
    lib = _lib.load()
    x = _chk(_d(x), "x")
    n, h, w, c = x.shape
    y = torch.empty((n, h + 2 * pad, w + 2 * pad, c), device=x.device, dtype=torch.float32)
    _lib.check(lib.shdr_pad_symmetric_f32(_ptr(x), _ptr(y), n, h, w, c, int(pad), _stream()), "shdr_pad_symmetric_f32")
    return y


def rgbe_encode(x, reverse_channels=False):
    """float [..., 3] -> Radiance RGBE bytes uint8 [..., 4]"""
    lib = _lib.load()
    x, npix = _pix3(x, "x")
    y = torch.empty(tuple(x.shape[:-1]) + (4,), device=x.device, dtype=torch.uint8)
    _lib.check(lib.shdr_rgbe_encode_f32(_ptr(x), _ptr(y), npix, int(reverse_channels), _stream()), "shdr_rgbe_encode_f32")
    return y


# ---------------------------------------------------------------------------
# camera-pipeline simulator (joint_training.py:26-69), see csrc/camera.hip
# ---------------------------------------------------------------------------
def camera_expose(hdr, t, seed):
    """(hdr_t, clipped_hdr_t) of joint_training.py:30-43: exposure, shot + read noise, relu, clip to [0,1]"""
    lib = _lib.load()
    hdr, npix = _pix3(hdr, "hdr")
    t = _chk(_d(t), "t").reshape(-1)
    n, h, w, _ = hdr.shape
    if t.numel() != n:
        raise ValueError("camera_expose: t must have one exposure per sample")
    hdr_t, clipped = torch.empty_like(hdr), torch.empty_like(hdr)
    _lib.check(lib.shdr_camera_expose_f32(_ptr(hdr), _ptr(t), _ptr(hdr_t), _ptr(clipped), n, h, w, int(seed) & (2 ** 64 - 1),
                                          _stream()), "shdr_camera_expose_f32")
    return hdr_t, clipped


def jpeg_round_trip(ldr, quality):
    """(jpeg_img_float, loss_mask [b,1,1,1]) of joint_training.py:46-63; `quality`: one int per sample"""
    lib = _lib.load()
    ldr, npix = _pix3(ldr, "ldr")
    n, h, w, _ = ldr.shape
    q = torch.as_tensor(list(quality), dtype=torch.int32).to(ldr.device) if not isinstance(quality, torch.Tensor) else quality
    if q.dtype != torch.int32 or q.numel() != n or not q.is_cuda:
        raise ValueError("jpeg_round_trip: quality must be %d int32 values" % n)
    jpeg = torch.empty_like(ldr)
    mask = torch.empty((n, 1, 1, 1), device=ldr.device, dtype=torch.float32)
    planes = torch.empty(n * h * w * 3 // 2, device=ldr.device, dtype=torch.uint8)
    counts = torch.empty(2 * n, device=ldr.device, dtype=torch.int32)
    _lib.check(lib.shdr_jpeg_round_trip_f32(_ptr(ldr), _ptr(q), _ptr(jpeg), _ptr(mask), _ptr(planes), _ptr(counts), n, h, w,
                                            _stream()), "shdr_jpeg_round_trip_f32")
    return jpeg, mask


def flip_rot90(x, flip, rot, divisor=1.0):
    """per-sample rot90(flip_left_right(x) if flip else x, k) / divisor on square NHWC images; flip / rot: int32 [N] on the device"""
    lib = _lib.load()
    x = _chk(_d(x), "x")
    n, h, w, c = x.shape
    if h != w:
        raise ValueError("flip_rot90: square images only, got %dx%d" % (h, w))
    for t, nm in ((flip, "flip"), (rot, "rot")):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int32 and t.numel() == n):
            raise TypeError("flip_rot90: %s must be %d int32 values on the device" % (nm, n))
    y = torch.empty_like(x)
    _lib.check(lib.shdr_flip_rot90_f32(_ptr(x), _ptr(y), _ptr(flip), _ptr(rot), n, h, c, float(divisor), _stream()),
               "shdr_flip_rot90_f32")
    return y
