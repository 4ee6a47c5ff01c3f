"""Keras-like parameter containers for the drop-in network modules.

Mirrors the parts of tf.keras.Model / layers the reference's callers rely on
(SURVEY.md section 8b): `net(x, training=bool)`, `net.trainable_variables` (ordered:
attribute-creation order, kernel before bias, gamma before beta), HWIO kernels,
Keras default initialisers.  Parameters are torch tensors in device memory;
all arithmetic goes through `_ops` (the HIP kernels).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

try:
    from . import _ops as K
except ImportError:
    import _ops as K

BN_EPS = 1e-3        # tf.keras.layers.BatchNormalization default epsilon
BN_MOMENTUM = 0.99   # ... default momentum


def default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class Layer:
    """Tracks child layers and variables in attribute-creation order, like Keras."""

    def __init__(self):
        object.__setattr__(self, "_children", [])   # (name, Layer)
        object.__setattr__(self, "_vars", [])       # (name, tensor, trainable)

    def __setattr__(self, name, value):
        if isinstance(value, Layer) and not name.startswith("_"):
            self._children.append((name, value))
        object.__setattr__(self, name, value)

    def add_weight(self, name, value, trainable=True):
        value.requires_grad_(trainable)
        self._vars.append((name, value, trainable))
        return value

    # -- Keras-compatible views ------------------------------------------------
    def named_weights(self, prefix=""):
        """Ordered (name, tensor, trainable) triples: own variables first, then children."""
        out = [(prefix + n, t, tr) for n, t, tr in self._vars]
        for cname, child in self._children:
            out.extend(child.named_weights(prefix + cname + "."))
        return out

    @property
    def trainable_variables(self):
        return [t for _, t, tr in self.named_weights() if tr]

    @property
    def non_trainable_variables(self):
        return [t for _, t, tr in self.named_weights() if not tr]

    @property
    def weights(self):
        return [t for _, t, _ in self.named_weights()]

    def state_dict(self):
        return {n: t.detach() for n, t, _ in self.named_weights()}

    def load_numpy(self, params, strict=True):
        """Copy a {name: ndarray} dict (oracle naming) into the parameters."""
        own = dict((n, t) for n, t, _ in self.named_weights())
        if strict and set(own) != set(params):
            raise KeyError("parameter names differ: missing %s, unexpected %s"
                           % (sorted(set(own) - set(params)), sorted(set(params) - set(own))))
        with torch.no_grad():
            for n, t in own.items():
                if n in params:
                    v = torch.as_tensor(np.asarray(params[n]), dtype=torch.float32)
                    if tuple(v.shape) != tuple(t.shape):
                        raise ValueError("%s: shape %s != %s" % (n, tuple(v.shape), tuple(t.shape)))
                    t.copy_(v)
        return self

    def to(self, device):
        for n, t, _ in self.named_weights():
            t.data = t.data.to(device)
            K.clear_filter_caches(t)      # `.data =` does not bump the version: the derived forms kept on the tensor live on the old device
        self._drop_derived()
        return self

    def _drop_derived(self):
        """forget per-version derived tensors kept on the layers themselves (padded / composed filters)"""
        for attr in ("_padded", "_tail", "_tail_proj"):
            if getattr(self, attr, None) is not None:
                object.__setattr__(self, attr, None)
        for _, child in self._children:
            child._drop_derived()

    def __call__(self, *args, **kwargs):
        if K._RANGE_SCOPES:
            return self.call(*args, **kwargs)
        with K.range_scope():          # a model called on its own: its range slots come out of one zeroed slab (K: "range slots")
            return self.call(*args, **kwargs)


def _glorot_uniform(shape, fan_in, fan_out, device):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, dtype=torch.float32) * 2.0 - 1.0).mul_(lim).to(device)


class Conv2D(Layer):
    """tf.keras.layers.Conv2D(filters, kernel_size, strides, padding='SAME', use_bias)."""

    def __init__(self, in_channels, filters, kernel_size, strides=1, use_bias=True, device=None):
        super().__init__()
        device = device or default_device()
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.strides = strides if isinstance(strides, int) else strides[0]
        self.kernel = self.add_weight("kernel", _glorot_uniform((k, k, in_channels, filters),
                                                                k * k * in_channels, k * k * filters, device))
        self.bias = self.add_weight("bias", torch.zeros(filters, device=device)) if use_bias else None
        self._padded = None  # (version, cin_pad, tensor)

    def kernel_padded(self, cin_pad=None, cout_pad=None):
        """Kernel zero-padded along the input- and/or output-channel axis (cached per parameter
        version).  Padding never changes the result: extra input channels meet zero taps, extra
        output channels are not stored (`cout_valid`)."""
        k = self.kernel.detach()
        cin_pad = cin_pad or k.shape[2]
        cout_pad = cout_pad or k.shape[3]
        key = (self.kernel._version, cin_pad, cout_pad)
        if self._padded is None or self._padded[0] != key:
            pad = torch.zeros((k.shape[0], k.shape[1], cin_pad, cout_pad), device=k.device, dtype=k.dtype)
            pad[:, :, :k.shape[2], :k.shape[3]] = k
            pad = pad.contiguous()
            pad._shdr_const = True         # one tensor per parameter version: the library's prepared form is cached on it (_ops._prepared_filter)
            self._padded = (key, pad)
        return self._padded[1]

    def call_padded(self, x, cin_pad=None, cout_pad=None, **kw):
        """Run on the MFMA tile with a zero-padded filter; stores only the true output channels."""
        if torch.is_grad_enabled() and self.kernel.requires_grad:
            # training: the padding is part of the tape (F.pad is parameter-sized plumbing), so the
            # gradient of the padded filter is sliced back onto the variable
            k = self.kernel
            w = F.pad(k, (0, (cout_pad or k.shape[3]) - k.shape[3], 0, (cin_pad or k.shape[2]) - k.shape[2]))
        else:
            w = self.kernel_padded(cin_pad, cout_pad)
        return K.conv2d(x, w, self.bias, stride=self.strides, cout_valid=self.kernel.shape[3], **kw)

    def call(self, x, **kw):
        return K.conv2d(x, self.kernel, self.bias, stride=self.strides, **kw)

    def call_avgpool2(self, x, **kw):
        """(y, AveragePooling2D(2)(y)) of this layer: the pooled tensor from the conv kernel's epilogue where the plan allows it"""
        return K.conv2d_avgpool2(x, self.kernel, self.bias, **kw)

    def call_up2(self, x, **kw):
        """this layer applied to tf.image.resize(x, 2x, BILINEAR): one fused kernel where the library's plan allows it"""
        return K.conv2d_up2(x, self.kernel, self.bias, **kw)          # (kw may carry proj=: None comes back when it cannot be fused)


class BatchNormalization(Layer):
    """tf.keras.layers.BatchNormalization() with Keras defaults (eps 1e-3, momentum 0.99)."""

    def __init__(self, channels, device=None):
        super().__init__()
        device = device or default_device()
        self.gamma = self.add_weight("gamma", torch.ones(channels, device=device))
        self.beta = self.add_weight("beta", torch.zeros(channels, device=device))
        self.moving_mean = self.add_weight("moving_mean", torch.zeros(channels, device=device), trainable=False)
        self.moving_variance = self.add_weight("moving_variance", torch.ones(channels, device=device), trainable=False)
        self._folded = None

    def folded(self):
        """(scale, shift) of the inference transform, cached per parameter version:
        scale = gamma / sqrt(var + eps), shift = beta - mean * scale."""
        ver = (self.gamma._version, self.beta._version, self.moving_mean._version, self.moving_variance._version)
        if self._folded is None or self._folded[0] != ver:
            with torch.no_grad():
                scale = self.gamma.detach() * torch.rsqrt(self.moving_variance + BN_EPS)
                shift = self.beta.detach() - self.moving_mean * scale
            self._folded = (ver, scale.contiguous(), shift.contiguous())
        return self._folded[1], self._folded[2]

    def train_apply(self, x, relu=False):
        """Training mode: normalise with the batch statistics (biased variance), update the moving
        statistics in place (momentum 0.99, unbiased variance), optional fused relu."""
        return K.AUTOGRAD.batch_norm_train(x, self.gamma, self.beta, self.moving_mean, self.moving_variance,
                                           BN_EPS, BN_MOMENTUM, relu)


    def frozen_apply(self, x, residual=None, relu=False):
        """Inference mode on a gradient tape (`net(x, training=False)` while fine-tuning with frozen statistics): normalise with
        the moving statistics [+ residual] [+ relu]; gamma / beta and x (and the residual) receive gradients."""
        return K.AUTOGRAD.batch_norm_frozen(x, self.gamma, self.beta, self.moving_mean, self.moving_variance, BN_EPS,
                                            residual, K.ACT_RELU if relu else K.ACT_NONE)


class Dense(Layer):
    """tf.keras.layers.Dense(units)."""

    def __init__(self, in_features, units, device=None):
        super().__init__()
        device = device or default_device()
        self.kernel = self.add_weight("kernel", _glorot_uniform((in_features, units), in_features, units, device))
        self.bias = self.add_weight("bias", torch.zeros(units, device=device))


def is_training(training):
    """The reference passes `training="training"` (a truthy string) by default."""
    return bool(training)


def taping(*tensors):
    """True when the call is being recorded for backward (gradients enabled and some input needs them)."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
