"""Linearization-Net on MI355X -- drop-in for the reference module of the same name.

linearization_net.py:303-392 of the reference: sobel + spatial-aware soft
histogram front end (93 channels), ResNet-style `crfFeatureNet`, EMoR PCA
decoder (`AEInvcrfDecodeNet`) and the monotone fix-up `_increase`; maps an NHWC
float32 image in [0,1] to an inverse CRF [b, 1024].

MI355X design: the 93-channel feature tensor is produced by one fused kernel
and zero-padded to 96 channels so that the 7x7/stride-2 conv1 runs on the
fp32-MFMA implicit-GEMM kernel with 128-byte channel chunks; the filter is
zero-padded to match (cached per parameter version).  Inference BatchNorm,
relu and the residual adds are fused into the conv epilogues.
"""
import os

import numpy as np
import torch

try:
    from . import _ops as K
    from ._layers import Layer, Conv2D, BatchNormalization, Dense, is_training, taping
except ImportError:
    import _ops as K
    from _layers import Layer, Conv2D, BatchNormalization, Dense, is_training, taping

_HERE = os.path.dirname(os.path.abspath(__file__))
FRONTEND_CHANNELS = 93          # linearization_net.py:322
FRONTEND_CHANNELS_PADDED = 96   # MFMA-friendly (3 x 32-channel chunks)


def _conv_bn(conv, norm, x, relu, residual=None, cin_pad=None, training=False):
    """conv -> BatchNorm [-> + residual] [-> relu].

    inference: one kernel (BN folded into the conv epilogue, residual and relu fused);
    training : conv, batch-statistics BN (+relu), residual join relu(a + b) as separate taped ops."""
    if is_training(training):
        z = conv(x) if cin_pad is None else conv.call_padded(x, cin_pad=cin_pad)
        if residual is None:
            return norm.train_apply(z, relu=relu)
        y = norm.train_apply(z, relu=False)
        return K.AUTOGRAD.add_relu(residual, y) if relu else K.add(residual, y)
    if taping(x, residual, conv.kernel, conv.bias, norm.gamma, norm.beta):
        # inference mode while a tape records (frozen statistics): conv and the normalisation / join are separate tape entries
        z = conv(x) if cin_pad is None else conv.call_padded(x, cin_pad=cin_pad)
        return norm.frozen_apply(z, residual, relu)
    scale, shift = norm.folded()
    kw = dict(scale=scale, shift=shift, residual=residual, act2=K.ACT_RELU if relu else K.ACT_NONE)
    if cin_pad is None:
        return K.conv2d(x, conv.kernel, conv.bias, stride=conv.strides, **kw)
    return conv.call_padded(x, cin_pad=cin_pad, **kw)     # (under a tape the zero-padding of the filter is part of the tape)


class resBlock_type1(Layer):
    """linearization_net.py:6-48: projection shortcut (conv1x1 -> BN) + bottleneck branch."""

    def __init__(self, in_channels, branch1_filter, branch2_filters, strides=(1, 1), device=None):
        super().__init__()
        s = strides if isinstance(strides, int) else strides[0]
        self.conv1 = Conv2D(in_channels, branch1_filter, 1, s, use_bias=False, device=device)
        self.norm1 = BatchNormalization(branch1_filter, device=device)
        self.conv2 = Conv2D(in_channels, branch2_filters[0], 1, s, use_bias=False, device=device)
        self.norm2 = BatchNormalization(branch2_filters[0], device=device)
        self.conv3 = Conv2D(branch2_filters[0], branch2_filters[1], 3, 1, use_bias=False, device=device)
        self.norm3 = BatchNormalization(branch2_filters[1], device=device)
        self.conv4 = Conv2D(branch2_filters[1], branch2_filters[2], 1, 1, use_bias=False, device=device)
        self.norm4 = BatchNormalization(branch2_filters[2], device=device)

    def call(self, x, training="training"):
        t = training
        x, xb = K.fork(x)                  # shortcut branch and bottleneck branch
        norm1 = _conv_bn(self.conv1, self.norm1, x, relu=False, training=t)
        act2 = _conv_bn(self.conv2, self.norm2, xb, relu=True, training=t)
        act3 = _conv_bn(self.conv3, self.norm3, act2, relu=True, training=t)
        return _conv_bn(self.conv4, self.norm4, act3, relu=True, residual=norm1, training=t)  # relu(norm1 + norm4)


class resBlock_type2(Layer):
    """linearization_net.py:50-83: identity shortcut + bottleneck branch."""

    def __init__(self, in_channels, filters, device=None):
        super().__init__()
        self.conv1 = Conv2D(in_channels, filters[0], 1, 1, use_bias=False, device=device)
        self.norm1 = BatchNormalization(filters[0], device=device)
        self.conv2 = Conv2D(filters[0], filters[1], 3, 1, use_bias=False, device=device)
        self.norm2 = BatchNormalization(filters[1], device=device)
        self.conv3 = Conv2D(filters[1], filters[2], 1, 1, use_bias=False, device=device)
        self.norm3 = BatchNormalization(filters[2], device=device)

    def call(self, x, training="training"):
        t = training
        x, xb = K.fork(x)                  # identity shortcut and bottleneck branch
        act1 = _conv_bn(self.conv1, self.norm1, xb, relu=True, training=t)
        act2 = _conv_bn(self.conv2, self.norm2, act1, relu=True, training=t)
        return _conv_bn(self.conv3, self.norm3, act2, relu=True, residual=x, training=t)  # relu(x + norm3)


class crfFeatureNet(Layer):
    """linearization_net.py:85-118."""

    def __init__(self, padding="SAME", device=None):
        super().__init__()
        self.conv1 = Conv2D(FRONTEND_CHANNELS, 64, (7, 7), (2, 2), device=device)
        self.norm1 = BatchNormalization(64, device=device)
        self.res1 = resBlock_type1(64, 256, [64, 64, 256], (1, 1), device=device)
        self.res2 = resBlock_type2(256, [64, 64, 256], device=device)
        self.res3 = resBlock_type2(256, [64, 64, 256], device=device)
        self.res4 = resBlock_type1(256, 512, [128, 128, 512], (2, 2), device=device)
        self.res5 = resBlock_type2(512, [128, 128, 512], device=device)

    def call(self, ldr, training="training"):
        """`ldr` is the front-end tensor with 93 or (zero-padded) 96 channels."""
        cin = ldr.shape[-1]
        x = _conv_bn(self.conv1, self.norm1, ldr, relu=True,
                     cin_pad=None if cin == FRONTEND_CHANNELS else cin, training=training)
        x = K.maxpool3s2(x)
        x = self.res1(x, training)
        x = self.res2(x, training)
        x = self.res3(x, training)
        x = self.res4(x, training)
        x = self.res5(x, training)
        return K.global_avg_pool(x)


def load_invemor_table(device):
    """g0 | hinv(1..11) as a [1024,12] tensor.

    Like the reference (linearization_net.py:217-227) an `invemor.txt` in the
    current working directory takes precedence; otherwise the packaged binary
    conversion of the same table is used (tools/convert_invemor.py)."""
    if os.path.exists("invemor.txt"):
        with open("invemor.txt") as f:
            lines = [ln.strip() for ln in f]

        def block(tag):
            i = lines.index(tag) + 1
            vals = []
            for ln in lines[i:i + 256]:
                vals.extend(ln.split())
            return np.asarray(vals, dtype=np.float32)

        tab = np.stack([block("g0 =")] + [block("hinv(%d)=" % (j + 1)) for j in range(11)], axis=-1)
    else:
        tab = np.load(os.path.join(_HERE, "data", "invemor_g0_hinv11.npy"))
    return torch.from_numpy(np.ascontiguousarray(tab, dtype=np.float32)).to(device)


class AEInvcrfDecodeNet(Layer):
    """linearization_net.py:173-268: Dense(11) then invcrf = g0 + HINV @ w."""

    def __init__(self, n_digit=2, device=None):
        super().__init__()
        self.s = 1024
        self.n_p = 12
        self.fc = Dense(512, self.n_p - 1, device=device)
        self._table = None

    def table(self, device):
        if self._table is None or self._table.device != device:
            self._table = load_invemor_table(device)
        return self._table

    def call(self, feature):
        return K.invcrf_decode(feature, self.fc.kernel, self.fc.bias, self.table(feature.device))


class model(Layer):
    def __init__(self, device=None):
        super().__init__()
        self.crf_feature_net = crfFeatureNet(device=device)
        self.ae_invcrf_decode_net = AEInvcrfDecodeNet(device=device)

    def call(self, img, training="training"):
        feat_in = K.lin_frontend(img, FRONTEND_CHANNELS_PADDED)   # linearization_net.py:312-322, fused
        feature = self.crf_feature_net(feat_in, training)
        invcrf = self.ae_invcrf_decode_net(feature)
        return self._increase(invcrf)

    def histogram_layer(self, img, max_bin):
        """Spatial-aware soft histogram (linearization_net.py:336-350)."""
        return K.soft_hist(img, max_bin)

    @staticmethod
    def _increase(rf):
        """linearization_net.py:368-392."""
        return K.increase(rf)
