"""Dequantization-Net on MI355X -- drop-in for the reference module of the same name.

Call surface (dequantization_net.py:31-65 of the reference): `model()` with no
required arguments, `net(x, training=...)` on NHWC float32 in [0,1],
`net.trainable_variables`.  4-level U-Net 3->3 channels, leaky-relu(0.1),
`tanh` head added to the input.  `training` is ignored (no BatchNorm), as in
the reference.  All arithmetic runs in the libshdr HIP kernels.
"""
import torch

try:
    from . import _ops as K
    from ._layers import Layer, Conv2D
except ImportError:
    import _ops as K
    from _layers import Layer, Conv2D


class down(Layer):
    """AveragePooling2D(2) -> conv+lrelu -> conv+lrelu (dequantization_net.py:4-15)."""

    def __init__(self, inChannels, outChannels, kernel_size=(3, 3), device=None):
        super().__init__()
        self.conv1 = Conv2D(inChannels, outChannels, kernel_size, device=device)
        self.conv2 = Conv2D(outChannels, outChannels, kernel_size, device=device)

    def call(self, x):
        x = K.avgpool2(x)
        x = self.conv1(x, act1=K.ACT_LRELU)
        return self.conv2(x, act1=K.ACT_LRELU)

    def call_pooled(self, xp):
        """the block on an input that is pooled already (the previous level's conv wrote it from its epilogue); returns this
        level's output and its own AveragePooling2D(2) for the next level"""
        x = self.conv1(xp, act1=K.ACT_LRELU)
        return self.conv2.call_avgpool2(x, act1=K.ACT_LRELU)


class up(Layer):
    """bilinear 2x -> conv+lrelu -> conv(concat[x, skip])+lrelu (dequantization_net.py:17-29).
    The concat is never materialised: the conv kernel reads two sources."""

    def __init__(self, inChannels, skipChannels, outChannels, kernel_size=3, device=None):
        super().__init__()
        self.conv1 = Conv2D(inChannels, outChannels, kernel_size, device=device)
        self.conv2 = Conv2D(outChannels + skipChannels, outChannels, kernel_size, device=device)

    def call(self, x, skpCn):
        x = self.conv1.call_up2(x, act1=K.ACT_LRELU)       # resize fused into the conv without a tape (K.conv2d_up2)
        return self.conv2(x, x2=skpCn, act1=K.ACT_LRELU)


class _unet(Layer):
    def __init__(self, in_channels, bottleneck, device=None):
        super().__init__()
        self.conv1 = Conv2D(in_channels, 16, (7, 7), device=device)
        self.conv2 = Conv2D(16, 16, (7, 7), device=device)
        self.d2 = down(16, 32, (5, 5), device=device)
        self.d3 = down(32, 64, (3, 3), device=device)
        self.d4 = down(64, 128, (3, 3), device=device)
        self.enc = down(128, bottleneck, (3, 3), device=device)
        self.u4 = up(bottleneck, 128, 128, device=device)
        self.u3 = up(128, 64, 64, device=device)
        self.u2 = up(64, 32, 32, device=device)
        self.u1 = up(32, 16, 16, device=device)
        self.out = Conv2D(16, 3, (3, 3), device=device)

    def _trunk(self, input_images):
        cin = input_images.shape[-1]
        if cin == 3 and K.native_fp16():     # BASELINE configs[4]: fp16 feature maps from here on (3 -> 8 channels = one 16-byte group)
            x = self.conv1.call_padded(K.pack3([input_images], 8, K.HALF), cin_pad=8, act1=K.ACT_LRELU)
        elif cin == 3:    # 3 -> 4 channels (zero) so that the 7x7 conv runs on the MFMA tile
            x = self.conv1.call_padded(K.pack3([input_images], 4), cin_pad=4, act1=K.ACT_LRELU)
        elif cin % 4 == 0 and cin != self.conv1.kernel.shape[2]:   # caller passed a zero-padded input
            x = self.conv1.call_padded(input_images, cin_pad=cin, act1=K.ACT_LRELU)
        else:
            x = self.conv1(input_images, act1=K.ACT_LRELU)
        # each encoder output feeds its skip connection and, average-pooled, the next level (dequantization_net.py:9-10): the
        # producing conv writes both tensors (K.conv2d_avgpool2; with a tape: conv, fork, pooling launch)
        s1, p1 = self.conv2.call_avgpool2(x, act1=K.ACT_LRELU)
        s2, p2 = self.d2.call_pooled(p1)
        s3, p3 = self.d3.call_pooled(p2)
        s4, p4 = self.d4.call_pooled(p3)
        x = self.enc.conv2(self.enc.conv1(p4, act1=K.ACT_LRELU), act1=K.ACT_LRELU)
        x = self.u4(x, s4)
        x = self.u3(x, s3)
        x = self.u2(x, s2)
        return self.u1(x, s1)


class model(_unet):
    def __init__(self, strides=(1, 1), padding="SAME", device=None):
        super().__init__(3, 256, device=device)

    def call(self, input_images, training="training"):
        x = self._trunk(input_images)
        if x.requires_grad and torch.is_grad_enabled():   # taped: the residual add is a separate op
            return K.add(self.out.call_padded(x, cout_pad=16, act1=K.ACT_TANH), input_images)
        # tanh(out(x)) + input  (dequantization_net.py:62-63) fused into the conv epilogue
        return self.out.call_padded(x, cout_pad=16, act1=K.ACT_TANH, residual=input_images)
