"""Refinement-Net on MI355X -- drop-in for the reference module of the same name.

refinement_net.py:31-66 of the reference: the Dequantization-Net U-Net with a
9-channel input [A, B, C], bottleneck 128, no tanh; output
relu(input[..., 0:3] + conv).  `training` is ignored (no BatchNorm).

The input may also be given zero-padded to 12 channels ([A, B, C, 0]): the first
7x7 conv then runs on the MFMA tile instead of the VALU fallback.
"""
import torch

try:
    from . import _ops as K
    from .dequantization_net import _unet, down, up  # noqa: F401  (same blocks, refinement_net.py:4-29)
except ImportError:
    import _ops as K
    from dequantization_net import _unet, down, up  # noqa: F401


class model(_unet):
    def __init__(self, strides=(1, 1), padding="SAME", device=None):
        super().__init__(9, 128, device=device)

    def call(self, input_images, training="training"):
        x = self._trunk(input_images)
        if x.requires_grad and torch.is_grad_enabled():   # taped: slice + residual join as separate ops
            (first3,) = K.unpack3(input_images, 1)
            return K.AUTOGRAD.add_relu(first3, self.out.call_padded(x, cout_pad=16))
        # relu(input[..., 0:3] + out(x))  (refinement_net.py:63-66): residual read with channel stride 9
        return self.out.call_padded(x, cout_pad=16, residual=input_images, act2=K.ACT_RELU)
