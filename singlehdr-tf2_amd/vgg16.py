"""VGG16 feature extractor (conv1_1 .. pool3) on MI355X -- drop-in for the reference `vgg16`.

vgg16.py:56-133 of the reference: frozen filters from `vgg16.npy`
(dict name -> [HWIO kernel, bias]); input RGB in [0,1]; returns
(pool1, pool2, pool3).  The weights are constants (no trainable variables).
"""
import inspect
import os

import numpy as np
import torch

try:
    from . import _ops as K
    from ._layers import Layer, default_device
except ImportError:
    import _ops as K
    from _layers import Layer, default_device

_LAYERS = (("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128),
           ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256))


class Vgg16(Layer):
    def __init__(self, vgg16_npy_path=None, VGG_MEAN=(103.939, 116.779, 123.68), device=None, data_dict=None):
        super().__init__()
        device = device or default_device()
        if data_dict is None:
            if vgg16_npy_path is None:  # vgg16.py:60-65: vgg16.npy beside the module
                path = os.path.abspath(os.path.join(inspect.getfile(Vgg16), os.pardir))
                vgg16_npy_path = os.path.join(path, "vgg16.npy")
            data_dict = np.load(vgg16_npy_path, encoding="latin1", allow_pickle=True).item()
        self.VGG_MEAN = list(VGG_MEAN)
        self.params = {}
        for name, cin, cout in _LAYERS:
            w = torch.as_tensor(np.asarray(data_dict[name][0]), dtype=torch.float32)
            b = torch.as_tensor(np.asarray(data_dict[name][1]), dtype=torch.float32)
            if tuple(w.shape) != (3, 3, cin, cout) or tuple(b.shape) != (cout,):
                raise ValueError("vgg16: %s has shape %s / %s" % (name, tuple(w.shape), tuple(b.shape)))
            if cin == 3:   # zero-pad Cin 3 -> 4: the first conv runs on the MFMA tile
                w = torch.cat([w, torch.zeros(3, 3, 1, cout)], dim=2)
            self.params[name] = (w.contiguous().to(device), b.contiguous().to(device))
            self.params[name][0]._shdr_const = True     # frozen: the packed Winograd form is kept on the tensor (_ops._packed_filter)

    def _conv(self, name, x):
        w, b = self.params[name]
        return K.conv2d(x, w, b, act1=K.ACT_RELU)   # vgg16.py:33-35

    def _conv_pool(self, name, x):
        w, b = self.params[name]
        return K.conv2d_maxpool2(x, w, b, act1=K.ACT_RELU, keep_y=False)

    def call(self, rgb, training="training"):
        x = K.vgg_preprocess(rgb, 4)                 # vgg16.py:101-109 (+ zero 4th channel)
        # only the pooled tensors leave the net: without a tape (the target branch of the perceptual loss) the last conv of a
        # block writes MaxPool2D(2)(relu(conv)) straight from the fused Winograd epilogue and never stores the conv output
        pool1, p1 = K.fork(self._conv_pool("conv1_2", self._conv("conv1_1", x)))      # a feature of the loss AND the next block's input
        pool2, p2 = K.fork(self._conv_pool("conv2_2", self._conv("conv2_1", p1)))
        pool3 = self._conv_pool("conv3_3", self._conv("conv3_2", self._conv("conv3_1", p2)))
        return pool1, pool2, pool3
